// PatchMerging as ONE launch (gfx950) at C = 96: the 4-neighbour concat + LayerNorm(4C) + Linear(4C -> 2C, no bias)
// (swin_backbone.py:533-556) [+ norm1 + pad/roll/window_partition of the next stage's first block, :416-449] — three launches
// (gather-LayerNorm 26 us, GEMM 26 us, LayerNorm 16 us at 4 clips) and a 16-bit [rows][4C] round trip through HBM.
//
// Token-per-lane, like csrc/embed.hip: Out^T[2C][32 tokens] = W'[2C][4C] * d^T[4C][32 tokens] with the weights as the MFMA A
// operand (fragment-major, the whole 144 KB matrix resident in LDS: one DMA burst per workgroup) and the merged tokens as the 32
// columns of v_mfma_f32_32x32x16.  LayerNorm is affine, so it is folded around the GEMM:
//     W (gamma * (x - mean) * rstd + beta)  =  rstd * (W diag(gamma)) (x - mean)  +  W beta
// W' = W diag(gamma) (16-bit), its row sums and W beta (fp32) are built once by kvq_patch_merge_pack.  The launch is bound by the
// per-lane row accesses (a wave-load of 64 different rows costs the texture addresser 64 cycles, not 16), so the rows are read
// ONCE: the B operand is d = x - K in 16 bits with K = the token's first value (a shift inside the row's own range, shared by the
// lane pair of a token), read straight from the residual stream — k-step s is 16 channels of neighbour s / 6, lane half h takes 8
// of them, two 16-byte loads — and sum(d), sum(d^2) accumulate on the way (shifted one-pass variance: with K inside the data the
// subtraction sum(d^2)/n - mean_d^2 loses a few bits at most).  mean - K leaves the GEMM through the row sums:
//     W' (x - mean) = W' d - (mean - K) W' 1.
// A lane owns all 2C outputs of its token: the epilogue applies rstd, the two corrections, stores the fp32 residual stream of the
// next stage and, in registers, the next block's norm1 row in ITS window order.
#include "common.hpp"

namespace kvq {

struct MergeParams {
  const float* x;            // [B*L][C]
  const int32_t* map;        // [Ln][4] neighbour tokens (concat order x0 x1 x2 x3, swin_backbone.py:546-550), -1 = zero padding
  int B, L, Ln;
  const unsigned char* pack;
  float* out;                // [B*Ln][2C]
  const float* nn_w;         // next block's norm1 (EMIT)
  const float* nn_b;
  const int32_t* next_dst;   // merged token -> window row
  uint16_t* next_ln;         // [B*next_rows][2C]
  int next_rows;
  float eps;
};

typedef __attribute__((address_space(3))) void* mg_lds_t;
typedef __attribute__((address_space(1))) const void* mg_gbl_t;

constexpr int MG_C = 96, MG_K = 4 * MG_C, MG_N = 2 * MG_C, MG_CM = MG_N / 32, MG_KS = MG_K / 16, MG_WBYTES = MG_CM * MG_KS * 1024;
constexpr int MG_QS = MG_C / 16;                        // k-steps per neighbour
constexpr int MG_WAVES = 8, MG_TOK = 32 * MG_WAVES;
constexpr int MG_PACK_BYTES = MG_WBYTES + 2048;         // + fp32 W beta [2C] | row sums of W' [2C] (padded to 2 KB)
constexpr int MG_LDS = MG_PACK_BYTES + 2 * MG_N * 4;    // + the next norm1's gamma | beta

// image: CM panels [KS fragment rows][64 lanes][8 x 16-bit] = W'[32 i + m][16 s + 8 h + e], W' = W diag(gamma); then fp32 (W beta)[2C] and
// the row sums of W' AS ROUNDED (what the MFMAs multiply) [2C]
template <typename E>
__global__ void merge_pack_kernel(const float* w, const float* gamma, const float* beta, unsigned char* out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int n_chunks = MG_WBYTES / 16;
  if (g < n_chunks) {
    const int panel = g / (MG_KS * 64), rem = g % (MG_KS * 64);
    const int s = rem >> 6, lane = rem & 63, m = lane & 31, h = lane >> 5;
    const float* row = w + (size_t)(32 * panel + m) * MG_K + 16 * s + 8 * h;
    const float* gm = gamma + 16 * s + 8 * h;
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = E::pack2(row[2 * e] * gm[2 * e], row[2 * e + 1] * gm[2 * e + 1]);
    *reinterpret_cast<u32x4*>(out + (size_t)g * 16) = v;
  } else if (g < n_chunks + MG_N) {
    const int n = g - n_chunks;
    float acc = 0.f, rs = 0.f;
    for (int k = 0; k < MG_K; ++k) {
      acc = fmaf(w[(size_t)n * MG_K + k], beta[k], acc);
      rs += E::to_f32(E::cvt(w[(size_t)n * MG_K + k] * gamma[k]));
    }
    reinterpret_cast<float*>(out + MG_WBYTES)[n] = acc;
    reinterpret_cast<float*>(out + MG_WBYTES)[MG_N + n] = rs;
  }
}

template <typename E, bool EMIT>
__global__ __launch_bounds__(64 * MG_WAVES, 1) void patch_merge_kernel(MergeParams p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E::v8;
  const float* wbeta = reinterpret_cast<const float*>(lds + MG_WBYTES);
  float* s_nn = reinterpret_cast<float*>(lds + MG_PACK_BYTES);           // [gamma 2C][beta 2C]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int q = wave; q < MG_PACK_BYTES / 1024; q += MG_WAVES)            // weights + W beta: 1 KB wave-loads
    __builtin_amdgcn_global_load_lds((mg_gbl_t)(p.pack + q * 1024 + lane * 16), (mg_lds_t)(lds + q * 1024), 16, 0, 0);
  if (EMIT && tid < 2 * MG_N / 4)
    *reinterpret_cast<f32x4*>(s_nn + 4 * tid) = *reinterpret_cast<const f32x4*>((tid < MG_N / 4 ? p.nn_w : p.nn_b - MG_N) + 4 * tid);

  // this lane's merged token and its four neighbour rows (channels 8 h .. 8 h + 7 of every 16)
  const long total = (long)p.B * p.Ln;
  const long row = (long)blockIdx.x * MG_TOK + wave * 32 + j;
  const long rc = row < total ? row : total - 1;
  const int b = (int)(rc / p.Ln), r = (int)(rc - (long)b * p.Ln);
  const bool live = row < total;
  const i32x4 nb = *reinterpret_cast<const i32x4*>(p.map + 4 * (size_t)r);
  const float* src[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) src[n] = p.x + ((size_t)b * p.L + (nb[n] < 0 ? 0 : nb[n])) * MG_C + 8 * h;
  // ---- one pass: d = x - K -> 16-bit B operand, sum(d) and sum(d^2) on the way; 6 k-steps x 6 row tiles per neighbour ----
  f32x16 acc[MG_CM];
#pragma unroll
  for (int i = 0; i < MG_CM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float sq = 0.f, s1 = 0.f;
  constexpr int PF = 6;                                   // k-steps of row pieces in flight (two 16-byte loads each)
  f32x4 ring[PF][2];
  auto piece = [&](int s, f32x4 (&v)[2]) __attribute__((always_inline)) {       // k-step s = neighbour s / 6, channels 16 (s % 6) + 8 h ..
    const int n = s / MG_QS, q = s - n * MG_QS;
    v[0] = *reinterpret_cast<const f32x4*>(src[n] + 16 * q);
    v[1] = *reinterpret_cast<const f32x4*>(src[n] + 16 * q + 4);
    if (nb[n] < 0) v[0] = v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};       // F.pad zeros take part in the statistics (swin_backbone.py:541-544)
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) piece(s, ring[s]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the weight image, and the first row pieces behind it
  __syncthreads();
  const float shift = __shfl(ring[0][0][0], j);            // K: channel 0 of neighbour 0, from the h = 0 lane of the token
#pragma unroll
  for (int s = 0; s < MG_KS; ++s) {
    float d[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[e] = ring[s % PF][0][e] - shift;
      d[4 + e] = ring[s % PF][1][e] - shift;
    }
    if (s + PF < MG_KS) piece(s + PF, ring[s % PF]);      // the slot just read takes the piece PF k-steps ahead
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1 += d[e];
      sq = fmaf(d[e], d[e], sq);
    }
    const u32x4 w = {E::pack2(d[0], d[1]), E::pack2(d[2], d[3]), E::pack2(d[4], d[5]), E::pack2(d[6], d[7])};
    const V8 bx = __builtin_bit_cast(V8, w);
#pragma unroll
    for (int i = 0; i < MG_CM; ++i) {
      const V8 a = *reinterpret_cast<const V8*>(lds + (i * MG_KS + s) * 1024 + lane * 16);
      acc[i] = E::mfma32(a, bx, acc[i]);
    }
  }
  s1 += __shfl_xor(s1, 32);
  sq += __shfl_xor(sq, 32);
  const float md = s1 * (1.0f / (float)MG_K);                                  // mean - K
  const float rstd = rsqrtf(fmaxf(sq * (1.0f / (float)MG_K) - md * md, 0.f) + p.eps);

  // ---- out = rstd * (acc - (mean - K) * rowsum(W')) + W beta; tile i, register 4 q + e <-> channel 32 i + 8 q + 4 h + e ----
#pragma unroll
  for (int i = 0; i < MG_CM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wb = *reinterpret_cast<const f32x4*>(wbeta + 32 * i + 8 * q + 4 * h);
      const f32x4 ws = *reinterpret_cast<const f32x4*>(wbeta + MG_N + 32 * i + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = fmaf(fmaf(-md, ws[e], acc[i][4 * q + e]), rstd, wb[e]);
    }
  if (live) {
    float* o = p.out + (size_t)rc * MG_N + 4 * h;
#pragma unroll
    for (int i = 0; i < MG_CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(o + 32 * i + 8 * q) = (f32x4){acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
  }
  if (EMIT) {
    // LayerNorm over the 2C channels of the merged token: in-lane sums + one exchange with lane ^ 32 (two-pass, as ln.hip)
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MG_CM; ++i)
#pragma unroll
      for (int e = 0; e < 16; e += 4) s1 += (acc[i][e] + acc[i][e + 1]) + (acc[i][e + 2] + acc[i][e + 3]);
    s1 += __shfl_xor(s1, 32);
    float m2 = s1 * (1.0f / (float)MG_N);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MG_CM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float dd = acc[i][e] - m2;
        s2 = fmaf(dd, dd, s2);
      }
    s2 += __shfl_xor(s2, 32);
    const float r2 = rsqrtf(s2 * (1.0f / (float)MG_N) + p.eps);
    asm volatile("" : "+v"(m2));       // opaque: no CSE of (acc - mean) between the variance and the normalise pass
    if (live) {
      const long drow = (long)b * p.next_rows + p.next_dst[r];
      uint16_t* o = p.next_ln + (size_t)drow * MG_N + 4 * h;
#pragma unroll
      for (int i = 0; i < MG_CM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(s_nn + 32 * i + 8 * q + 4 * h);
          const f32x4 be = *reinterpret_cast<const f32x4*>(s_nn + MG_N + 32 * i + 8 * q + 4 * h);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (acc[i][4 * q + e] - m2) * r2 * g[e] + be[e];
          *reinterpret_cast<u32x2*>(o + 32 * i + 8 * q) = (u32x2){E::pack2(y[0], y[1]), E::pack2(y[2], y[3])};
        }
    }
  }
}

template <typename E>
static int launch_merge(const MergeParams& p, hipStream_t st) {
  const long total = (long)p.B * p.Ln;
  dim3 grid((unsigned)((total + MG_TOK - 1) / MG_TOK)), block(64 * MG_WAVES);
  if (p.next_ln) {
    auto k = patch_merge_kernel<E, true>;
    static LdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(k), MG_LDS)) return rc;
    hipLaunchKernelGGL(k, grid, block, MG_LDS, st, p);
  } else {
    auto k = patch_merge_kernel<E, false>;
    static LdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(k), MG_LDS)) return rc;
    hipLaunchKernelGGL(k, grid, block, MG_LDS, st, p);
  }
  KVQ_CHECK_LAUNCH("patch_merge_kernel");
  return KVQ_OK;
}

}  // namespace kvq

extern "C" int kvq_patch_merge_supported(int C) { return C == kvq::MG_C ? 1 : 0; }

extern "C" size_t kvq_patch_merge_pack_bytes(int C) { return C == kvq::MG_C ? (size_t)kvq::MG_PACK_BYTES : 0; }

extern "C" int kvq_patch_merge_pack(const float* red_w, const float* norm_w, const float* norm_b, int C, int dtype, void* pack,
                                    void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(red_w && norm_w && norm_b && pack, KVQ_ERR_NULL, "kvq_patch_merge_pack: NULL pointer");
  KVQ_REQUIRE(kvq_patch_merge_supported(C), KVQ_ERR_UNSUPPORTED, "kvq_patch_merge_pack: C=%d", C);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_patch_merge_pack: dtype %d", dtype);
  const int total = MG_WBYTES / 16 + MG_N;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(merge_pack_kernel<Fp16>, grid, block, 0, (hipStream_t)stream, red_w, norm_w, norm_b, (unsigned char*)pack);
  else hipLaunchKernelGGL(merge_pack_kernel<Bf16>, grid, block, 0, (hipStream_t)stream, red_w, norm_w, norm_b, (unsigned char*)pack);
  KVQ_CHECK_LAUNCH("merge_pack_kernel");
  return KVQ_OK;
}

extern "C" int kvq_patch_merge(const KvqPatchMergeArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->x && a->merge_map && a->pack && a->out, KVQ_ERR_NULL, "kvq_patch_merge: NULL pointer");
  KVQ_REQUIRE(kvq_patch_merge_supported(a->C), KVQ_ERR_UNSUPPORTED, "kvq_patch_merge: C=%d is not the fused width", a->C);
  KVQ_REQUIRE(a->B > 0 && a->L > 0 && a->Ln > 0, KVQ_ERR_SHAPE, "kvq_patch_merge: B=%d L=%d Ln=%d", a->B, a->L, a->Ln);
  KVQ_REQUIRE(!a->next_ln || (a->next_norm_w && a->next_norm_b && a->next_dst && a->next_rows > 0), KVQ_ERR_NULL,
              "kvq_patch_merge: next_ln without its norm / map");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_patch_merge: dtype %d", a->dtype);
  KVQ_REQUIRE((((size_t)a->x | (size_t)a->out | (size_t)a->merge_map) & 15) == 0, KVQ_ERR_SHAPE, "kvq_patch_merge: 16-byte aligned buffers");
  MergeParams p{};
  p.x = a->x; p.map = a->merge_map; p.B = a->B; p.L = a->L; p.Ln = a->Ln; p.pack = (const unsigned char*)a->pack; p.out = a->out;
  p.nn_w = a->next_norm_w; p.nn_b = a->next_norm_b; p.next_dst = a->next_dst; p.next_ln = (uint16_t*)a->next_ln; p.next_rows = a->next_rows;
  p.eps = a->eps;
  return a->dtype == KVQ_DT_FP16 ? launch_merge<Fp16>(p, (hipStream_t)stream) : launch_merge<Bf16>(p, (hipStream_t)stream);
}
