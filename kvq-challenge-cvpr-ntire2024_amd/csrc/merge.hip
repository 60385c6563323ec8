// PatchMerging as ONE launch (gfx950) at C = 96 / 128 / 192: the 4-neighbour concat + LayerNorm(4C) + Linear(4C -> 2C, no bias)
// (swin_backbone.py:533-556) [+ norm1 + pad/roll/window_partition of the next stage's first block, :416-449] — three launches
// (gather-LayerNorm 26 us, GEMM 26 us, LayerNorm 16 us at 4 clips) and a 16-bit [rows][4C] round trip through HBM.
//
// Token-per-lane, like csrc/embed.hip: Out^T[2C][32 tokens] = W'[2C][4C] * d^T[4C][32 tokens] with the weights as the MFMA A
// operand (fragment-major in LDS: the whole 144 KB matrix at C = 96, two 64-72 KB chunk buffers it streams through above) and the merged tokens as the 32
// columns of v_mfma_f32_32x32x16.  LayerNorm is affine, so it is folded around the GEMM:
//     W (gamma * (x - mean) * rstd + beta)  =  rstd * (W diag(gamma)) (x - mean)  +  W beta
// W' = W diag(gamma) (16-bit), its row sums and W beta (fp32) are built once by kvq_patch_merge_pack.  The launch is bound by the
// per-lane row accesses (a wave-load of 64 different rows costs the texture addresser 64 cycles, not 16), so the rows are read
// ONCE: the B operand is d = x - K in 16 bits with K = the mean of the token's first 96 channels (a shift inside the row's own range that one outlier channel cannot drag away, shared by the
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
  int x16, out16;            // round 6: x / out are fp16 residual streams (rows of 2 C / 4 C bytes behind the same pointers)
  const float* nn_w;         // next block's norm1 (EMIT)
  const float* nn_b;
  const int32_t* next_dst;   // merged token -> window row
  uint16_t* next_ln;         // [B*next_rows][2C]
  int next_rows;
  float eps;
};

typedef __attribute__((address_space(3))) void* mg_lds_t;
typedef __attribute__((address_space(1))) const void* mg_gbl_t;

// geometry by width.  The weight image is cut into chunks of KC k-steps (all 2C / 32 row tiles of those k-steps: <= 72 KB) that
// stream through two LDS buffers; at C = 96 the two chunks ARE the matrix (both requested at the start, nothing refilled).
#ifndef KVQ_MERGE96_SMALL
#define KVQ_MERGE96_SMALL 0
#endif
template <int C_>
struct MGc {
  static constexpr int C = C_, K = 4 * C, N = 2 * C, CM = N / 32, KS = K / 16, QS = C / 16;      // QS: k-steps per neighbour
  // KVQ_MERGE96_SMALL (round 5 A/B): at C = 96 stream the matrix through two 36 KB chunk buffers for 4 waves (75 KB of LDS: two workgroups
  // per CU, room for another lane's workgroup) instead of keeping all 144 KB resident for 8 waves
  static constexpr int KC = C == 96 ? (KVQ_MERGE96_SMALL ? 6 : 12) : C == 128 ? 8 : 6;
  static constexpr int NCH = KS / KC, CHUNK = CM * KC * 1024, WBYTES = CM * KS * 1024;
  static constexpr int WAVES = C == 96 && !KVQ_MERGE96_SMALL ? 8 : 4, TOK = 32 * WAVES;       // wider rows: one wave per SIMD (the accumulators alone are 128 / 192 registers)
  static constexpr int TAIL = ((2 * N * 4) + 1023) & ~1023;               // fp32 W beta [2C] | row sums of W' [2C], whole KB
  static constexpr int PACK_BYTES = WBYTES + TAIL;
  static constexpr int LDS = 2 * CHUNK + TAIL + 2 * N * 4;                // + the next norm1's gamma | beta
  static_assert(KS % KC == 0 && CHUNK % (1024 * WAVES) == 0 && LDS <= 163840, "chunking");
};

// image: chunk c = [CM panels][KC fragment rows][64 lanes][8 x 16-bit] = W'[32 i + m][16 (c KC + sl) + 8 h + e], W' = W diag(gamma);
// then fp32 (W beta)[2C] and the row sums of W' AS ROUNDED (what the MFMAs multiply) [2C]
template <typename E, int C_>
__global__ void merge_pack_kernel(const float* w, const float* gamma, const float* beta, unsigned char* out) {
  using G = MGc<C_>;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int n_chunks = G::WBYTES / 16;
  if (g < n_chunks) {
    const int lane = g & 63, fr = g >> 6;                                 // fragment = (chunk, panel, k-step in chunk)
    const int sl = fr % G::KC, panel = (fr / G::KC) % G::CM, c = fr / (G::KC * G::CM);
    const int s = c * G::KC + sl, m = lane & 31, h = lane >> 5;
    const float* row = w + (size_t)(32 * panel + m) * G::K + 16 * s + 8 * h;
    const float* gm = gamma + 16 * s + 8 * h;
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = E::pack2(row[2 * e] * gm[2 * e], row[2 * e + 1] * gm[2 * e + 1]);
    *reinterpret_cast<u32x4*>(out + (size_t)g * 16) = v;
  } else if (g < n_chunks + G::N) {
    const int n = g - n_chunks;
    float acc = 0.f, rs = 0.f;
    for (int k = 0; k < G::K; ++k) {
      acc = fmaf(w[(size_t)n * G::K + k], beta[k], acc);
      rs += E::to_f32(E::cvt(w[(size_t)n * G::K + k] * gamma[k]));
    }
    reinterpret_cast<float*>(out + G::WBYTES)[n] = acc;
    reinterpret_cast<float*>(out + G::WBYTES)[G::N + n] = rs;
  }
}

// X16: the input stream is fp16 (round 6) — a compile-time form: a run-time branch inside the unrolled row-piece loop made the compiler wait
// for every load where it is issued (C = 192: 65.7 -> 88.8 us)
template <typename E, bool EMIT, int C_, bool X16 = false>
__global__ __launch_bounds__(64 * MGc<C_>::WAVES, (2 * MGc<C_>::LDS <= 163840 ? 2 : 1)) void patch_merge_kernel(MergeParams p) {
  using G = MGc<C_>;
  constexpr int C = G::C, K = G::K, N = G::N, CM = G::CM, KS = G::KS, QS = G::QS, KC = G::KC, NCH = G::NCH, CHUNK = G::CHUNK, WAVES = G::WAVES;
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E::v8;
  const float* wbeta = reinterpret_cast<const float*>(lds + 2 * CHUNK);
  float* s_nn = reinterpret_cast<float*>(lds + 2 * CHUNK + G::TAIL);     // [gamma 2C][beta 2C]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  auto issue_chunk = [&](int c) __attribute__((always_inline)) {         // chunk c -> buffer c & 1, 1 KB wave-loads
    for (int q = wave; q < CHUNK / 1024; q += WAVES)
      __builtin_amdgcn_global_load_lds((mg_gbl_t)(p.pack + (size_t)c * CHUNK + q * 1024 + lane * 16),
                                       (mg_lds_t)(lds + (c & 1) * CHUNK + q * 1024), 16, 0, 0);
  };
  for (int q = wave; q < G::TAIL / 1024; q += WAVES)
    __builtin_amdgcn_global_load_lds((mg_gbl_t)(p.pack + G::WBYTES + q * 1024 + lane * 16), (mg_lds_t)(lds + 2 * CHUNK + q * 1024), 16, 0, 0);
  issue_chunk(0);
  if (NCH > 1) issue_chunk(1);
  if (EMIT && tid < 2 * N / 4)
    *reinterpret_cast<f32x4*>(s_nn + 4 * tid) = *reinterpret_cast<const f32x4*>((tid < N / 4 ? p.nn_w : p.nn_b - N) + 4 * tid);

  // this lane's merged token and its four neighbour rows (channels 8 h .. 8 h + 7 of every 16)
  const long total = (long)p.B * p.Ln;
  const long row = (long)blockIdx.x * G::TOK + wave * 32 + j;
  const long rc = row < total ? row : total - 1;
  const int b = (int)(rc / p.Ln), r = (int)(rc - (long)b * p.Ln);
  const bool live = row < total;
  const i32x4 nb = *reinterpret_cast<const i32x4*>(p.map + 4 * (size_t)r);
  const float* src[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) src[n] = p.x + ((size_t)b * p.L + (nb[n] < 0 ? 0 : nb[n])) * C + 8 * h;
  // ---- one pass: d = x - K -> 16-bit B operand, sum(d) and sum(d^2) on the way; C / 16 k-steps x 2C / 32 row tiles per neighbour ----
  f32x16 acc[CM];
#pragma unroll
  for (int i = 0; i < CM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float sq = 0.f, s1 = 0.f;
  constexpr int PF = 6;                                   // k-steps of row pieces in flight (two 16-byte loads each)
  f32x4 ring[PF][2];
  // fp16 stream (x16): the 8 channels of a piece are ONE 16-byte load; it stays raw in v[0] (widened where it is used: a conversion here
  // would wait for the load and undo the PF k-steps of prefetch)
  auto piece = [&](int s, f32x4 (&v)[2]) __attribute__((always_inline)) {       // k-step s = neighbour s / QS, channels 16 (s % QS) + 8 h ..
    const int n = s / QS, q = s - n * QS;
    if (X16) {
      v[0] = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.x) + (src[n] - p.x) + 16 * q));
      if (nb[n] < 0) v[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      return;
    }
    v[0] = *reinterpret_cast<const f32x4*>(src[n] + 16 * q);
    v[1] = *reinterpret_cast<const f32x4*>(src[n] + 16 * q + 4);
    if (nb[n] < 0) v[0] = v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};       // F.pad zeros take part in the statistics (swin_backbone.py:541-544)
  };
  auto widen = [&](const f32x4 (&v)[2], float (&o)[8]) __attribute__((always_inline)) {      // the 8 channels of a piece as fp32
    if (X16) {
      const f16x8 hv = __builtin_bit_cast(f16x8, v[0]);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (float)hv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = v[0][e]; o[4 + e] = v[1][e]; }
    }
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) piece(s, ring[s]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first two weight chunks, and the first row pieces behind them
  __syncthreads();
  // K: the mean of the token's first PF k-steps (PF x 8 values in each lane of the pair = 96 channels of neighbour 0).  A single
  // value (channel 0, rounds 4's choice) is not robust: one massive-activation channel there puts every one of the 4C operands near
  // |K|, where the 16-bit ulp is the operand's whole signal; a mean over 96 channels moves by outlier / 96.
  float shift = 0.f;
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    float o8[8];
    widen(ring[s], o8);
#pragma unroll
    for (int e = 0; e < 4; ++e) shift += o8[e] + o8[4 + e];
  }
  shift += __shfl_xor(shift, 32);
  shift *= 1.0f / (float)(PF * 16);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (NCH > 2 && s > 0 && s % KC == 0) {
      // chunk s / KC - 1 is done: once every wave is past it, its buffer takes chunk s / KC + 1.  The chunk about to be used
      // was requested a whole chunk ago; the wait also drains the row pieces in flight (a bubble per chunk at C >= 128).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s / KC + 1 < NCH) issue_chunk(s / KC + 1);
    }
    float d[8];
    widen(ring[s % PF], d);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] -= shift;
    if (s + PF < KS) piece(s + PF, ring[s % PF]);          // the slot just read takes the piece PF k-steps ahead
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1 += d[e];
      sq = fmaf(d[e], d[e], sq);
    }
    const u32x4 w = {E::pack2(d[0], d[1]), E::pack2(d[2], d[3]), E::pack2(d[4], d[5]), E::pack2(d[6], d[7])};
    const V8 bx = __builtin_bit_cast(V8, w);
    const unsigned char* wb = lds + ((s / KC) & 1) * CHUNK + (s % KC) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < CM; ++i) {
      const V8 a = *reinterpret_cast<const V8*>(wb + i * KC * 1024);
      acc[i] = E::mfma32(a, bx, acc[i]);
      if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // keeps the fragment reads from piling up in registers
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  s1 += __shfl_xor(s1, 32);
  sq += __shfl_xor(sq, 32);
  const float md = s1 * (1.0f / (float)K);                                    // mean - K
  const float rstd = rsqrtf(fmaxf(sq * (1.0f / (float)K) - md * md, 0.f) + p.eps);

  // ---- out = rstd * (acc - (mean - K) * rowsum(W')) + W beta; tile i, register 4 q + e <-> channel 32 i + 8 q + 4 h + e.
  // The accumulator tiles are only READ from here on (as in tail.hip: element-wise updates of a 16-register MFMA tuple make
  // the allocator copy whole tuples around and spill): every pass below recomputes the two fma of a value it needs.
  auto outv = [&](int i, int q) __attribute__((always_inline)) -> f32x4 {
    const f32x4 wbv = *reinterpret_cast<const f32x4*>(wbeta + 32 * i + 8 * q + 4 * h);
    const f32x4 ws = *reinterpret_cast<const f32x4*>(wbeta + N + 32 * i + 8 * q + 4 * h);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(fmaf(-md, ws[e], acc[i][4 * q + e]), rstd, wbv[e]);
    return v;
  };
  float t1 = 0.f;
  {
    float* o = p.out + (size_t)rc * N + 4 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = outv(i, q);
        if (live && p.out16) *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (size_t)rc * N + 4 * h + 32 * i + 8 * q) = (u32x2){Fp16::pack2(v[0], v[1]), Fp16::pack2(v[2], v[3])};
        else if (live) *reinterpret_cast<f32x4*>(o + 32 * i + 8 * q) = v;
        if (EMIT) t1 += (v[0] + v[1]) + (v[2] + v[3]);
        if (q == 3) __builtin_amdgcn_sched_barrier(0);
      }
  }
  if (EMIT) {
    // LayerNorm over the 2C channels of the merged token: in-lane sums + one exchange with lane ^ 32 (two-pass, as ln.hip)
    t1 += __shfl_xor(t1, 32);
    const float m2 = t1 * (1.0f / (float)N);
    float t2 = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = outv(i, q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dd = v[e] - m2;
          t2 = fmaf(dd, dd, t2);
        }
        if (q == 3) __builtin_amdgcn_sched_barrier(0);
      }
    t2 += __shfl_xor(t2, 32);
    const float r2 = rsqrtf(t2 * (1.0f / (float)N) + p.eps);
    // 16-bit rows leave as 16 bytes per lane: a lane pair (h = 0 | 1) holds 4 + 4 consecutive channels of every 8; the pair
    // exchanges the 8-byte pieces of (q, q + 1) by v_permlane32_swap, lane h then owns channels 8 (2 t + h) .. + 7 — half the
    // row-divergent store instructions
    const long drow = (long)b * p.next_rows + p.next_dst[r];
    uint16_t* o = p.next_ln + (size_t)drow * N;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t pk[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = 2 * t + u;
          const f32x4 v = outv(i, q);
          const f32x4 g = *reinterpret_cast<const f32x4*>(s_nn + 32 * i + 8 * q + 4 * h);
          const f32x4 be = *reinterpret_cast<const f32x4*>(s_nn + N + 32 * i + 8 * q + 4 * h);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (v[e] - m2) * r2 * g[e] + be[e];
          pk[u][0] = E::pack2(y[0], y[1]);
          pk[u][1] = E::pack2(y[2], y[3]);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        if (live) *reinterpret_cast<u32x4*>(o + 32 * i + 8 * (2 * t + h)) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_sched_barrier(0);
      }
  }
}

template <typename E, int C_>
static int launch_merge(const MergeParams& p, hipStream_t st) {
  using G = MGc<C_>;
  const long total = (long)p.B * p.Ln;
  dim3 grid((unsigned)((total + G::TOK - 1) / G::TOK)), block(64 * G::WAVES);
  auto go = [&](auto k) -> int {
    LdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(k), G::LDS)) return rc;
    hipLaunchKernelGGL(k, grid, block, G::LDS, st, p);
    return KVQ_OK;
  };
  int rc;
  if (p.next_ln) rc = p.x16 ? go(patch_merge_kernel<E, true, C_, true>) : go(patch_merge_kernel<E, true, C_, false>);
  else rc = p.x16 ? go(patch_merge_kernel<E, false, C_, true>) : go(patch_merge_kernel<E, false, C_, false>);
  if (rc) return rc;
  KVQ_CHECK_LAUNCH("patch_merge_kernel");
  return KVQ_OK;
}

template <typename E>
static int launch_merge_c(int C, const MergeParams& p, hipStream_t st) {
  return C == 96 ? launch_merge<E, 96>(p, st) : C == 128 ? launch_merge<E, 128>(p, st) : launch_merge<E, 192>(p, st);
}

template <typename E, int C_>
static void launch_merge_pack(const float* w, const float* g, const float* b, unsigned char* out, hipStream_t st) {
  const int total = MGc<C_>::WBYTES / 16 + MGc<C_>::N;
  hipLaunchKernelGGL((merge_pack_kernel<E, C_>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, g, b, out);
}

}  // namespace kvq

extern "C" int kvq_patch_merge_supported(int C) { return C == 96 || C == 128 || C == 192 ? 1 : 0; }

extern "C" size_t kvq_patch_merge_pack_bytes(int C) {
  using namespace kvq;
  return C == 96 ? (size_t)MGc<96>::PACK_BYTES : C == 128 ? (size_t)MGc<128>::PACK_BYTES : C == 192 ? (size_t)MGc<192>::PACK_BYTES : 0;
}

extern "C" int kvq_patch_merge_pack(const float* red_w, const float* norm_w, const float* norm_b, int C, int dtype, void* pack,
                                    void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(red_w && norm_w && norm_b && pack, KVQ_ERR_NULL, "kvq_patch_merge_pack: NULL pointer");
  KVQ_REQUIRE(kvq_patch_merge_supported(C), KVQ_ERR_UNSUPPORTED, "kvq_patch_merge_pack: C=%d", C);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_patch_merge_pack: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  unsigned char* o = (unsigned char*)pack;
  if (dtype == KVQ_DT_FP16) {
    if (C == 96) launch_merge_pack<Fp16, 96>(red_w, norm_w, norm_b, o, st);
    else if (C == 128) launch_merge_pack<Fp16, 128>(red_w, norm_w, norm_b, o, st);
    else launch_merge_pack<Fp16, 192>(red_w, norm_w, norm_b, o, st);
  } else {
    if (C == 96) launch_merge_pack<Bf16, 96>(red_w, norm_w, norm_b, o, st);
    else if (C == 128) launch_merge_pack<Bf16, 128>(red_w, norm_w, norm_b, o, st);
    else launch_merge_pack<Bf16, 192>(red_w, norm_w, norm_b, o, st);
  }
  KVQ_CHECK_LAUNCH("merge_pack_kernel");
  return KVQ_OK;
}

extern "C" int kvq_patch_merge(const KvqPatchMergeArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->x && a->merge_map && a->pack && a->out, KVQ_ERR_NULL, "kvq_patch_merge: NULL pointer");
  KVQ_REQUIRE(kvq_patch_merge_supported(a->C), KVQ_ERR_UNSUPPORTED, "kvq_patch_merge: C=%d is not a fused width", a->C);
  KVQ_REQUIRE(a->B > 0 && a->L > 0 && a->Ln > 0, KVQ_ERR_SHAPE, "kvq_patch_merge: B=%d L=%d Ln=%d", a->B, a->L, a->Ln);
  KVQ_REQUIRE(!a->next_ln || (a->next_norm_w && a->next_norm_b && a->next_dst && a->next_rows > 0), KVQ_ERR_NULL,
              "kvq_patch_merge: next_ln without its norm / map");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_patch_merge: dtype %d", a->dtype);
  KVQ_REQUIRE((((size_t)a->x | (size_t)a->out | (size_t)a->merge_map) & 15) == 0, KVQ_ERR_SHAPE, "kvq_patch_merge: 16-byte aligned buffers");
  MergeParams p{};
  p.x = a->x; p.map = a->merge_map; p.B = a->B; p.L = a->L; p.Ln = a->Ln; p.pack = (const unsigned char*)a->pack; p.out = a->out; p.x16 = a->x_f16; p.out16 = a->out_f16;
  p.nn_w = a->next_norm_w; p.nn_b = a->next_norm_b; p.next_dst = a->next_dst; p.next_ln = (uint16_t*)a->next_ln; p.next_rows = a->next_rows;
  p.eps = a->eps;
  return a->dtype == KVQ_DT_FP16 ? launch_merge_c<Fp16>(a->C, p, (hipStream_t)stream) : launch_merge_c<Bf16>(a->C, p, (hipStream_t)stream);
}
