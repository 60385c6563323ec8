// 16-bit-operand MFMA GEMM with fused epilogues for the Swin-3D trunk (gfx950).
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]      A [M][K], W [N][K] (nn.Linear layout) bf16|fp16, fp32 acc
//
// Replaces the reference's nn.Linear calls on the hot path (swin_backbone.py:254 qkv, :323 proj,
// :84-87 fc1/fc2, :553 reduction, :726 patch-embed conv as an im2col GEMM) together with what
// surrounds them: bias, exact GELU, the q scale + head split (:255-260), window_reverse + inverse
// roll + crop + residual add (:472-488, :509, :514).
//
// Structure: 256 threads = 4 waves in a 2x2 grid; block tile (64*MI) x (64*NI), wave tile
// (32*MI) x (32*NI) built from v_mfma_f32_32x32x16_{bf16,f16}; K staged through LDS in BK slices,
// register-prefetched (global loads of slice t+1 are in flight while slice t is multiplied),
// double-buffered so there is one barrier per slice.  LDS rows are padded by 16 B: with a pitch of
// BK+8 elements the 16-lane service groups of ds_read_b128 hit 16 distinct 16-B slots (pitch/16 B
// is odd), i.e. the fragment reads are bank-conflict free.
#include "common.hpp"

namespace kvq {

struct GemmParams {
  const uint16_t* A;
  const uint16_t* W;
  const float* bias;
  int M, N, K;
  uint16_t* out_h;
  float* out_f32;
  int num_heads;
  float q_scale;
  const int32_t* scatter_map;
  int map_rows, out_rows;
};

template <typename E, int MI, int NI, int BK, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int BM = 64 * MI, BN = 64 * NI;
  constexpr int PITCH = BK + 8;            // 16-bit elements
  constexpr int CPR = BK / 8;              // 16-B chunks per row
  constexpr int A_CHUNKS = BM * CPR / 256; // per thread
  constexpr int B_CHUNKS = BN * CPR / 256;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // 2 * (BM + BN) * PITCH elements
  uint16_t* As = lds;
  uint16_t* Bs = lds + 2 * BM * PITCH;
  using V8 = typename E::v8;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // blockIdx.x walks N fastest so that blocks sharing an A row-panel are adjacent in dispatch order.
  const int nbn = (p.N + BN - 1) / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  const uint16_t* a_src[A_CHUNKS];
  int a_dst[A_CHUNKS];
#pragma unroll
  for (int i = 0; i < A_CHUNKS; ++i) {
    int c = tid + 256 * i, row = c / CPR, kc = c % CPR;
    int gm = min(m0 + row, p.M - 1);
    a_src[i] = p.A + (size_t)gm * p.K + kc * 8;
    a_dst[i] = row * PITCH + kc * 8;
  }
  const uint16_t* b_src[B_CHUNKS];
  int b_dst[B_CHUNKS];
#pragma unroll
  for (int i = 0; i < B_CHUNKS; ++i) {
    int c = tid + 256 * i, row = c / CPR, kc = c % CPR;
    int gn = min(n0 + row, p.N - 1);
    b_src[i] = p.W + (size_t)gn * p.K + kc * 8;
    b_dst[i] = row * PITCH + kc * 8;
  }

  u32x4 a_reg[A_CHUNKS], b_reg[B_CHUNKS];
  auto load_slice = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = *reinterpret_cast<const u32x4*>(a_src[i] + k0);
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) b_reg[i] = *reinterpret_cast<const u32x4*>(b_src[i] + k0);
  };
  auto store_slice = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) *reinterpret_cast<u32x4*>(As + buf * BM * PITCH + a_dst[i]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) *reinterpret_cast<u32x4*>(Bs + buf * BN * PITCH + b_dst[i]) = b_reg[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // wave-uniform: which 32-column tiles of this wave exist at all (N % 32 == 0)
  bool n_live[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) n_live[j] = (n0 + wn * 32 * NI + j * 32) < p.N;

  const int nk = p.K / BK;
  load_slice(0);
  store_slice(0);
  __syncthreads();
  const int frag_row = lane & 31, frag_k = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_slice((kt + 1) * BK);
    const uint16_t* as = As + buf * BM * PITCH + (wm * 32 * MI + frag_row) * PITCH + frag_k;
    const uint16_t* bs = Bs + buf * BN * PITCH + (wn * 32 * NI + frag_row) * PITCH + frag_k;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      V8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const V8*>(as + i * 32 * PITCH + kk * 16);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const V8*>(bs + j * 32 * PITCH + kk * 16);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          if (n_live[j]) acc[i][j] = E::mfma32(af[i], bfr[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_slice(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Each wave transposes its accumulators through a private LDS slab (32 rows x 32*NI fp32, reusing the
  // A/B slices: the K loop ended with a barrier) so that a lane owns 4 CONSECUTIVE columns of one row:
  // bias/activation run on float4s and the global stores are 8-16 B per lane, >= 128 B contiguous per
  // row, instead of 64 two-byte stores per lane.
  constexpr int SW = 32 * NI;                                   // slab width (floats)
  float* slab = reinterpret_cast<float*>(lds) + wave * 32 * SW;
  const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
  constexpr int CPRW = SW / 4;                                   // float4 chunks per slab row
  constexpr int ROWS_PER_IT = 64 / CPRW;
  const int ch = lane % CPRW, rsub = lane / CPRW;
  const int nbase = n0 + wn * SW;
  const int n = nbase + ch * 4;
  const bool col_live = n < p.N;                                 // N % 32 == 0 and 4 | 32: whole chunk in or out
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && col_live) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
  int which = 0, head = 0, e0 = 0;
  float scale = 1.f;
  if (EPI == KVQ_EPI_QKV_BF16 && col_live) {                     // a 32-column tile = one head of q|k|v
    const int C = p.N / 3;
    which = n / C;
    head = (n % C) >> 5;
    e0 = n & 31;
    scale = which == 0 ? p.q_scale : 1.f;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[((r & 3) + 8 * (r >> 2) + row_hi) * SW + j * 32 + col_in] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 / ROWS_PER_IT; ++it) {
      const int rl = it * ROWS_PER_IT + rsub;
      const int m = m0 + wm * 32 * MI + i * 32 + rl;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + rl * SW + ch * 4);
      if (m >= p.M || !col_live) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += bias4[k];
      if (EPI == KVQ_EPI_BIAS_BF16) {
        u32x2 o = {E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(p.out_h + (size_t)m * p.N + n) = o;
      } else if (EPI == KVQ_EPI_GELU_BF16) {
        u32x2 o = {E::pack2(gelu_fast(v[0]), gelu_fast(v[1])), E::pack2(gelu_fast(v[2]), gelu_fast(v[3]))};
        *reinterpret_cast<u32x2*>(p.out_h + (size_t)m * p.N + n) = o;
      } else if (EPI == KVQ_EPI_QKV_BF16) {
        u32x2 o = {E::pack2(v[0] * scale, v[1] * scale), E::pack2(v[2] * scale, v[3] * scale)};
        *reinterpret_cast<u32x2*>(p.out_h + ((size_t)(which * p.num_heads + head) * p.M + m) * 32 + e0) = o;
      } else if (EPI == KVQ_EPI_RESID_F32) {
        long orow = m;
        if (p.scatter_map) {
          const int b = m / p.map_rows, rr = m - b * p.map_rows;
          const int s = p.scatter_map[rr];
          if (s < 0) continue;
          orow = (long)b * p.out_rows + s;
        }
        f32x4* o = reinterpret_cast<f32x4*>(p.out_f32 + (size_t)orow * p.N + n);
        f32x4 x = *o;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] += v[k];
        *o = x;
      } else {  // KVQ_EPI_STORE_F32
        *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)m * p.N + n) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename E, int MI, int NI, int BK, int EPI>
static int launch_one(const GemmParams& p, hipStream_t st) {
  constexpr int BM = 64 * MI, BN = 64 * NI;
  constexpr size_t main_bytes = 2 * (BM + BN) * (BK + 8) * sizeof(uint16_t);
  constexpr size_t epi_bytes = 4 * 32 * (32 * NI) * sizeof(float);     // one fp32 slab per wave
  constexpr size_t lds_bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  auto kern = gemm_kernel<E, MI, NI, BK, EPI>;
  static bool attr_set = false;   // > 64 KiB of LDS needs the opt-in attribute (one-time, per instantiation)
  if (!attr_set && lds_bytes > 64 * 1024) {
    KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  dim3 grid(ceil_div(p.M, BM) * ceil_div(p.N, BN)), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, p);
  KVQ_CHECK_LAUNCH("gemm_kernel");
  return KVQ_OK;
}

int gemm_variant(int M, int N, int K) {
  const long blocks128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
  const bool big = blocks128 >= 512;     // >= 2 tiles per CU: use the 128x128 tile
  return (big ? 2 : 1) * 100 + ((K % 64) == 0 ? 64 : 32);
}

template <typename E, int EPI>
static int launch_gemm(const GemmParams& p, hipStream_t st) {
  const int var = gemm_variant(p.M, p.N, p.K);
  const bool big = var / 100 == 2;
  const bool k64 = var % 100 == 64;
  if (big) return k64 ? launch_one<E, 2, 2, 64, EPI>(p, st) : launch_one<E, 2, 2, 32, EPI>(p, st);
  return k64 ? launch_one<E, 1, 1, 64, EPI>(p, st) : launch_one<E, 1, 1, 32, EPI>(p, st);
}

template <int EPI>
static int launch_dt(int dtype, const GemmParams& p, hipStream_t st) {
  return dtype == KVQ_DT_FP16 ? launch_gemm<Fp16, EPI>(p, st) : launch_gemm<Bf16, EPI>(p, st);
}

}  // namespace kvq

extern "C" int kvq_gemm_bf16(const KvqGemmArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->A && a->W, KVQ_ERR_NULL, "kvq_gemm_bf16: NULL A/W");
  KVQ_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0 && a->N % 32 == 0 && a->K % 32 == 0, KVQ_ERR_SHAPE,
              "kvq_gemm_bf16: need M>0, N%%32==0, K%%32==0 (got M=%d N=%d K=%d)", a->M, a->N, a->K);
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED,
              "kvq_gemm_bf16: unknown dtype %d", a->dtype);
  GemmParams p{a->A, a->W, a->bias, a->M, a->N, a->K, a->out_bf16, a->out_f32, a->num_heads, a->q_scale,
               a->scatter_map, a->map_rows, a->out_rows};
  hipStream_t st = (hipStream_t)stream;
  switch (a->epilogue) {
    case KVQ_EPI_BIAS_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      return launch_dt<KVQ_EPI_BIAS_BF16>(a->dtype, p, st);
    case KVQ_EPI_GELU_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      return launch_dt<KVQ_EPI_GELU_BF16>(a->dtype, p, st);
    case KVQ_EPI_QKV_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      KVQ_REQUIRE(a->num_heads > 0 && a->N == 96 * a->num_heads, KVQ_ERR_SHAPE,
                  "kvq_gemm_bf16: QKV epilogue needs N == 3*32*num_heads (N=%d nH=%d)", a->N, a->num_heads);
      return launch_dt<KVQ_EPI_QKV_BF16>(a->dtype, p, st);
    case KVQ_EPI_RESID_F32:
      KVQ_REQUIRE(a->out_f32, KVQ_ERR_NULL, "kvq_gemm_bf16: out_f32 NULL");
      KVQ_REQUIRE(!a->scatter_map || (a->map_rows > 0 && a->out_rows > 0), KVQ_ERR_SHAPE,
                  "kvq_gemm_bf16: scatter map needs map_rows/out_rows");
      return launch_dt<KVQ_EPI_RESID_F32>(a->dtype, p, st);
    case KVQ_EPI_STORE_F32:
      KVQ_REQUIRE(a->out_f32, KVQ_ERR_NULL, "kvq_gemm_bf16: out_f32 NULL");
      return launch_dt<KVQ_EPI_STORE_F32>(a->dtype, p, st);
    default:
      set_error("kvq_gemm_bf16: unknown epilogue %d", a->epilogue);
      return KVQ_ERR_UNSUPPORTED;
  }
}
