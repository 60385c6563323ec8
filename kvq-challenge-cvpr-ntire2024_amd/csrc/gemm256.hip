// The 256 x 256 x 64 eight-phase GEMM (main loop: gemm8p.hpp) behind the launch parameters and fused epilogues of
// kvq_gemm_bf16 / kvq_conv_implicit (gemm_common.hpp).  Replaces, for the shapes gemm8p_wanted() names, the nn.Linear calls of
// swin_backbone.py:64-89 (Mlp), :252-326 (qkv / proj), :533-556 (reduction) and the conv nets' long-K convolutions.
#include <stdlib.h>

#include "gemm8p.hpp"
#include "gemm_common.hpp"

namespace kvq {

template <typename E, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(GemmParams p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int nbm = (p.M + g8::BM - 1) / g8::BM, nbn = (p.N + g8::BN - 1) / g8::BN;
  const int lid0 = g8::logical_block();
  const int ntile = nbm * nbn, ksl = lid0 / ntile;                     // K range index (slowest), tile
  int bm, bn;
  g8::tile_of(lid0 - ksl * ntile, nbm, nbn, bm, bn);
  const int m0 = bm * g8::BM, n0 = bn * g8::BN;
  const int nkt = p.K / g8::BK;
  const int kt0 = ksl * nkt / p.ksplit, nk = (ksl + 1) * nkt / p.ksplit - kt0;

  g8::StageGeom sg;
  sg.init();
  g8::PlainSrc sa, sb;
  if (p.a_gather) {          // row m reads A row (m / a_rows) * a_phys_rows + a_gather[m % a_rows]; rows past M repeat the last one
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int m = min(m0 + sg.a_row[h][q], p.M - 1);
        const int bq = m / p.a_rows;
        sg.a_row[h][q] = bq * p.a_phys_rows + p.a_gather[m - bq * p.a_rows];
      }
    sa.init(p.A, (size_t)(p.M / p.a_rows) * p.a_phys_rows, p.K, sg.a_row, sg.lc, kt0);
  } else {
    sa.init(p.A + (size_t)m0 * p.K, (size_t)(p.M - m0), p.K, sg.a_row, sg.lc, kt0);
  }
  sb.init(p.W + (size_t)n0 * p.K, (size_t)(p.N - n0), p.K, sg.b_row, sg.lc, kt0);

  f32x16 acc[2][2][2];
  g8::mainloop<E>(lds, sa, sb, nk, acc);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3;
  using Ep = GemmEpilogue<E, EPI, 2>;
  Ep ep;
  ep.init(p, reinterpret_cast<float*>(lds) + wave * Ep::SLAB_FLOATS, n0 + wc * 64, lane);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      ep.tile(p, m0 + wr * 128 + i * 64 + mi * 32, ksl, [&](int j, int r) { return acc[i][mi][j][r]; });
}

template <typename E, int EPI>
int launch_gemm8p(const GemmParams& p_in, hipStream_t st) {
  GemmParams p = p_in;
  p.ksplit = p.ksplit < 1 ? 1 : p.ksplit;
  auto kern = gemm8p_kernel<E, EPI>;
  static LdsOptIn opt;            // > 64 KiB of LDS needs the opt-in attribute (once per instantiation and device)
  if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), g8::LDS_BYTES)) return rc;
  dim3 grid(ceil_div(p.M, g8::BM) * ceil_div(p.N, g8::BN) * p.ksplit), block(512);
  hipLaunchKernelGGL(kern, grid, block, g8::LDS_BYTES, st, p);
  KVQ_CHECK_LAUNCH("gemm8p_kernel");
  return KVQ_OK;
}

#define KVQ_INST8P(EPI)                                                             \
  template int launch_gemm8p<Fp16, EPI>(const GemmParams&, hipStream_t);            \
  template int launch_gemm8p<Bf16, EPI>(const GemmParams&, hipStream_t);
KVQ_INST8P(KVQ_EPI_BIAS_BF16)
KVQ_INST8P(KVQ_EPI_GELU_BF16)
KVQ_INST8P(KVQ_EPI_QKV_BF16)
KVQ_INST8P(KVQ_EPI_RESID_F32)
KVQ_INST8P(KVQ_EPI_STORE_F32)
KVQ_INST8P(KVQ_EPI_RELU_BF16)
KVQ_INST8P(KVQ_EPI_QGELU_BF16)

// Which shapes take the wide tile.  A 256 x 256 tile keeps a CU at 5-7 TFLOP/s when its K loop is long enough to amortise the
// 7-half-tile prologue and the epilogue; what it cannot do is fill the chip with few tiles.  KVQ_GEMM8P = 0: never, 1: whenever
// the shape is eligible (K % 64 == 0), unset: the measured rule below.
static int g_mode8p = -2;          // -2: not read yet; -1 auto; 0 never; 1 whenever eligible
static int mode8p() {
  if (g_mode8p == -2) g_mode8p = getenv("KVQ_GEMM8P") ? atoi(getenv("KVQ_GEMM8P")) : -1;
  return g_mode8p;
}
bool gemm8p_wanted(int M, int N, int K) {
  const int mode = mode8p();
  if (mode == 0 || K % 64 != 0 || K < 128) return false;
  if (mode == 1) return true;
  const long tiles = (long)ceil_div(M, g8::BM) * ceil_div(N, g8::BN);
  const double fill = (double)M * N / ((double)tiles * g8::BM * g8::BN);     // useful part of the tile grid
  // measured (profiles/r03_gemm_sweep*.txt): ahead of the ring kernel from ~120 tiles on (3136 x 3072 x 768, 156 tiles: 22.6 vs
  // 29.5 us; 3136 x 2304 x 768, 117 tiles: level), behind it below (84 tiles: 20.2 vs 19.0 us; 39-98 tiles of long K: 1.2-2x slower)
  return tiles >= 128 && fill >= 0.8 && K >= 256;
}

}  // namespace kvq

extern "C" int kvq_gemm_tile_mode(int mode) {
  const int prev = kvq::mode8p();
  if (mode >= -1 && mode <= 1) kvq::g_mode8p = mode;
  return prev;
}
