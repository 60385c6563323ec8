// What the two GEMM main loops (gemm.hip: 128 x 128 x 32 slices, 4 waves; gemm256.hip: 256 x 256 x 64 eight-phase, 8 waves)
// share: the launch parameters and the fused epilogue (bias, exact GELU / QuickGELU / ReLU + identity, the q scale + head
// split of swin_backbone.py:252-260, window_reverse + roll-back + crop + residual of :472-488,509,514, split-K partial tiles).
#pragma once
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
  const uint16_t* resid_h;     // KVQ_EPI_RELU_BF16: optional 16-bit [M][N] identity branch
  const float* resid_f32;      // KVQ_EPI_RELU_BF16: optional fp32 [M][N] identity branch (residual stream kept in fp32)
  unsigned long long* trace;   // diagnostic: per-block s_memtime stamps (kvq_debug_gemm_trace), else NULL
  int trace_blocks;
  // IMPL (implicit-GEMM convolution): A is the channels-last 16-bit activation (B, D, H, W, Cin) itself; GEMM row m is the
  // output pixel (b, do, ho, wo), K index ((kd*KH + kh)*KW + kw)*Cin + c, zero-padded to K.  taps[K/8]: per 8-channel chunk
  // of K {kd, kh, kw, element offset ((kd*H + kh)*W + kw)*Cin + c0}, offset < 0 = a chunk of the K padding.
  // taps == NULL (C % 32 == 0, the full kd x kh x kw tap set): the four chunks of a 32-deep slice belong to ONE tap, the same for
  // every lane — the kernel walks (kd, kh, kw, c) with wave-uniform counters (scalar ALU) and reads no table.
  const int4* taps;
  int cD, cH, cW, cC, cDo, cHo, cWo, csd, csh, csw, cpd, cph, cpw, ckd, ckh, ckw;
  // split-K: the grid holds ksplit copies of the tile grid; copy s multiplies k-slices [s nk / ksplit, (s + 1) nk / ksplit) and
  // stores its fp32 partial tile at out_f32 + s M N (KVQ_EPI_STORE_F32 instantiation, no bias); splitk_reduce_kernel finishes
  int ksplit;
  int ldc, col_off;            // 16-bit outputs (not QKV): row pitch / first column inside a wider destination (0 = N / 0)
  float* sk_ws;                // host side only: split-K scratch of the caller (NULL = never split) and its size
  size_t sk_bytes;
  const int32_t* a_gather;     // plain GEMM: row m reads A row (m / a_rows) * a_phys_rows + a_gather[m % a_rows]
  int a_rows, a_phys_rows;
};

// ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Each wave transposes its accumulators through a private LDS slab (32 rows x 32*NI fp32, reusing the operand ring: the K loop
// ended with a barrier) so that a lane owns CW CONSECUTIVE columns of one row: bias / activation run on float4s and every
// global store is a full dwordx4 (8 x 16-bit or 4 x fp32 per lane, >= 128 B contiguous per row).  Narrower stores are
// issue-bound: 16 dwordx2 per lane cost ~9k cycles of a 128x128 tile's life, twice the 8 dwordx4 that carry the same bytes.
template <typename E, int EPI, int NI>
struct GemmEpilogue {
  static constexpr bool OUT16 = EPI == KVQ_EPI_BIAS_BF16 || EPI == KVQ_EPI_GELU_BF16 || EPI == KVQ_EPI_RELU_BF16 ||
                                EPI == KVQ_EPI_QKV_BF16 || EPI == KVQ_EPI_QGELU_BF16;
  static constexpr int CW = OUT16 ? 8 : 4;                       // columns per lane (N % 8 == 0)
  static constexpr int SW = 32 * NI;                             // slab width (floats)
  static constexpr int CPRW = SW / CW;                           // lane chunks per slab row
  static constexpr int ROWS_PER_IT = 64 / CPRW;
  static constexpr int SLAB_FLOATS = 32 * SW;                    // per wave

  float* slab;
  int col_in, row_hi, ch, rsub, n;
  bool col_live;
  float bias[CW];
  int which, head, e0;
  float scale;

  // n_wave0: first output column of this wave's 32*NI-wide strip
  __device__ __forceinline__ void init(const GemmParams& p, float* slab_of_wave, int n_wave0, int lane) {
    slab = slab_of_wave;
    col_in = lane & 31; row_hi = (lane >> 5) * 4;
    ch = lane % CPRW; rsub = lane / CPRW;
    n = n_wave0 + ch * CW;
    col_live = n < p.N;                                          // N % 8 == 0: a chunk is wholly in or out
#pragma unroll
    for (int k = 0; k < CW; ++k) bias[k] = 0.f;
    if (p.bias && col_live) {
#pragma unroll
      for (int q = 0; q < CW / 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) bias[4 * q + k] = b4[k];
      }
    }
    which = 0; head = 0; e0 = 0; scale = 1.f;
    if (EPI == KVQ_EPI_QKV_BF16 && col_live) {                   // a 32-column tile = one head of q|k|v
      const int C = p.N / 3;
      which = n / C;
      head = (n % C) >> 5;
      e0 = n & 31;
      scale = which == 0 ? p.q_scale : 1.f;
    }
  }

  // one 32-row m-tile: get(j, r) = accumulator register r of the j-th 32-column tile; rows m_base + [0, 32)
  template <typename Get>
  __device__ __forceinline__ void tile(const GemmParams& p, int m_base, int ksl, Get get) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + row_hi) * SW + j * 32 + col_in] = get(j, r);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 / ROWS_PER_IT; ++it) {
      const int rl = it * ROWS_PER_IT + rsub;
      const int m = m_base + rl;
      float v[CW];
#pragma unroll
      for (int q = 0; q < CW / 4; ++q) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(slab + rl * SW + ch * CW + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 * q + k] = s4[k] + bias[4 * q + k];
      }
      if (m >= p.M || !col_live) continue;
      if (OUT16) {
        uint16_t* dst = p.out_h + (size_t)m * (p.ldc ? p.ldc : p.N) + p.col_off + n;
        if (EPI == KVQ_EPI_GELU_BF16) {
#pragma unroll
          for (int k = 0; k < CW; ++k) v[k] = gelu_fast(v[k]);
        } else if (EPI == KVQ_EPI_QGELU_BF16) {  // CLIP's QuickGELU: x * sigmoid(1.702 x) (clip/model.py:179-181)
#pragma unroll
          for (int k = 0; k < CW; ++k) v[k] = v[k] / (1.f + __expf(-1.702f * v[k]));
        } else if (EPI == KVQ_EPI_RELU_BF16) {   // conv + folded BN (+ identity) + ReLU (simpleVQA_model.py:104-124)
          if (p.resid_h) {
            const u32x4 rr = *reinterpret_cast<const u32x4*>(p.resid_h + (size_t)m * p.N + n);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[2 * k] += E::to_f32((uint16_t)(rr[k] & 0xffffu));
              v[2 * k + 1] += E::to_f32((uint16_t)(rr[k] >> 16));
            }
          }
          if (p.resid_f32) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 rr = *reinterpret_cast<const f32x4*>(p.resid_f32 + (size_t)m * p.N + n + 4 * q);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[4 * q + k] += rr[k];
            }
          }
#pragma unroll
          for (int k = 0; k < CW; ++k) v[k] = fmaxf(v[k], 0.f);
          if (p.out_f32) {                       // fp32 copy for the identity path
#pragma unroll
            for (int q = 0; q < 2; ++q)
              *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)m * p.N + n + 4 * q) =
                  (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
          }
        } else if (EPI == KVQ_EPI_QKV_BF16) {
#pragma unroll
          for (int k = 0; k < CW; ++k) v[k] *= scale;
          // with a row map (token -> window row, padded geometries: the GEMM runs over the tokens only) the head-major buffer has
          // out_rows rows per batch element; kvq_qkv_fill_pad writes the padding rows
          size_t mo = m, mtot = p.M;
          if (p.scatter_map) {
            const int bq = m / p.map_rows;
            mo = (size_t)bq * p.out_rows + p.scatter_map[m - bq * p.map_rows];
            mtot = (size_t)(p.M / p.map_rows) * p.out_rows;
          }
          dst = p.out_h + ((size_t)(which * p.num_heads + head) * mtot + mo) * 32 + e0;
        }
        const u32x4 o = {E::pack2(v[0], v[1]), E::pack2(v[2], v[3]), E::pack2(v[4 % CW], v[5 % CW]),
                         E::pack2(v[6 % CW], v[7 % CW])};
        *reinterpret_cast<u32x4*>(dst) = o;
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
      } else {  // KVQ_EPI_STORE_F32 (split-K: partial tile of K range ksl)
        *reinterpret_cast<f32x4*>(p.out_f32 + ((size_t)ksl * p.M + m) * p.N + n) = (f32x4){v[0], v[1], v[2], v[3]};
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
};

// gemm256.hip: the 256 x 256 x 64 eight-phase kernel behind the same parameters (a_gather and split-K included; no conv taps).
// The caller asks gemm8p_wanted() for the shapes that take it: K % 64 == 0, K >= 256, a tile grid that fills the chip.
template <typename E, int EPI>
int launch_gemm8p(const GemmParams& p, hipStream_t st);

}  // namespace kvq
