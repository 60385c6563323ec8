// LayerNorm over gathered rows (gfx950): HBM-bound, one row per lane group.
//
// Replaces, in one pass over the fp32 residual stream:
//   norm1 + F.pad + torch.roll(-shift) + window_partition   (swin_backbone.py:416-449)
//   norm2                                                   (:491)
//   PatchMerging's 4-neighbour concat + norm                (:546-552)
//   the final norm                                          (:1066-1068)
// The output row r of batch element b is LN(concat_p x[b*rows_in + map[r][p]]).  Statistics are
// two-pass in registers (mean, then centred sum of squares), fp32, biased variance, eps inside the
// rsqrt — the same formula torch's LayerNorm uses.  A row is owned by G = 16/32/64 lanes (picked
// so that every lane holds at least one float4); reductions are wavefront shuffles (xor butterflies);
// a lane group carries R consecutive rows with all their loads issued up front (memory-level parallelism).
#include "common.hpp"

namespace kvq {

struct LnParams {
  const float* x;
  const int32_t* map;
  int nparts, n_batch, rows_in, rows_out, Cin;
  const float* gamma;
  const float* beta;
  float eps;
  uint16_t* out_h;
  float* out_f32;
  int x16;          // round 6: x is an fp16 residual stream (rows of 2 Cin bytes behind the same pointer)
};

template <typename E, int G, int NV, int R>  // G lanes per row, NV float4 per lane (NV*G*4 >= C), R rows per group
__global__ __launch_bounds__(256) void layernorm_rows_kernel(LnParams p) {
  fp16_saturate_mode();
  const int C = p.nparts * p.Cin;
  const int lane_in = threadIdx.x % G;
  const long row0 = ((long)blockIdx.x * (256 / G) + threadIdx.x / G) * R;
  const long total = (long)p.n_batch * p.rows_out;

  // all R rows' loads are issued before any reduction: R*NV independent 16-B loads in flight per lane
  f32x4 v[R][NV];
  bool all_pad[R];
  long rowi[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    rowi[j] = row0 + j < total ? row0 + j : total - 1;          // clamp: every lane stays in the shuffles
    const int b = (int)(rowi[j] / p.rows_out), r = (int)(rowi[j] - (long)b * p.rows_out);
    all_pad[j] = (p.map != nullptr) && (p.nparts == 1) && (p.map[r] < 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * G + lane_in) * 4;
      v[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (c < C && !all_pad[j]) {
        const int part = c / p.Cin, cc = c - part * p.Cin;   // Cin % 4 == 0: a float4 never straddles parts
        const int s = p.map ? p.map[r * p.nparts + part] : r;
        if (s >= 0 && p.x16) {      // 8 bytes, kept RAW in the first two registers (widened below: a conversion here would wait for the load)
          const uint64_t u = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint16_t*>(p.x) + ((size_t)b * p.rows_in + s) * p.Cin + cc);
          v[j][i][0] = __uint_as_float((uint32_t)u); v[j][i][1] = __uint_as_float((uint32_t)(u >> 32));
        } else if (s >= 0) v[j][i] = *reinterpret_cast<const f32x4*>(p.x + ((size_t)b * p.rows_in + s) * p.Cin + cc);
      }
    }
  }
  if (p.x16) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const f16x4_t hv = __builtin_bit_cast(f16x4_t, (uint64_t)__float_as_uint(v[j][i][0]) | ((uint64_t)__float_as_uint(v[j][i][1]) << 32));
        v[j][i] = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};      // (zero rows: raw 0 -> 0.0)
      }
  }
  // gamma / beta of this lane's columns: requested with the rows (one latency, not one more behind the statistics), kept for all R rows
  constexpr bool HOIST = NV <= 3;      // (NV = 6, the 1536-wide merge rows: 48 more registers cost occupancy — 19.6 -> 35.6 us)
  f32x4 gmv[HOIST ? NV : 1], bev[HOIST ? NV : 1];
  if (HOIST) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * G + lane_in) * 4;
      gmv[i] = bev[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (c < C) {
        gmv[i] = *reinterpret_cast<const f32x4*>(p.gamma + c);
        bev[i] = *reinterpret_cast<const f32x4*>(p.beta + c);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += (v[j][i][0] + v[j][i][1]) + (v[j][i][2] + v[j][i][3]);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, G);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * G + lane_in) * 4;
      if (c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = v[j][i][k] - mean;
          sq += d * d;
        }
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, G);
    const float rstd = rsqrtf(sq / (float)C + p.eps);
    if (row0 + j >= total) continue;
    const long row = row0 + j;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * G + lane_in) * 4;
      if (c >= C) continue;
      f32x4 y;
      if (all_pad[j]) {
        y = (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
        const f32x4 gm = HOIST ? gmv[i] : *reinterpret_cast<const f32x4*>(p.gamma + c);     // wide rows: L1/L2-resident, reloaded per row
        const f32x4 be = HOIST ? bev[i] : *reinterpret_cast<const f32x4*>(p.beta + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = (v[j][i][k] - mean) * rstd * gm[k] + be[k];
      }
      if (p.out_h) {
        u32x2 o = {E::pack2(y[0], y[1]), E::pack2(y[2], y[3])};
        *reinterpret_cast<u32x2*>(p.out_h + (size_t)row * C + c) = o;
      } else {
        *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)row * C + c) = y;
      }
    }
  }
}

template <typename E, int G, int NV>
static int launch_ln_e(const LnParams& p, hipStream_t st) {
  constexpr int R = NV <= 1 ? 4 : (NV <= 3 ? 2 : 1);       // 4-6 float4 loads in flight per lane; more only adds VGPRs
  const long total = (long)p.n_batch * p.rows_out;
  const int rows_per_block = (256 / G) * R;
  dim3 grid((unsigned)((total + rows_per_block - 1) / rows_per_block)), block(256);
  hipLaunchKernelGGL((layernorm_rows_kernel<E, G, NV, R>), grid, block, 0, st, p);
  KVQ_CHECK_LAUNCH("layernorm_rows_kernel");
  return KVQ_OK;
}

template <int G, int NV>
static int launch_ln(const LnParams& p, int dtype, hipStream_t st) {
  return dtype == KVQ_DT_FP16 ? launch_ln_e<Fp16, G, NV>(p, st) : launch_ln_e<Bf16, G, NV>(p, st);
}

}  // namespace kvq

extern "C" int kvq_layernorm_rows(const float* x, const int32_t* map, int nparts, int n_batch, int rows_in,
                                  int rows_out, int Cin, const float* gamma, const float* beta, float eps,
                                  uint16_t* out_h, int dtype, float* out_f32, void* stream) {
  return kvq::layernorm_rows_stream(x, 0, map, nparts, n_batch, rows_in, rows_out, Cin, gamma, beta, eps, out_h, dtype, out_f32, stream);
}

// x_f16 != 0 (csrc/plan.hip only): x is an fp16 residual stream
int kvq::layernorm_rows_stream(const float* x, int x_f16, const int32_t* map, int nparts, int n_batch, int rows_in,
                               int rows_out, int Cin, const float* gamma, const float* beta, float eps,
                               uint16_t* out_h, int dtype, float* out_f32, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && gamma && beta, KVQ_ERR_NULL, "kvq_layernorm_rows: NULL input");
  KVQ_REQUIRE((out_h != nullptr) != (out_f32 != nullptr), KVQ_ERR_NULL,
              "kvq_layernorm_rows: exactly one of out_h/out_f32 must be set");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_layernorm_rows: dtype %d", dtype);
  KVQ_REQUIRE(nparts >= 1 && n_batch > 0 && rows_in > 0 && rows_out > 0 && Cin > 0 && Cin % 4 == 0,
              KVQ_ERR_SHAPE, "kvq_layernorm_rows: bad shape (nparts=%d n_batch=%d rows=%d/%d Cin=%d)", nparts,
              n_batch, rows_in, rows_out, Cin);
  KVQ_REQUIRE(map || (nparts == 1 && rows_in == rows_out), KVQ_ERR_SHAPE,
              "kvq_layernorm_rows: identity map needs nparts==1 and rows_in==rows_out");
  const int C = nparts * Cin;
  LnParams p{x, map, nparts, n_batch, rows_in, rows_out, Cin, gamma, beta, eps, out_h, out_f32, x_f16};
  hipStream_t st = (hipStream_t)stream;
  const int nvec = C / 4;
  // widths of the form 3 * 2^k (C = 96, 192, 384, 768, 1536) split exactly into G lanes x 3 (x 6) float4: no idle lanes
  if (nvec == 24) return launch_ln<8, 3>(p, dtype, st);
  if (nvec == 48) return launch_ln<16, 3>(p, dtype, st);
  if (nvec == 96) return launch_ln<32, 3>(p, dtype, st);
  if (nvec == 384) return launch_ln<64, 6>(p, dtype, st);
  if (nvec <= 16) return launch_ln<16, 1>(p, dtype, st);
  if (nvec <= 32) return launch_ln<32, 1>(p, dtype, st);
  if (nvec <= 64) return launch_ln<64, 1>(p, dtype, st);
  if (nvec <= 128) return launch_ln<64, 2>(p, dtype, st);
  if (nvec <= 192) return launch_ln<64, 3>(p, dtype, st);
  if (nvec <= 256) return launch_ln<64, 4>(p, dtype, st);
  if (nvec <= 512) return launch_ln<64, 8>(p, dtype, st);
  if (nvec <= 1024) return launch_ln<64, 16>(p, dtype, st);
  set_error("kvq_layernorm_rows: C=%d > 4096 unsupported", C);
  return KVQ_ERR_UNSUPPORTED;
}
