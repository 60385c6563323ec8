// One launch per IDENTITY residual block of SlowFast's SLOW pathway at res2 (gfx950): conv_a 1x1x1 (256 -> 64) + BN + ReLU -> conv_b
// 1x3x3 (64 -> 64, pad 0,1,1) + BN + ReLU -> conv_c 1x1x1 (64 -> 256) + BN -> + identity -> ReLU (pytorchvideo's ResBlock /
// BottleneckBlock as SlowFast_features.py:137-165 runs them; restated in oracle/slowfast_oracle.py).
//
// As three implicit-GEMM launches the block moves 412 MB per 8 clips (a: 103 in, 26 out; b: 26 + 26; c: 26 + 103 identity + 103 out)
// and conv_c alone is HBM-bound at 47-71 us; the three take 114-144 us (profiles/r03_slowfast_layers.txt).  Fused, the block reads its
// input once (plus the identity re-read, mostly from L2) and writes its output once.
//
// Geometry as csrc/bottleneck.hip: a workgroup (8 waves; one workgroup per CU: the weights take 138 KB of LDS) owns a 14 x 14 output tile
// of one frame; TOKEN-PER-LANE, D[channel][pixel] = W[channel][k] . In[k][pixel] on v_mfma_f32_32x32x16, weights as fragment-major A
// operands in LDS; 64 inner channels = two row tiles per k-step.
//   a: the 16 x 16 halo, one 32-pixel column tile per wave; B fragments are 16-byte global loads; the 16-bit result goes to LDS
//      [256 pixels][64] with the 16-byte chunk c of pixel p at chunk c ^ (p & 7) (zero outside the image: conv_b's padding);
//   b: the 196 outputs as 7 column tiles (the eighth wave only keeps the barriers); 36 k-steps (9 taps x 4), fragment sets requested
//      three k-steps ahead of their MFMAs;
//   c: conv_b's accumulators become the B operand by a 16-bit pack (k order folded into the packed weights); the conv_c weights arrive
//      by LDS-DMA over conv_a's (dead after phase a) while conv_b runs.  Identity and output move as 16 bytes per lane: the halves of a
//      32-lane pixel group exchange 8-byte pieces by v_permlane32_swap so a lane holds 16 consecutive channels, not 4 + 4 + 4 + 4.
// Measured, 8 clips (200 704 pixels): 83 us per block against 114 / 144 us for the three launches (tools/sneck_probe.py; 4 waves with
// 2 x 2 register blocking: 105 us).  Not the 45 us the byte count allows: a wave lives 39 k cycles for 4.4 k of MFMA, 54 % of it in
// s_waitcnt / barriers (profiles/r03_slowneck_pmc.txt) — one workgroup per CU leaves each tile's load -> a -> b -> c chain uncovered.
// Rounding points are those of the unfused launches (16-bit a, b and block output; fp32 accumulation and identity add).
#include "common.hpp"

namespace kvq {

typedef __attribute__((address_space(3))) void* sn_lds_t;
typedef __attribute__((address_space(1))) const void* sn_gbl_t;

constexpr int SN_CIN = 256, SN_CI = 64, SN_COUT = 256, SN_T = 14, SN_HALO = 16, SN_WAVES = 8;
constexpr int SN_KSA = SN_CIN / 16, SN_KSB = 9 * SN_CI / 16, SN_KSC = SN_CI / 16, SN_RTC = SN_COUT / 32;
// packed image: A [2][KSA] | B [2][KSB] | C [RTC][KSC] fragments of 1 KB | fp32 bias_a[64] bias_b[64] bias_c[256] (2 KB)
constexpr int SN_A_BYTES = 2 * SN_KSA * 1024, SN_B_BYTES = 2 * SN_KSB * 1024, SN_C_BYTES = SN_RTC * SN_KSC * 1024, SN_BIAS_BYTES = 2048;
constexpr int SN_PACK_BYTES = SN_A_BYTES + SN_B_BYTES + SN_C_BYTES + SN_BIAS_BYTES;
// LDS: [A, later C] | B | bias | a_tile
constexpr int SN_OFF_AC = 0, SN_OFF_B = SN_A_BYTES, SN_OFF_BIAS = SN_OFF_B + SN_B_BYTES, SN_OFF_TILE = SN_OFF_BIAS + SN_BIAS_BYTES;
constexpr int SN_LDS_BYTES = SN_OFF_TILE + SN_HALO * SN_HALO * SN_CI * 2;
static_assert(SN_A_BYTES == SN_C_BYTES, "conv_c's weights overlay conv_a's");

struct SlowneckParams {
  const uint16_t* x;       // (B, T, H, W, 256)
  uint16_t* out;           // (B, T, H, W, out_C): channels 0 .. 255 written
  const unsigned char* pack;
  int B, T, H, W, out_C, tiles_y, tiles_x;
};

// the two 8-byte pieces (q, q + 2) of a lane pair (h = 0 | 1) <-> one 16-byte piece per lane: d[q].upper <-> d[q + 2].lower
__device__ __forceinline__ void sn_swap(uint32_t& lo, uint32_t& hi) {
  const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
  lo = r[0];
  hi = r[1];
}

template <typename E>
__global__ __launch_bounds__(512, 1) void slow_bottleneck_kernel(SlowneckParams p) {
  fp16_saturate_mode();
  using V8 = typename E::v8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = p.tiles_y * p.tiles_x;
  const int bt = blockIdx.x / ntile, tile = blockIdx.x - bt * ntile;
  const int y0 = (tile / p.tiles_x) * SN_T, x0 = (tile % p.tiles_x) * SN_T;
  const size_t frame = (size_t)p.H * p.W;
  const uint16_t* xf = p.x + (size_t)bt * frame * SN_CIN;           // this frame
  uint16_t* of = p.out + (size_t)bt * frame * p.out_C;

  // conv_a / conv_b weights + biases -> LDS (LDS-DMA, 1 KB per wave-load)
  for (int q = wave; q < (SN_A_BYTES + SN_B_BYTES) / 1024; q += SN_WAVES)
    __builtin_amdgcn_global_load_lds((sn_gbl_t)(p.pack + q * 1024 + lane * 16), (sn_lds_t)(lds + q * 1024), 16, 0, 0);
  if (wave < SN_BIAS_BYTES / 1024)
    __builtin_amdgcn_global_load_lds((sn_gbl_t)(p.pack + SN_A_BYTES + SN_B_BYTES + SN_C_BYTES + wave * 1024 + lane * 16),
                                     (sn_lds_t)(lds + SN_OFF_BIAS + wave * 1024), 16, 0, 0);
  const float* s_ba = reinterpret_cast<const float*>(lds + SN_OFF_BIAS);
  const float* s_bb = s_ba + 64;
  const float* s_bc = s_ba + 128;
  unsigned char* a_tile = lds + SN_OFF_TILE;

  // ---- conv_a on the halo: column tile `wave` (all 16 fragment loads in flight before the first wait) ---------------------------------
  {
    V8 bx[SN_KSA];
    const int ap = wave * 32 + j;
    const int yy = y0 - 1 + (ap >> 4), xx = x0 - 1 + (ap & 15);
    const bool inside = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    const uint16_t* px = xf + ((size_t)(inside ? yy : 0) * p.W + (inside ? xx : 0)) * SN_CIN + 8 * h;
#pragma unroll
    for (int s = 0; s < SN_KSA; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (inside) v = *reinterpret_cast<const u32x4*>(px + 16 * s);
      bx[s] = __builtin_bit_cast(V8, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight image has landed ...
    __syncthreads();                                       // ... and is visible to all
    f32x16 acc[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < SN_KSA; ++s) {
      acc[0] = E::mfma32(*reinterpret_cast<const V8*>(lds + SN_OFF_AC + s * 1024 + lane * 16), bx[s], acc[0]);
      acc[1] = E::mfma32(*reinterpret_cast<const V8*>(lds + SN_OFF_AC + (SN_KSA + s) * 1024 + lane * 16), bx[s], acc[1]);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ba = *reinterpret_cast<const f32x4*>(s_ba + 32 * rt + 8 * q + 4 * h);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(acc[rt][4 * q + e] + ba[e], 0.f) : 0.f;
        *reinterpret_cast<u32x2*>(a_tile + ap * 128 + (((4 * rt + q) ^ (ap & 7)) << 4) + 8 * h) = (u32x2){E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
      }
  }
  __syncthreads();                                       // a_tile complete; conv_a's weights dead
  // conv_c's weights over conv_a's, under conv_b
  for (int q = wave; q < SN_C_BYTES / 1024; q += SN_WAVES)
    __builtin_amdgcn_global_load_lds((sn_gbl_t)(p.pack + SN_A_BYTES + SN_B_BYTES + q * 1024 + lane * 16),
                                     (sn_lds_t)(lds + SN_OFF_AC + q * 1024), 16, 0, 0);

  // ---- conv_b on the 14 x 14 outputs: column tile `wave` (7 tiles: the eighth wave only keeps the barriers) ----------------------------
  constexpr int NOUT = SN_T * SN_T;
  const bool has_tile = wave * 32 < NOUT;
  const int op_raw = wave * 32 + j;
  const int op = op_raw < NOUT ? op_raw : NOUT - 1;
  const int oy = op / SN_T, ox = op - oy * SN_T;
  const int hp = oy * SN_HALO + ox;                      // halo pixel under the tap (0, 0) of the lane's output pixel
  const bool live = op_raw < NOUT && y0 + oy < p.H && x0 + ox < p.W;
  const size_t opix = (size_t)(live ? y0 + oy : 0) * p.W + (live ? x0 + ox : 0);
  V8 hb[SN_KSC];
  if (has_tile) {
    f32x16 accb[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accb[rt][r] = 0.f;
    constexpr int DEPTH = 3;                               // fragment sets requested ahead of their MFMAs (2 MFMAs = 64 cycles per k-step)
    V8 fa[DEPTH + 1][2], fb[DEPTH + 1];
    auto rd = [&](int s, V8 (&a)[2], V8& b) {
      const int tap = s >> 2, dy = tap / 3, dx = tap - 3 * dy, chunk = 2 * (s & 3) + h;
      const int pi = hp + dy * SN_HALO + dx;
      b = *reinterpret_cast<const V8*>(a_tile + pi * 128 + ((chunk ^ (pi & 7)) << 4));
      a[0] = *reinterpret_cast<const V8*>(lds + SN_OFF_B + s * 1024 + lane * 16);
      a[1] = *reinterpret_cast<const V8*>(lds + SN_OFF_B + (SN_KSB + s) * 1024 + lane * 16);
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) rd(s, fa[s], fb[s]);
#pragma unroll
    for (int s = 0; s < SN_KSB; ++s) {
      if (s + DEPTH < SN_KSB) rd(s + DEPTH, fa[(s + DEPTH) % (DEPTH + 1)], fb[(s + DEPTH) % (DEPTH + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      accb[0] = E::mfma32(fa[s % (DEPTH + 1)][0], fb[s % (DEPTH + 1)], accb[0]);
      accb[1] = E::mfma32(fa[s % (DEPTH + 1)][1], fb[s % (DEPTH + 1)], accb[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // bias + ReLU; accumulator order IS the k order of the packed conv_c weights: k-step s = rows 16 s .. 16 s + 15 of conv_b's output
#pragma unroll
    for (int s = 0; s < SN_KSC; ++s) {
      u32x4 w;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bb + 16 * s + 8 * q + 4 * h);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(accb[s >> 1][8 * (s & 1) + 4 * q + e] + bb[e], 0.f);
        w[2 * q] = E::pack2(v[0], v[1]);
        w[2 * q + 1] = E::pack2(v[2], v[3]);
      }
      hb[s] = __builtin_bit_cast(V8, w);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // conv_c's weights have landed ...
  __syncthreads();                                       // ... for every wave
  if (!has_tile) return;

  // ---- conv_c + identity + ReLU: 8 row tiles of 32 channels; a lane moves channels 32 rt + 16 h .. + 15 of its pixel (two 16-byte pieces)
  auto load_id = [&](int rt, u32x4 (&d)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      d[i] = (u32x4){0u, 0u, 0u, 0u};
      if (live) d[i] = *reinterpret_cast<const u32x4*>(xf + opix * SN_CIN + 32 * rt + 16 * h + 8 * i);
    }
  };
  u32x4 idn[2][2];          // [buffer][piece]
  load_id(0, idn[0]);
#pragma unroll
  for (int rt = 0; rt < SN_RTC; ++rt) {
    if (rt + 1 < SN_RTC) load_id(rt + 1, idn[(rt + 1) & 1]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < SN_KSC; ++s)
      acc = E::mfma32(*reinterpret_cast<const V8*>(lds + SN_OFF_AC + (rt * SN_KSC + s) * 1024 + lane * 16), hb[s], acc);
    // identity pieces -> accumulator layout: d[q] = channels 8 q + 4 h .. + 3 (2 dwords)
    uint32_t d[4][2];
    const u32x4 i0 = idn[rt & 1][0], i1 = idn[rt & 1][1];
    d[0][0] = i0[0]; d[0][1] = i0[1]; d[2][0] = i0[2]; d[2][1] = i0[3];
    d[1][0] = i1[0]; d[1][1] = i1[1]; d[3][0] = i1[2]; d[3][1] = i1[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sn_swap(d[0][i], d[2][i]);
      sn_swap(d[1][i], d[3][i]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bc = *reinterpret_cast<const f32x4*>(s_bc + 32 * rt + 8 * q + 4 * h);
      float v[4];
      v[0] = acc[4 * q + 0] + bc[0] + E::to_f32((uint16_t)(d[q][0] & 0xffffu));
      v[1] = acc[4 * q + 1] + bc[1] + E::to_f32((uint16_t)(d[q][0] >> 16));
      v[2] = acc[4 * q + 2] + bc[2] + E::to_f32((uint16_t)(d[q][1] & 0xffffu));
      v[3] = acc[4 * q + 3] + bc[3] + E::to_f32((uint16_t)(d[q][1] >> 16));
      d[q][0] = E::pack2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
      d[q][1] = E::pack2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sn_swap(d[0][i], d[2][i]);
      sn_swap(d[1][i], d[3][i]);
    }
    if (live) {
      uint16_t* o = of + opix * p.out_C + 32 * rt + 16 * h;
      *reinterpret_cast<u32x4*>(o) = (u32x4){d[0][0], d[0][1], d[2][0], d[2][1]};
      *reinterpret_cast<u32x4*>(o + 8) = (u32x4){d[1][0], d[1][1], d[3][0], d[3][1]};
    }
  }
}

}  // namespace kvq

extern "C" size_t kvq_slow_bottleneck_pack_bytes(int cin, int ci, int cout) {
  return (cin == kvq::SN_CIN && ci == kvq::SN_CI && cout == kvq::SN_COUT) ? (size_t)kvq::SN_PACK_BYTES : 0;
}

extern "C" int kvq_slow_bottleneck(const uint16_t* x, const int32_t dims4[4], int cin, int ci, int cout, const void* pack, int dtype,
                                   uint16_t* out, int out_C, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && dims4 && pack && out, KVQ_ERR_NULL, "kvq_slow_bottleneck: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_slow_bottleneck: dtype %d", dtype);
  KVQ_REQUIRE(kvq_slow_bottleneck_pack_bytes(cin, ci, cout), KVQ_ERR_UNSUPPORTED, "kvq_slow_bottleneck: block (%d, %d, %d) is not built (256, 64, 256 only)",
              cin, ci, cout);
  SlowneckParams p{};
  p.x = x; p.out = out; p.pack = static_cast<const unsigned char*>(pack);
  p.B = dims4[0]; p.T = dims4[1]; p.H = dims4[2]; p.W = dims4[3]; p.out_C = out_C > 0 ? out_C : cout;
  KVQ_REQUIRE(p.B > 0 && p.T > 0 && p.H > 0 && p.W > 0 && p.out_C >= cout && p.out_C % 8 == 0, KVQ_ERR_SHAPE,
              "kvq_slow_bottleneck: dims (%d, %d, %d, %d), output row of %d channels", p.B, p.T, p.H, p.W, p.out_C);
  KVQ_REQUIRE(((size_t)x & 15) == 0 && ((size_t)out & 15) == 0 && ((size_t)pack & 15) == 0, KVQ_ERR_SHAPE, "kvq_slow_bottleneck: 16-byte aligned pointers");
  p.tiles_y = ceil_div(p.H, SN_T); p.tiles_x = ceil_div(p.W, SN_T);
  const long blocks = (long)p.B * p.T * p.tiles_y * p.tiles_x;
  KVQ_REQUIRE(blocks < (1L << 31), KVQ_ERR_UNSUPPORTED, "kvq_slow_bottleneck: %ld tiles", blocks);
  auto launch = [&](auto kern) -> int {
    static LdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), SN_LDS_BYTES)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SN_WAVES * 64), SN_LDS_BYTES, (hipStream_t)stream, p);
    return KVQ_OK;
  };
  const int rc = dtype == KVQ_DT_FP16 ? launch(slow_bottleneck_kernel<Fp16>) : launch(slow_bottleneck_kernel<Bf16>);
  if (rc) return rc;
  KVQ_CHECK_LAUNCH("slow_bottleneck_kernel");
  return KVQ_OK;
}
