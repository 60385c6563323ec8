// One launch per residual block of SlowFast's FAST pathway (gfx950): conv_a (3x1x1, temporal) -> BN -> ReLU -> conv_b (1x3x3)
// -> BN -> ReLU -> conv_c (1x1x1) -> BN -> + shortcut (identity, or the block's 1x1x1 projection + BN) -> ReLU
// (pytorchvideo's ResBlock / BottleneckBlock as SlowFast_features.py:137-165 runs them; restated in oracle/slowfast_oracle.py).
//
// The fast pathway carries beta = 1/8 of the channels (8..32 inner, 32..128 outer) on 4x the frames: as separate implicit-GEMM
// launches its convolutions are launch-latency / HBM sized — 3-4 launches of 10-40 us per block that each read or write the
// block's full-resolution tensor, 1.0 ms of a 3.8 ms network for 8 % of its FLOPs (profiles/r02_slowfast_layers.txt).  Fused,
// a block reads its input once (three frames through L2) and writes its output once.
//
// Geometry: a workgroup (4 waves) owns a 14 x 14 output tile of ONE frame (56 / 28 / 14-pixel maps tile exactly); the stage's first
// block, whose conv_b and projection have stride 2, a 7 x 7 tile (its conv_a halo is 15 x 15 input pixels).  Everything is
// TOKEN-PER-LANE as in tail.hip: D[channel][pixel] = W[channel][k] . In[k][pixel] on v_mfma_f32_32x32x16 with the (BatchNorm-
// folded) weights as the A operand, fragment-major in LDS, and 32 pixels as the 32 columns, so a lane owns ONE pixel:
//   a: the 16 x 16 halo of the tile (8 column tiles, 2 per wave); B fragments are 16-byte global loads (8 channels of one
//      (frame, pixel) of the channels-last input); bias + ReLU in registers; the 16-bit result goes to LDS [256 pixels][CI]
//      (zero outside the image: conv_b's padding);
//   b: 7 column tiles of output pixels; a B fragment half = the 8 channels of one tap's neighbour pixel, one ds_read_b128;
//   c: conv_b's accumulators are packed straight into the B operand (k order = accumulator order, folded into the packed
//      weights), + the shortcut (16-byte loads of the lane's own pixel, the lane pair exchanging 8-byte pieces by v_permlane32_swap,
//      or one more MFMA over the input channels), ReLU, 16-byte stores through the same exchange.
// Rounding points are those of the unfused launches (16-bit a, b and block output; fp32 accumulation and shortcut add).
#include "common.hpp"

namespace kvq {

typedef __attribute__((address_space(3))) void* bn_lds_t;
typedef __attribute__((address_space(1))) const void* bn_gbl_t;

constexpr int BN_HALO = 16;
__host__ __device__ constexpr int bn_tile(int stride) { return stride == 1 ? 14 : 7; }     // outputs per tile edge: halo (T - 1) s + 3 <= 16

struct BneckParams {
  const uint16_t* x;
  uint16_t* out;
  const unsigned char* pack;
  int B, T, H, W, Ho, Wo, tiles_y, tiles_x;      // input map H x W, output map Ho x Wo (= ceil(H / stride))
};

__host__ __device__ constexpr int bn_ks(int k) { return (k + 15) / 16; }
// packed image (bytes): A fragments | B fragments | C fragments [RT][KSC] | S fragments [RT][KSS] (projection only) | fp32
// bias_a[32] bias_b[32] bias_c[COUT]  — every part a whole number of 1 KB wave-loads
template <int CIN, int CI, int COUT, bool SC>
struct BnLayout {
  static constexpr int KSA = bn_ks(3 * CIN), KSB = bn_ks(9 * CI), KSC = bn_ks(CI), KSS = SC ? bn_ks(CIN) : 0, RT = COUT / 32;
  static constexpr int OFF_A = 0, OFF_B = OFF_A + KSA * 1024, OFF_C = OFF_B + KSB * 1024, OFF_S = OFF_C + RT * KSC * 1024;
  static constexpr int OFF_BIAS = OFF_S + RT * KSS * 1024;
  static constexpr int BIAS_BYTES = ((64 + COUT) * 4 + 1023) / 1024 * 1024;
  static constexpr int PACK_BYTES = OFF_BIAS + BIAS_BYTES;
  static constexpr int OFF_TILE = PACK_BYTES;                           // [256 halo pixels][CI] 16-bit
  static constexpr int LDS_BYTES = OFF_TILE + BN_HALO * BN_HALO * CI * 2;
};

template <typename E, int CIN, int CI, int COUT, bool SC, int STRIDE, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void fast_bottleneck_kernel(BneckParams p) {
  static_assert(NW == 4 || NW == 8, "4 waves (two conv_a column tiles each) or 8 (one each)");
  static_assert(STRIDE == 1 || (STRIDE == 2 && SC), "a strided block has a projection shortcut");
  constexpr int BN_T = bn_tile(STRIDE);
  fp16_saturate_mode();
  using L = BnLayout<CIN, CI, COUT, SC>;
  using V8 = typename E::v8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = p.tiles_y * p.tiles_x;
  const int bt = blockIdx.x / ntile, tile = blockIdx.x - bt * ntile;
  const int b = bt / p.T, t = bt - b * p.T;
  const int y0 = (tile / p.tiles_x) * BN_T, x0 = (tile % p.tiles_x) * BN_T;          // output coordinates
  const size_t frame = (size_t)p.H * p.W, oframe = (size_t)p.Ho * p.Wo;
  const uint16_t* xb = p.x + (size_t)b * p.T * frame * CIN;

  // weights + biases -> LDS (LDS-DMA, 1 KB per wave-load)
  for (int q = wave; q < L::PACK_BYTES / 1024; q += NW)
    __builtin_amdgcn_global_load_lds((bn_gbl_t)(p.pack + q * 1024 + lane * 16), (bn_lds_t)(lds + q * 1024), 16, 0, 0);
  const float* s_ba = reinterpret_cast<const float*>(lds + L::OFF_BIAS);
  const float* s_bb = s_ba + 32;
  const float* s_bc = s_ba + 64;
  uint16_t* a_tile = reinterpret_cast<uint16_t*>(lds + L::OFF_TILE);

  // ---- conv_a on the halo: 8 column tiles, TPW per wave.  4 waves: the loads of BOTH tiles are in flight before the first wait
  // where the registers allow it (<= 12 k-steps).  8 waves (one workgroup per CU, 256 registers per wave): one tile each - for the
  // 14 x 14 maps of res4, where a launch is ONE tile per CU and its time is the workgroup's serial chain, not throughput ----------
  constexpr int TPW = 8 / NW;
  constexpr int NLOAD = (TPW == 2 && L::KSA <= 12) ? 2 : 1;
  V8 bx[NLOAD][L::KSA];
  auto load_tile = [&](int c2, V8 (&dst)[L::KSA], bool& inside, int& ap) {
    ap = (TPW * wave + c2) * 32 + j;
    const int yy = STRIDE * y0 - 1 + (ap >> 4), xx = STRIDE * x0 - 1 + (ap & 15);
    inside = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    const uint16_t* px = xb + ((size_t)(inside ? yy : 0) * p.W + (inside ? xx : 0)) * CIN;
#pragma unroll
    for (int s = 0; s < L::KSA; ++s) {
      const int k0 = 16 * s + 8 * h;
      const int dt = k0 / CIN, c = k0 - dt * CIN, tt = t + dt - 1;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k0 < 3 * CIN && inside && tt >= 0 && tt < p.T) v = *reinterpret_cast<const u32x4*>(px + (size_t)tt * frame * CIN + c);
      dst[s] = __builtin_bit_cast(V8, v);
    }
  };
  bool inside[2];
  int ap[2];
  load_tile(0, bx[0], inside[0], ap[0]);
  if (NLOAD == 2) load_tile(1, bx[1], inside[1], ap[1]);
  // the weight image has landed and is visible to all
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int c2 = 0; c2 < TPW; ++c2) {
    if (NLOAD == 1 && c2 == 1) load_tile(1, bx[0], inside[1], ap[1]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < L::KSA; ++s)
      acc = E::mfma32(*reinterpret_cast<const V8*>(lds + L::OFF_A + s * 1024 + lane * 16), bx[NLOAD == 2 ? c2 : 0][s], acc);
#pragma unroll
    for (int q = 0; q < CI / 8; ++q) {
      const f32x4 ba = *reinterpret_cast<const f32x4*>(s_ba + 8 * q + 4 * h);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = inside[c2] ? fmaxf(acc[4 * q + e] + ba[e], 0.f) : 0.f;
      *reinterpret_cast<u32x2*>(a_tile + ap[c2] * CI + 8 * q + 4 * h) = (u32x2){E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
    }
  }
  __syncthreads();

  // ---- conv_b, conv_c, shortcut on the 14 x 14 outputs: column tiles wave, wave + 4 ---------------------------------------
  int n_ct = (BN_T * BN_T + 31) / 32;
  if (NW == 8) asm volatile("" : "+s"(n_ct));      // opaque trip count: with 8 waves the loop has one trip, and peeled into straight-line code the
                                      // scheduler hoisted every LDS weight read above the shortcut loads and spilled 1 KB per lane
#pragma unroll 1
  for (int ct = wave; ct < n_ct; ct += NW) {      // (stride 2: 49 outputs = 2 column tiles, waves 2 and 3 are done)
    const int op_raw = ct * 32 + j;
    const int op = op_raw < BN_T * BN_T ? op_raw : BN_T * BN_T - 1;
    const int oy = op / BN_T, ox = op - oy * BN_T;
    const int yy = y0 + oy, xx = x0 + ox;
    const bool live = op_raw < BN_T * BN_T && yy < p.Ho && xx < p.Wo;
    const size_t pix = ((size_t)t * p.H + (live ? STRIDE * yy : 0)) * p.W + (live ? STRIDE * xx : 0);       // input pixel under it
    const size_t opix = ((size_t)t * p.Ho + (live ? yy : 0)) * p.Wo + (live ? xx : 0);
    // shortcut operands first: their latency hides under conv_b
    V8 sx[SC ? L::KSS : 1];
    u32x2 idn[SC ? 1 : L::RT * 4];
    if (SC) {
#pragma unroll
      for (int s = 0; s < L::KSS; ++s) {
        const int k0 = 16 * s + 8 * h;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (k0 < CIN && live) v = *reinterpret_cast<const u32x4*>(xb + pix * CIN + k0);
        sx[s] = __builtin_bit_cast(V8, v);
      }
    } else {
      // 16 bytes per lane (channels 8 (2 u + h) .. + 7 of the lane's pixel), then the lane pair of a pixel exchanges 8-byte pieces
      // by v_permlane32_swap into the accumulator layout (4 + 4 channels of every 8): half the pixel-divergent load instructions
#pragma unroll
      for (int u = 0; u < L::RT * 2; ++u) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (live) v = *reinterpret_cast<const u32x4*>(xb + pix * CIN + 8 * (2 * u + h));       // CIN == COUT
        const auto s0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
        idn[2 * u] = (u32x2){s0[0], s1[0]};             // q = 2 u:     channels 8 q + 4 h .. + 3
        idn[2 * u + 1] = (u32x2){s0[1], s1[1]};         // q = 2 u + 1
      }
    }
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int s = 0; s < L::KSB; ++s) {
      const int k0 = 16 * s + 8 * h;
      const int tap = k0 / CI, ci0 = k0 - tap * CI, dy = tap / 3, dx = tap - 3 * dy;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k0 < 9 * CI) v = *reinterpret_cast<const u32x4*>(a_tile + ((STRIDE * oy + dy) * BN_HALO + STRIDE * ox + dx) * CI + ci0);
      accb = E::mfma32(*reinterpret_cast<const V8*>(lds + L::OFF_B + s * 1024 + lane * 16), __builtin_bit_cast(V8, v), accb);
    }
    // bias + ReLU; accumulator order IS the k order of the packed conv_c weights
    V8 hb[L::KSC];
#pragma unroll
    for (int s = 0; s < L::KSC; ++s) {
      u32x4 w;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bb + 16 * s + 8 * q + 4 * h);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(accb[8 * s + 4 * q + e] + bb[e], 0.f);
        w[2 * q] = E::pack2(v[0], v[1]);
        w[2 * q + 1] = E::pack2(v[2], v[3]);
      }
      hb[s] = __builtin_bit_cast(V8, w);
    }
    uint16_t* orow = p.out + ((size_t)b * p.T * oframe + opix) * COUT;
#pragma unroll
    for (int rt = 0; rt < L::RT; ++rt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < L::KSC; ++s)
        acc = E::mfma32(*reinterpret_cast<const V8*>(lds + L::OFF_C + (rt * L::KSC + s) * 1024 + lane * 16), hb[s], acc);
      if (SC) {
#pragma unroll
        for (int s = 0; s < L::KSS; ++s)
          acc = E::mfma32(*reinterpret_cast<const V8*>(lds + L::OFF_S + (rt * L::KSS + s) * 1024 + lane * 16), sx[s], acc);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint32_t pk[2][2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int q = 2 * u + w;
          const f32x4 bc = *reinterpret_cast<const f32x4*>(s_bc + 32 * rt + 8 * q + 4 * h);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e] + bc[e];
          if (!SC) {
            const u32x2 r2 = idn[rt * 4 + q];
            v[0] += E::to_f32((uint16_t)(r2[0] & 0xffffu));
            v[1] += E::to_f32((uint16_t)(r2[0] >> 16));
            v[2] += E::to_f32((uint16_t)(r2[1] & 0xffffu));
            v[3] += E::to_f32((uint16_t)(r2[1] >> 16));
          }
          pk[w][0] = E::pack2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
          pk[w][1] = E::pack2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
        }
        // the pair's 8-byte pieces of (q, q + 1) -> 16 bytes per lane: lane h stores channels 8 (2 u + h) .. + 7
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        if (live) *reinterpret_cast<u32x4*>(orow + 32 * rt + 8 * (2 * u + h)) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
      }
    }
  }
}

template <typename E, int CIN, int CI, int COUT, bool SC, int STRIDE, int NW>
static int launch_bneck(const BneckParams& p, hipStream_t st) {
  using L = BnLayout<CIN, CI, COUT, SC>;
  auto k = fast_bottleneck_kernel<E, CIN, CI, COUT, SC, STRIDE, NW>;
  static LdsOptIn opt;
  if (int rc = opt.ensure(reinterpret_cast<const void*>(k), L::LDS_BYTES)) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)((long)p.B * p.T * p.tiles_y * p.tiles_x)), dim3(NW * 64), L::LDS_BYTES, st, p);
  KVQ_CHECK_LAUNCH("fast_bottleneck_kernel");
  return KVQ_OK;
}

template <int CIN, int CI, int COUT, bool SC, int STRIDE = 1, int NW = 4>
static int launch_bneck_dt(const BneckParams& p, int dtype, hipStream_t st) {
  return dtype == KVQ_DT_FP16 ? launch_bneck<Fp16, CIN, CI, COUT, SC, STRIDE, NW>(p, st) : launch_bneck<Bf16, CIN, CI, COUT, SC, STRIDE, NW>(p, st);
}

// the supported (input, inner, output) channel triples: SlowFast-R50's fast pathway, res2 .. res4
static int bneck_variant(int cin, int ci, int cout, int proj, int stride) {
  if (cin == 8 && ci == 8 && cout == 32 && proj && stride == 1) return 1;
  if (cin == 32 && ci == 8 && cout == 32 && !proj && stride == 1) return 2;
  if (cin == 64 && ci == 16 && cout == 64 && !proj && stride == 1) return 3;
  if (cin == 128 && ci == 32 && cout == 128 && !proj && stride == 1) return 4;
  if (cin == 32 && ci == 16 && cout == 64 && proj && stride == 2) return 5;
  if (cin == 64 && ci == 32 && cout == 128 && proj && stride == 2) return 6;
  return 0;
}

}  // namespace kvq

extern "C" size_t kvq_fast_bottleneck_pack_bytes(int cin, int ci, int cout, int projection, int stride) {
  using namespace kvq;
  switch (bneck_variant(cin, ci, cout, projection, stride)) {
    case 1: return BnLayout<8, 8, 32, true>::PACK_BYTES;
    case 2: return BnLayout<32, 8, 32, false>::PACK_BYTES;
    case 3: return BnLayout<64, 16, 64, false>::PACK_BYTES;
    case 4: return BnLayout<128, 32, 128, false>::PACK_BYTES;
    case 5: return BnLayout<32, 16, 64, true>::PACK_BYTES;
    case 6: return BnLayout<64, 32, 128, true>::PACK_BYTES;
    default: return 0;
  }
}

extern "C" int kvq_fast_bottleneck(const uint16_t* x, const int32_t dims4[4], int cin, int ci, int cout, int projection, int stride,
                                   const void* pack, int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && dims4 && pack && out, KVQ_ERR_NULL, "kvq_fast_bottleneck: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_fast_bottleneck: dtype %d", dtype);
  const int B = dims4[0], T = dims4[1], H = dims4[2], W = dims4[3];
  KVQ_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && (long)B * T * ceil_div(H, 7) * ceil_div(W, 7) < (1L << 31),
              KVQ_ERR_SHAPE, "kvq_fast_bottleneck: bad shape (%d,%d,%d,%d) stride %d", B, T, H, W, stride);
  const int var = bneck_variant(cin, ci, cout, projection, stride);
  KVQ_REQUIRE(var, KVQ_ERR_UNSUPPORTED, "kvq_fast_bottleneck: channels (%d -> %d -> %d, projection %d, stride %d) not in the built set", cin,
              ci, cout, projection, stride);
  KVQ_REQUIRE(((size_t)pack & 15) == 0 && ((size_t)x & 15) == 0 && ((size_t)out & 7) == 0, KVQ_ERR_SHAPE,
              "kvq_fast_bottleneck: x / pack must be 16-byte aligned, out 8-byte");
  const int Ho = ceil_div(H, stride), Wo = ceil_div(W, stride), tile = bn_tile(stride);      // (H + 2 - 3) / s + 1
  BneckParams p{x, out, (const unsigned char*)pack, B, T, H, W, Ho, Wo, ceil_div(Ho, tile), ceil_div(Wo, tile)};
  hipStream_t st = (hipStream_t)stream;
  switch (var) {
    case 1: return launch_bneck_dt<8, 8, 32, true>(p, dtype, st);
    case 2: return launch_bneck_dt<32, 8, 32, false>(p, dtype, st);
    case 3: return launch_bneck_dt<64, 16, 64, false>(p, dtype, st);
    case 4:      // res4: 14 x 14 maps, one tile per frame - few tiles: the 8-wave form (one conv_a tile per wave, b / c tiles in one round)
      return (long)p.B * p.T * p.tiles_y * p.tiles_x <= 512 ? launch_bneck_dt<128, 32, 128, false, 1, 8>(p, dtype, st)
                                                             : launch_bneck_dt<128, 32, 128, false>(p, dtype, st);
    case 5: return launch_bneck_dt<32, 16, 64, true, 2>(p, dtype, st);
    default: return launch_bneck_dt<64, 32, 128, true, 2>(p, dtype, st);
  }
}
