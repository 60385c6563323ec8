// Window attention with gated relative-position bias (GRPB) for head_dim 32 on gfx950: the EXACT per-score bias path.  The trunk's
// default is attn32.hip (the bias pre-built per (window type, head) as an fp16 image); this kernel serves the blocks whose tables exceed
// the image's range (max |bias| > 16) or whose images would not pay (models/backbones/swin_backbone.py::_set_dense_bias).
//
// Replaces WindowAttention3D.forward's core (swin_backbone.py:261-322): q@k^T, the bias-table
// gathers table[rpi] (:272-288), the fragment gate mix rpb*g + fpb*(1-g) with g = |dfrag| summed
// (:291-302), the 0/-100 shift mask add (:311-316), softmax and attn@v — without materialising
// any (nW,nH,N,N) or (nW,N,N,3) tensor: the bias is rebuilt per score from three small integers per
// token (a linear position code, packed fragment ids, a region id) and the per-head tables in LDS.
//
// One workgroup (4 waves, one per SIMD; two workgroups per CU) = one (window, head).  K (swizzled),
// V^T, the head's table and the window's token descriptors are staged in LDS once (~80 KB).  Waves
// pull 16-query tiles from an LDS ticket counter (25 tiles over 4 waves would otherwise leave a
// 7:6 imbalance) and compute the TRANSPOSED scores S^T = K * Q^T with v_mfma_f32_16x16x32 (one MFMA
// per 16x16 tile since head_dim == 32 == MFMA K), so a lane holds, for ONE query (lane & 15), 4 keys
// of every 16-key tile: the whole 392-long softmax row lives in 4 lanes' registers.
//   * the bias tile is built first and passed as the MFMA's C operand: S + bias costs nothing;
//   * row max = in-lane max chain + two wavefront shuffles;
//   * key order inside the tiles is permuted (tile t, MFMA row i <-> key 32*(t>>1)+8*(i>>2)+4*(t&1)+(i&3))
//     so that after exp + pack the probabilities ARE the A-operand fragment of the P*V MFMA
//     (lane group g holds keys 32s+8g..+7): no LDS round trip, no cross-lane permute for P;
//   * the row sum comes from a third P*V MFMA against an all-ones B operand: it sums exactly the
//     rounded probabilities that multiply V, and lands in the O layout (no shuffles to normalise).
// The kernel is VALU-bound (bias/mask/exp per score), so the per-score instruction count is what
// matters: gate = one v_sad_u8, bias = one fma on a (fpb, rpb-fpb) table, table address = one v_sub.
#include <stdlib.h>

#include "common.hpp"

namespace kvq {

constexpr int ATT_WAVES = 4;
constexpr int ATT_NT = 26;              // 16-key tiles -> up to 416 keys (N <= 400 supported, 392 used)
constexpr int ATT_KROWS = ATT_NT * 16;  // 416
constexpr int ATT_VPITCH = 400;         // 16-bit elems per V^T row: 800 B = 50 slots == 2 (mod 16): conflict-free b128
constexpr int ATT_VT_BYTES = 32 * ATT_VPITCH * 2 + 64;
constexpr int ATT_OFF_VT = ATT_KROWS * 64;
constexpr int ATT_OFF_TOK = ATT_OFF_VT + ATT_VT_BYTES;
// token descriptors are skewed by one 16-B entry per 8 keys (entry of key k at k + k/8): the four lane
// groups of a wave read keys 8 apart, which would otherwise sit exactly 128 B apart = on the same banks
constexpr int ATT_TOK_ENTRIES = ATT_KROWS + ATT_KROWS / 8;
constexpr int ATT_OFF_CTR = ATT_OFF_TOK + ATT_TOK_ENTRIES * 16;
constexpr int ATT_OFF_TAB = ATT_OFF_CTR + 16;

struct AttnParams {
  const uint16_t* qkv;
  const int32_t* tok;
  const float* rpb;
  const float* fpb;
  const float* pack;           // optional [nH][table_len][2] = {fpb|rpb, rpb-fpb}: coalesced staging
  int table_len, center, BW, nW, N, nH, use_mask;
  uint16_t* out;
  unsigned long long* trace;   // diagnostic stamps: [0] start, [1] staging done, [2] end
  int trace_blocks;
};

// LDS addressed by a plain 32-bit byte offset (address space 3): no generic-pointer base add per access
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef const __attribute__((address_space(3))) f32x2* lds_f2_t;

__device__ __forceinline__ int k_slot(int row, int g) { return row * 4 + (g ^ ((-(row >> 3)) & 3)); }

// FULL: N >= 384, i.e. only the last two 16-key tiles can contain keys >= N (the hot path has N = 392);
// the generic instantiation checks every tile.
template <typename E, bool GATED, bool MASK, bool FULL>
__global__ __launch_bounds__(ATT_WAVES * 64, 2) void window_attention_kernel(AttnParams p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Ks = reinterpret_cast<u32x4*>(smem);                               // [416*4] 16-B slots
  uint16_t* Vt = reinterpret_cast<uint16_t*>(smem + ATT_OFF_VT);            // [32][400] (+tail)
  int4* tokL = reinterpret_cast<int4*>(smem + ATT_OFF_TOK);                 // [416] {8*code, frag, region, -}
  int* ticket = reinterpret_cast<int*>(smem + ATT_OFF_CTR);
  float2* tab = reinterpret_cast<float2*>(smem + ATT_OFF_TAB);              // [table_len] {fpb|rpb, rpb-fpb}
  using V8 = typename E::v8;

  const int tid = threadIdx.x;
  const int unit = blockIdx.x;
  const bool tr = p.trace && tid == 0 && (int)blockIdx.x < p.trace_blocks;
  if (tr) p.trace[blockIdx.x * 8 + 0] = __builtin_readcyclecounter();
  const int bw = unit / p.nH, h = unit - bw * p.nH;
  const int N = p.N;
  const size_t Mtot = (size_t)p.BW * N;
  const int C = p.nH * 32;
  const uint16_t* Qg = p.qkv + ((size_t)(0 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
  const uint16_t* Kg = p.qkv + ((size_t)(1 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
  const uint16_t* Vg = p.qkv + ((size_t)(2 * p.nH + h) * Mtot + (size_t)bw * N) * 32;

  // ---- stage K (swizzled rows), V^T, token descriptors and this head's bias table ----
  for (int c = tid; c < ATT_KROWS * 4; c += ATT_WAVES * 64) {
    const int row = c >> 2, g = c & 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < N) v = *reinterpret_cast<const u32x4*>(Kg + (size_t)row * 32 + g * 8);
    Ks[k_slot(row, g)] = v;
  }
  // V^T: a thread transposes the 8-feature chunks of TWO adjacent keys and writes (key, key+1) pairs: 8 4-byte LDS
  // stores per 32 bytes instead of 16 2-byte ones
  for (int c = tid; c < ATT_VPITCH * 2; c += ATT_WAVES * 64) {
    const int key = (c >> 2) * 2, g = c & 3;
    u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
    if (key < N) v0 = *reinterpret_cast<const u32x4*>(Vg + (size_t)key * 32 + g * 8);
    if (key + 1 < N) v1 = *reinterpret_cast<const u32x4*>(Vg + (size_t)(key + 1) * 32 + g * 8);
    uint32_t* vt32 = reinterpret_cast<uint32_t*>(Vt);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vt32[((g * 8 + 2 * i) * ATT_VPITCH + key) >> 1] = (v0[i] & 0xffffu) | (v1[i] << 16);
      vt32[((g * 8 + 2 * i + 1) * ATT_VPITCH + key) >> 1] = (v0[i] >> 16) | (v1[i] & 0xffff0000u);
    }
  }
  if (tid < 32) Vt[32 * ATT_VPITCH + tid] = 0;   // tail read by the last K-step of row 31
  if (tid == 0) *ticket = 0;
  const int w = bw % p.nW;
  for (int n = tid; n < ATT_KROWS; n += ATT_WAVES * 64) {
    int4 t = make_int4(0, 0, 0, 0);
    if (n < N) {
      const int2 g2 = *reinterpret_cast<const int2*>(p.tok + ((size_t)w * N + n) * 2);
      t = make_int4(g2.x * 8, g2.y & 0xffff, (g2.y >> 16) & 0xff, 0);
    }
    tokL[n + (n >> 3)] = t;
  }
  if (p.pack) {                                   // host-packed table: one contiguous 8*table_len-byte copy
    const int tl2 = (p.table_len + 1) >> 1;       // heads are padded to an even entry count: 16-B aligned rows
    const f32x4* src = reinterpret_cast<const f32x4*>(p.pack) + (size_t)h * tl2;
    f32x4* dst = reinterpret_cast<f32x4*>(tab);
    for (int i = tid; i < tl2; i += ATT_WAVES * 64) dst[i] = src[i];
  } else
  for (int i = tid; i < p.table_len; i += ATT_WAVES * 64) {
    const float r = p.rpb[(size_t)i * p.nH + h];
    if (GATED) {
      const float f = p.fpb[(size_t)i * p.nH + h];
      tab[i] = make_float2(f, r - f);             // bias = f + g*(r-f)  (== r*g + f*(1-g) up to 1 ulp)
    } else {
      tab[i] = make_float2(r, 0.f);
    }
  }
  __syncthreads();
  if (tr) p.trace[blockIdx.x * 8 + 1] = __builtin_readcyclecounter();

  const int lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int nqt = (N + 15) >> 4;
  const float kLog2e = 1.4426950408889634f;
  // all-ones B operand of the row-sum MFMA (1.0 in fp16 = 0x3C00, in bf16 = 0x3F80)
  const uint32_t one2 = (uint32_t)E::cvt(1.0f) * 0x10001u;
  const V8 ones = __builtin_bit_cast(V8, (u32x4){one2, one2, one2, one2});

  while (true) {
    int qt = 0;
    if (lane == 0) qt = atomicAdd(ticket, 1);
    qt = __builtin_amdgcn_readfirstlane(qt);
    if (qt >= nqt) break;
    const int q0 = qt * 16;
    const int qrow = min(q0 + j, N - 1);
    // B operand of S^T = K Q^T: lane (j,g) holds Q[q0+j][8g..8g+7]
    const V8 qf = *reinterpret_cast<const V8*>(Qg + (size_t)qrow * 32 + g * 8);
    const int4 tq = tokL[qrow + (qrow >> 3)];
    const int cqb = tq.x + p.center * 8 + ATT_OFF_TAB;      // byte address of tab[cq + center - 0]
    const unsigned fq = (unsigned)tq.y;
    const int rq = tq.z;

    // Software pipeline over the 26 key tiles: the descriptor read (tile t+2) and the dependent table
    // gather (tile t+1) are issued before tile t's arithmetic, so two LDS round trips are in flight
    // behind ~60 VALU instructions instead of being waited for back to back.
#define ATT_KEY0(t) (32 * ((t) >> 1) + 4 * ((t) & 1))
#define ATT_TOKIDX(t, r) (ATT_KEY0(t) + (ATT_KEY0(t) >> 3) + (r))     /* + 9*g folded into tokg */
    const int4* tokg = tokL + 9 * g;
    int4 tkC[4], tkN[4];
    f32x2 tbC[4], tbN[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tkC[r] = tokg[ATT_TOKIDX(0, r)];
#pragma unroll
    for (int r = 0; r < 4; ++r) tkN[r] = tokg[ATT_TOKIDX(1, r)];
#pragma unroll
    for (int r = 0; r < 4; ++r) tbC[r] = *reinterpret_cast<lds_f2_t>((uintptr_t)(unsigned)(cqb - tkC[r].x));
    // K fragments are prefetched one tile ahead too: LDS returns in order, so a fragment read issued in the
    // same iteration as the gathers would make its s_waitcnt drain them all.
    const int krow0 = 8 * (j >> 2) + (j & 3);                  // + key0(t): MFMA row j <-> key (see header)
    V8 kfC = __builtin_bit_cast(V8, Ks[k_slot(ATT_KEY0(0) + krow0, g)]), kfN = kfC;
    f32x4 S[ATT_NT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < ATT_NT; ++t) {
      const int key0 = ATT_KEY0(t);                            // this lane's keys: key0 + 8g + r
      int4 tkNN[4];
      if (t + 1 < ATT_NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tbN[r] = *reinterpret_cast<lds_f2_t>((uintptr_t)(unsigned)(cqb - tkN[r].x));
      }
      if (t + 2 < ATT_NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tkNN[r] = tokg[ATT_TOKIDX(t + 2 < ATT_NT ? t + 2 : 0, r)];
      }
      if (t + 1 < ATT_NT) kfN = __builtin_bit_cast(V8, Ks[k_slot(ATT_KEY0(t + 1 < ATT_NT ? t + 1 : 0) + krow0, g)]);
      // ---- bias tile first, then S = K Q^T + bias (the bias rides in as the MFMA C operand) ----
      f32x4 b4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float bias = tbC[r][0];
        if (GATED) bias = fmaf((float)__builtin_amdgcn_sad_u8(fq, (unsigned)tkC[r].y, 0u), tbC[r][1], tbC[r][0]);
        // shift mask (compute_mask's -100, swin_backbone.py:583): a masked score sits ~100 below the row max, its
        // probability (< e^-87) is below fp32's normal range in the reference too — the -100 REPLACES the bias
        // term instead of being added to it (one select instead of select + add).
        if (MASK) bias = (tkC[r].z != rq) ? -100.0f : bias;
        b4[r] = bias;
      }
      S[t] = E::mfma16(kfC, qf, b4);   // consumed only after the loop: no MFMA-latency stall per tile
      // bound live ranges: without it the compiler hoists all 104 gathers ahead of their use and spills
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tkC[r] = tkN[r];
        tbC[r] = tbN[r];
        tkN[r] = tkNN[r];
      }
      kfC = kfN;
    }
#undef ATT_KEY0
#undef ATT_TOKIDX
#pragma unroll
    for (int t = 0; t < ATT_NT; ++t) {
      if (!FULL || t >= ATT_NT - 2) {   // tiles that can hold keys >= N: exclude them from the softmax
        const int key0 = 32 * (t >> 1) + 4 * (t & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[t][r] = (key0 + 8 * g + r) < N ? S[t][r] : -INFINITY;
      }
      mx = fmaxf(mx, fmaxf(fmaxf(S[t][0], S[t][1]), fmaxf(S[t][2], S[t][3])));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // ---- exp + pack: the packed pairs are the P*V A-fragments (p in [0,1]: no saturation needed) ----
    const float mb = mx * kLog2e;
    uint32_t P[ATT_NT][2];
#pragma unroll
    for (int t = 0; t < ATT_NT; ++t) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(S[t][r], kLog2e, -mb));
      P[t][0] = E::pack2_raw(e[0], e[1]);
      P[t][1] = E::pack2_raw(e[2], e[3]);
    }
    // ---- O = P V (two 16-wide feature tiles) and the row sums l = P 1, 13 K-steps of 32 keys ----
    f32x4 O0 = {0.f, 0.f, 0.f, 0.f}, O1 = {0.f, 0.f, 0.f, 0.f}, Ls = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ATT_NT / 2; ++s) {
      const u32x4 pa = {P[2 * s][0], P[2 * s][1], P[2 * s + 1][0], P[2 * s + 1][1]};
      const V8 pf = __builtin_bit_cast(V8, pa);
      const V8 v0 = *reinterpret_cast<const V8*>(Vt + j * ATT_VPITCH + 32 * s + 8 * g);
      const V8 v1 = *reinterpret_cast<const V8*>(Vt + (j + 16) * ATT_VPITCH + 32 * s + 8 * g);
      O0 = E::mfma16(pf, v0, O0);
      O1 = E::mfma16(pf, v1, O1);
      Ls = E::mfma16(pf, ones, Ls);
    }
    // ---- normalise + store.  O / Ls layout: col = feature j (+16), row = query 4g + r ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = 4 * g + r;
      const float inv = 1.0f / Ls[r];
      if (q0 + qq < N) {
        uint16_t* o = p.out + ((size_t)bw * N + q0 + qq) * C + h * 32 + j;
        o[0] = E::cvt(O0[r] * inv);
        o[16] = E::cvt(O1[r] * inv);
      }
    }
  }
  if (tr) {
    __builtin_amdgcn_s_waitcnt(0);
    p.trace[blockIdx.x * 8 + 2] = __builtin_readcyclecounter();
  }
}

template <typename E, bool GATED, bool MASK, bool FULL>
static int launch_attn2(const AttnParams& p, size_t lds, hipStream_t st) {
  auto kern = window_attention_kernel<E, GATED, MASK, FULL>;
  static LdsOptIn opt;            // per instantiation and device: opt in to > 64 KiB of dynamic LDS once
  if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), (int)lds)) return rc;
  dim3 grid((unsigned)(p.BW * p.nH)), block(ATT_WAVES * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, st, p);
  KVQ_CHECK_LAUNCH("window_attention_kernel");
  return KVQ_OK;
}

template <typename E, bool GATED, bool MASK>
static int launch_attn(const AttnParams& p, size_t lds, hipStream_t st) {
  return p.N >= 384 ? launch_attn2<E, GATED, MASK, true>(p, lds, st) : launch_attn2<E, GATED, MASK, false>(p, lds, st);
}

}  // namespace kvq

extern "C" int kvq_window_attention(const uint16_t* qkv, const int32_t* tok, const float* rpb, const float* fpb,
                                    const float* bias_pack, int table_len, int center, int BW, int nW, int N, int num_heads, int use_mask,
                                    int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(qkv && tok && rpb && out, KVQ_ERR_NULL, "kvq_window_attention: NULL pointer");
  KVQ_REQUIRE(BW > 0 && nW > 0 && BW % nW == 0 && num_heads > 0 && table_len > 0, KVQ_ERR_SHAPE,
              "kvq_window_attention: bad shape BW=%d nW=%d nH=%d table_len=%d", BW, nW, num_heads, table_len);
  KVQ_REQUIRE(N >= 1 && N <= 400, KVQ_ERR_UNSUPPORTED,
              "kvq_window_attention: window of %d tokens unsupported (1..400)", N);
  const size_t lds = (size_t)ATT_OFF_TAB + (size_t)((table_len + 1) & ~1) * 8;
  KVQ_REQUIRE(lds <= 80 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_window_attention: bias table of %d entries exceeds LDS",
              table_len);
  KVQ_REQUIRE(!bias_pack || ((size_t)bias_pack & 15) == 0, KVQ_ERR_SHAPE,
              "kvq_window_attention: bias_pack must be 16-byte aligned");
  AttnParams p{qkv, tok, rpb, fpb, bias_pack, table_len, center, BW, nW, N, num_heads, use_mask, out, g_trace, g_trace_blocks};
  hipStream_t st = (hipStream_t)stream;
  const bool gated = fpb != nullptr, mask = use_mask != 0;
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_window_attention: dtype %d", dtype);
  if (dtype == KVQ_DT_FP16) {
    if (gated && mask) return launch_attn<Fp16, true, true>(p, lds, st);
    if (gated) return launch_attn<Fp16, true, false>(p, lds, st);
    if (mask) return launch_attn<Fp16, false, true>(p, lds, st);
    return launch_attn<Fp16, false, false>(p, lds, st);
  }
  if (gated && mask) return launch_attn<Bf16, true, true>(p, lds, st);
  if (gated) return launch_attn<Bf16, true, false>(p, lds, st);
  if (mask) return launch_attn<Bf16, false, true>(p, lds, st);
  return launch_attn<Bf16, false, false>(p, lds, st);
}
