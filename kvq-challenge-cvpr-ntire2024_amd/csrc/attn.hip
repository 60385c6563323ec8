// Window attention with gated relative-position bias (GRPB) for head_dim 32 on gfx950.
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

// ================================================================================================================
// Dense-bias variant.  The gather path above spends ~half of its VALU issue slots and most of its LDS traffic on
// REBUILDING the bias per score (descriptor reads, table gather, gate, mask select).  The bias of a (window, head)
// depends only on the block's tables and the window's position in the clip, not on the clip: it is built once per
// weight set by bias_dense_kernel and streamed from HBM/L2 (4 B per score) while the kernel is compute-bound.
//   * layout [window type][head][q-tile][key-tile][lane][4 x fp16] = the C operand of the score MFMA, narrowed: a
//     q-tile starts by loading its 26 bias tiles (26 independent 8-B loads in flight per lane, no LDS), widens them
//     into the score accumulators, and the MFMAs add K Q^T on top.  The launch is HBM-heavy on exactly this stream
//     (fp32 tiles: 1.2 GB per step, 410 MB of a stage-0 launch's 116 us), hence 2 B per score and one copy per
//     window TYPE (un-shifted windows that differ only in their depth index share a bias).  What is stored is
//     bias - max_key bias of the query's row: softmax is invariant to a per-row shift, and the shift puts the
//     entries that carry the probability mass next to 0, where fp16 resolves them to <= 2^-11 (the size of the
//     rounding of the probabilities themselves) whatever the magnitude of the tables;
//   * the -100 shift mask and the "key >= N" exclusion (-60000: exp2 underflows to exactly 0) are baked in, so
//     there is one instantiation per operand type instead of gated x masked x full;
//   * K and V are staged by LDS-DMA (no registers, no VALU; the gather kernel above transposes V through 2-byte LDS
//     stores: 18k of a 76k-tick stage-0 unit).  V stays row-major: score tiles cover the keys in natural order, so a lane
//     holds keys 16t+4g..+3 of tile t, and the PV step takes its V fragments through ds_read_b64_tr_b16 from
//     [32 keys][16 features] subtiles — the hardware transpose delivers exactly that k order.
namespace kvq {

constexpr float ATT_DENSE_OFF = -60000.0f;
constexpr int ATT_D_OFF_V = ATT_KROWS * 64;                       // K: 416 rows x 64 B
// the q-tile ticket lives in K row 415: rows 400..415 of K are never read (the 26th score tile cannot hold a key < N <= 400), and
// 2 x 26 KB is then the whole request — 16 bytes more would round up to the next LDS allocation granule and cost the third
// workgroup per CU if the granule is coarser than 16 B
constexpr int ATT_D_OFF_CTR = (ATT_KROWS - 1) * 64;
#ifndef ATT_D_OCC
#define ATT_D_OCC 3                 // workgroups per CU the register allocation is made for (1 / 2 / 3 per CU: 118 / 93 / 87 us at stage 0)
#endif
constexpr int ATT_D_LDS = ATT_D_OFF_V + ATT_NT * 1024;             // V: 13 key blocks x 2 feature halves x 1 KB; 53 248 B in all

struct DenseBuildParams {
  const int32_t* tok;
  const float* rpb;
  const float* fpb;
  int table_len, center, nW, N, nH, use_mask;
  uint16_t* out;
  unsigned* max_abs;         // optional: bit pattern of max |bias| over the real (un-masked, in-range) entries
  float* rowmax;             // unused (round 2's two-kernel builder kept the row maxima in global memory; the image's tail still reserves them)
};

// The builder: one workgroup per (window type, head).  The head's table pairs (f, r - f) — (r, 0) without a fragment table — and
// the window's token descriptors are staged in LDS once; pass 1 finds every query's row maximum over its un-masked keys, pass 2
// writes the tiles in the attention kernel's accumulator layout (8 bytes per lane, 512 contiguous bytes per wave and tile).  bias(q,
// key) is exactly the gather path's arithmetic: idx = code_q - code_k + center, b = fma(gate, r - f, f) with gate = the byte SAD of
// the fragment ids, the shift mask REPLACES it by -100.  (Round 2's pair of kernels evaluated every entry from global memory: 75 ms
// for Swin-B's 5.4 GiB at 64 x 256 x 256 — a start-up stall on every new clip geometry; this one is bound by writing the image.)
__global__ __launch_bounds__(256) void bias_dense_build_kernel(DenseBuildParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  f32x2* tab = reinterpret_cast<f32x2*>(bsm);                                   // [table_len] (f, r - f)
  int2* tokL = reinterpret_cast<int2*>(bsm + (size_t)p.table_len * 8);           // [N]
  float* rmax = reinterpret_cast<float*>(bsm + (size_t)p.table_len * 8 + (size_t)p.N * 8);      // [N]
  const int h = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, N = p.N;
  for (int i = tid; i < p.table_len; i += 256) {
    const float r = p.rpb[(size_t)i * p.nH + h];
    const float f = p.fpb ? p.fpb[(size_t)i * p.nH + h] : r;
    tab[i] = (f32x2){f, p.fpb ? r - f : 0.f};
  }
  for (int i = tid; i < N; i += 256) tokL[i] = *reinterpret_cast<const int2*>(p.tok + ((size_t)w * N + i) * 2);
  __syncthreads();
  auto value = [&](int2 tq, int key, bool* masked) -> float {
    const int2 tk = tokL[key];
    const f32x2 e = tab[tq.x - tk.x + p.center];
    const float gate = (float)__builtin_amdgcn_sad_u8((unsigned)(tq.y & 0xffff), (unsigned)(tk.y & 0xffff), 0u);
    *masked = p.use_mask && ((tq.y >> 16) & 0xff) != ((tk.y >> 16) & 0xff);
    return fmaf(gate, e[1], e[0]);
  };
  float big = 0.f;
  for (int q = tid; q < N; q += 256) {
    const int2 tq = tokL[q];
    float mx = -INFINITY;
    for (int key = 0; key < N; ++key) {
      bool masked;
      const float b = value(tq, key, &masked);
      if (!masked) { mx = fmaxf(mx, b); big = fmaxf(big, fabsf(b)); }
    }
    rmax[q] = mx;                        // the query itself is never masked: finite
  }
  if (p.max_abs) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = fmaxf(big, __shfl_xor(big, o));
    if ((tid & 63) == 0) atomicMax(p.max_abs, __float_as_uint(big));      // non-negative floats order like their bit patterns
  }
  __syncthreads();
  const int nqt = (N + 15) >> 4, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  uint16_t* img = p.out + ((size_t)w * p.nH + h) * nqt * ATT_NT * 256;
  for (int tile = wave; tile < nqt * ATT_NT; tile += 4) {
    const int qt = tile / ATT_NT, t = tile - qt * ATT_NT, q = 16 * qt + j;
    const int2 tq = tokL[q < N ? q : N - 1];
    const float shift = rmax[q < N ? q : N - 1];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * t + 4 * g + r;     // this lane's keys of score tile t: tiles cover keys in natural order
      float b = 0.f;
      if (key >= N) {
        b = ATT_DENSE_OFF;
      } else if (q < N) {
        bool masked;
        b = value(tq, key, &masked);
        if (masked) b = -100.0f;          // REPLACES the bias, as in the gather path
        b -= shift;
      }
      v[r] = b;
    }
    // fp16 whatever the operand type; un-masked entries are <= 0 after the row shift, the largest exactly 0
    *reinterpret_cast<u32x2*>(img + (size_t)tile * 256 + lane * 4) = (u32x2){Fp16::pack2_raw(v[0], v[1]), Fp16::pack2_raw(v[2], v[3])};
  }
}

typedef __attribute__((ext_vector_type(4))) short att_s4;
typedef __attribute__((address_space(3))) att_s4* att_tr_t;      // 8-B units: pointer arithmetic below is in fragments of 4
typedef __attribute__((address_space(3))) void* att_lds_t;
typedef __attribute__((address_space(1))) const void* att_gbl_t;

struct AttnDenseParams {
  const uint16_t* qkv;
  const u32x2* dense;
  int BW, nW, N, nH;
  int n_types;                 // distinct biases: window w uses type w % n_types
  int qsplit;                  // workgroups per (window, head, clip): each takes a contiguous share of the q-tiles
  uint16_t* out;
  unsigned long long* trace;   // -DKVQ_ATT_TRACE builds only
  int trace_blocks;
  const uint32_t* tile_skip;   // optional [nW]: bit t = q-tile t of the window holds padding rows only (its output is never read)
  int dsplit_from;             // >= 0: windows w >= dsplit_from of a clip are depth-split at token 196 of 392 (see tile_body); -1: none
  // fused qkv projection (x_ln != NULL): q | k | v of this (window, head) are computed in the prologue from the window's norm1 rows
  const uint16_t* x_ln;        // [BW*N][C] 16-bit, window order
  const uint16_t* w_qkv;       // [3C][C] 16-bit
  const float* b_qkv;          // [3C]
  int C_in;                    // = 32 nH
  float q_scale;
  uint16_t* q_out;             // = the q third of `qkv`: [nH][BW*N][32], written here and read back per q-tile
};

// Fused qkv projection (swin_backbone.py:252-260) for one (window, head): D^T[feature][row] = W[feature][:] . x[row][:] on
// v_mfma_f32_16x16x32 (weights = A operand, the window's norm1 rows = B operand), so a lane ends up with 4 CONSECUTIVE features of
// ONE row: + bias, (q: x head_dim^-0.5), 16-bit rounding as the qkv GEMM's epilogue does, then 8 bytes straight into the K image
// (XOR-swizzled rows), the V image ([32 keys][16 features] subtiles) or the q scratch.  Stages 0 / 1 (C = 96 / 192): the qkv GEMM
// there is an HBM-bound launch that writes 77-115 MB the attention launch reads right back; here the rows are read once per head
// (L2 hits) and q | k | v never exist in HBM (q: 25 KB per workgroup, L2-resident).  Wave w takes row tiles w, w+4, ...; all six
// 16-feature column tiles (q0 q1 k0 k1 v0 v1) in ONE pass over the rows (C <= 128: their 72-96 weight-fragment registers stay resident;
// the per-CU load path, 64 B/clk, is what bounds this prologue: at C = 192 the rows would be read twice and the launch loses to the GEMM).
template <typename E, int KS>
__device__ __forceinline__ void fused_qkv_prologue(const AttnDenseParams& p, unsigned char* smem, unsigned char* Vs, int bw, int h, int N) {
  using V8 = typename E::v8;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = KS * 32;
  const size_t Mtot = (size_t)p.BW * N;
  const uint16_t* xw = p.x_ln + (size_t)bw * N * C;
  uint16_t* qo = p.q_out + ((size_t)h * Mtot + (size_t)bw * N) * 32;
  constexpr int CP = KS <= 4 ? 6 : 3;            // column tiles per pass: their weight fragments (CP x KS x 4 registers) stay resident
#pragma unroll 1
  for (int pass = 0; pass < 6 / CP; ++pass) {
    V8 wf[CP][KS];
    float bias[CP][4];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const int ct = CP * pass + c, which = ct >> 1, half = ct & 1;
      const int frow = which * C + h * 32 + half * 16;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[c][ks] = *reinterpret_cast<const V8*>(p.w_qkv + (size_t)(frow + j) * C + 32 * ks + 8 * g);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b_qkv + frow + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[c][r] = b4[r];
    }
    // the row fragments of tile i+1 are requested before tile i is multiplied (an L2 round trip each: un-pipelined, seven
    // dependent round trips per pass were a fifth of the workgroup's life)
    const int nrt = (N + 15) / 16;
    V8 xn[KS];
    {
      const int rowc = min(16 * wave + j, N - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xn[ks] = *reinterpret_cast<const V8*>(xw + (size_t)rowc * C + 32 * ks + 8 * g);
    }
#pragma unroll 1
    for (int rt = wave; rt < nrt; rt += ATT_WAVES) {
      const int row = 16 * rt + j;
      V8 xf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = xn[ks];
      if (rt + ATT_WAVES < nrt) {
        const int rowc = min(row + 16 * ATT_WAVES, N - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xn[ks] = *reinterpret_cast<const V8*>(xw + (size_t)rowc * C + 32 * ks + 8 * g);
      }
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        const int ct = CP * pass + c, which = ct >> 1, half = ct & 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = E::mfma16(wf[c][ks], xf[ks], acc);
        const float sc = which == 0 ? p.q_scale : 1.f;
        const u32x2 v = {E::pack2((acc[0] + bias[c][0]) * sc, (acc[1] + bias[c][1]) * sc),
                         E::pack2((acc[2] + bias[c][2]) * sc, (acc[3] + bias[c][3]) * sc)};
        if (row < N) {                                    // rows N.. are the zero padding written by the caller
          if (which == 0) {
            *reinterpret_cast<u32x2*>(qo + (size_t)row * 32 + half * 16 + 4 * g) = v;
          } else if (which == 1) {
            *reinterpret_cast<u32x2*>(smem + k_slot(row, half * 2 + (g >> 1)) * 16 + (g & 1) * 8) = v;
          } else {
            *reinterpret_cast<u32x2*>(Vs + (2 * (row >> 5) + half) * 1024 + (row & 31) * 32 + 8 * g) = v;
          }
        }
      }
    }
  }
}


// FUSED: the qkv projection in the prologue; DSPLIT: depth-split windows take the half-range q-tile bodies (three bodies instead of
// one: compile-time variants, so that launches without such windows keep the single body's register allocation)
template <typename E, bool FUSED = false, bool DSPLIT = false>
__global__ __launch_bounds__(ATT_WAVES * 64, ATT_D_OCC) void window_attention_dense_kernel(AttnDenseParams p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Ks = reinterpret_cast<u32x4*>(smem);
  unsigned char* Vs = smem + ATT_D_OFF_V;     // row-major V as 13 x 2 subtiles of [32 keys][16 features] (1 KB each)
  int* ticket = reinterpret_cast<int*>(smem + ATT_D_OFF_CTR);
  using V8 = typename E::v8;

  // Block order: the nclip workgroups that share one (window, head) bias run on the SAME XCD (workgroup b -> XCD
  // b % 8, each XCD has its own L2) back to back, so the bias is fetched from HBM once per step, not once per clip.
  // Small grids (late stages: few windows) split a unit's q-tiles over qsplit workgroups, each staging K/V again.
  const int nclip = p.BW / p.nW, nrep = p.nW / p.n_types, npair = p.n_types * p.nH, per_pair = nclip * nrep * p.qsplit;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int pair, clip, rep, part;
  if (FUSED) {
    // fused qkv projection: the heads of one window read the same norm1 rows — they run back to back on ONE XCD (an XCD takes
    // whole window types: the three-to-six biases of a type and the rows of its windows share that L2), heads fastest
    const int per_type = p.nH * nclip * nrep, wt_ = (slot / per_type) * 8 + xcd, sub = slot % per_type;
    if (wt_ >= p.n_types) return;
    const int h_ = sub % p.nH, rest = sub / p.nH;
    pair = wt_ * p.nH + h_; clip = rest % nclip; rep = rest / nclip; part = 0;
  } else {
    const int sub = slot % per_pair;
    pair = (slot / per_pair) * 8 + xcd;                                    // pair = (window type, head): one bias
    clip = sub % nclip; rep = (sub / nclip) % nrep; part = sub / (nclip * nrep);
  }
  if (pair >= npair) return;
  const int wt = pair / p.nH, h = pair - wt * p.nH, w = rep * p.n_types + wt, bw = clip * p.nW + w;
  const int tid = threadIdx.x, N = p.N;
#ifdef KVQ_ATT_TRACE
  const bool tr = p.trace && tid == 0 && (int)blockIdx.x < p.trace_blocks;
  unsigned long long t_s = 0, t_x = 0, t_pv = 0, t_mark = 0, n_tiles = 0;
  if (tr) p.trace[blockIdx.x * 8 + 0] = __builtin_readcyclecounter();
#define ATT_MARK(acc) { const unsigned long long n_ = __builtin_readcyclecounter(); acc += n_ - t_mark; t_mark = n_; }
#else
#define ATT_MARK(acc)
#endif
  const size_t Mtot = (size_t)p.BW * N;
  const int C = p.nH * 32;
  const uint16_t* Qg = p.qkv + ((size_t)(0 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
  const uint16_t* Kg = p.qkv + ((size_t)(1 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
  const uint16_t* Vg = p.qkv + ((size_t)(2 * p.nH + h) * Mtot + (size_t)bw * N) * 32;

  // K and V go global -> LDS by LDS-DMA: no registers, no VALU.  The DMA writes lane-linear (16 B per lane behind a
  // wave-uniform base), so every lane picks the SOURCE chunk that belongs at its LDS position: K row-major with the
  // XOR swizzle of k_slot(); V row-major too — the PV step reads it through the hardware transpose (ds_read_b64_tr_b16),
  // which wants [32 keys][16 features] subtiles (32-B rows: the four lane groups of a read land on disjoint banks).
  if (FUSED) {           // q | k | v computed here (fused_qkv_prologue)
    fused_qkv_prologue<E, 3>(p, smem, Vs, bw, h, N);
  } else {
    const int lane_ = tid & 63, wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int it = wave_; it < ATT_KROWS * 4 / 64; it += ATT_WAVES) {
      const int c = it * 64 + lane_, row = c >> 2, gs = c & 3, g = gs ^ ((-(row >> 3)) & 3);
      if (row < N)
        __builtin_amdgcn_global_load_lds((att_gbl_t)(Kg + (size_t)row * 32 + g * 8), (att_lds_t)(smem + it * 1024), 16, 0, 0);
    }
    for (int it = wave_; it < ATT_KROWS * 4 / 64; it += ATT_WAVES) {
      const int key = 32 * (it >> 1) + (lane_ >> 1), feat = (it & 1) * 16 + (lane_ & 1) * 8;
      if (key < N)
        __builtin_amdgcn_global_load_lds((att_gbl_t)(Vg + (size_t)key * 32 + feat), (att_lds_t)(Vs + it * 1024), 16, 0, 0);
    }
  }
  {
    // keys N..415 exist only as padding (their bias is the -60000 of the image): finite zeros, never stale LDS
    for (int i = tid; i < (ATT_KROWS - N) * 8; i += ATT_WAVES * 64) {
      const int key = N + (i >> 3), q = i & 7;
      if (q < 4) { if (key < 16 * (ATT_NT - 1)) Ks[key * 4 + q] = (u32x4){0u, 0u, 0u, 0u}; }      // rows 400.. are never read (ticket)
      else *reinterpret_cast<u32x4*>(Vs + (2 * (key >> 5) + ((q >> 1) & 1)) * 1024 + (key & 31) * 32 + (q & 1) * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
  }
  const int nqt = (N + 15) >> 4;
  const int q_lo = part * nqt / p.qsplit, q_hi = (part + 1) * nqt / p.qsplit;
  if (tid == 0) *ticket = q_lo;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the DMA has landed
  __syncthreads();
#ifdef KVQ_ATT_TRACE
  t_mark = __builtin_readcyclecounter();
  if (tr) p.trace[blockIdx.x * 8 + 1] = t_mark;
#endif

  constexpr int NTD = ATT_NT - 1;      // score tiles that can hold a key < N (N <= 400): the 26th tile of the image is never live
  const int lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  // transpose-read source of this lane inside a V subtile: row 4g + j/4, features 4(j%4)..+3 (the hardware hands lane
  // (feature j, group g) rows 4g..4g+3 of column j: the B fragment of a 16-key half k-step)
  const att_tr_t vtr = (att_tr_t)(Vs + (4 * g + (j >> 2)) * 32 + (j & 3) * 8);
  const float kLog2e = 1.4426950408889634f;
  const uint32_t one2 = (uint32_t)E::cvt(1.0f) * 0x10001u;
  const V8 ones = __builtin_bit_cast(V8, (u32x4){one2, one2, one2, one2});
  const u32x2* dense = p.dense + (size_t)pair * nqt * ATT_NT * 64 + lane;

  // the ticket and the q fragment of the NEXT tile are fetched while this one is computed (both sit on the critical
  // path of a tile's first MFMA otherwise); B operand of S^T = K Q^T: lane (j, g) holds Q[q0+j][8g..8g+7]
  const uint32_t skip = p.tile_skip ? p.tile_skip[w] : 0u;      // wave-uniform
  auto take = [&]() -> int {
    int t_;
    do {                                                        // q-tiles of padding rows only are passed over
      t_ = 0;
      if (lane == 0) t_ = atomicAdd(ticket, 1);
      t_ = __builtin_amdgcn_readfirstlane(t_);
    } while (t_ < q_hi && ((skip >> t_) & 1u));
    return t_;
  };
  auto q_frag = [&](int t_) -> V8 { return *reinterpret_cast<const V8*>(Qg + (size_t)min(t_ * 16 + j, N - 1) * 32 + g * 8); };
  int qt = take();
  V8 qf = q_frag(qt);
  // One q-tile against the key tiles [T0, T1) (compile-time: the score registers are indexed statically).  [0, 25) is the
  // whole window.  Depth-split windows (shifted blocks, last window slab along D: the roll puts d = Dp-4.. and the wrapped
  // d = 0..3 into one window, the mask separates them, swin_backbone.py:563-579) only attend inside their own depth half —
  // the other half's scores are bias -100 and come out of the exponential as exact zeros — so a q-tile whose 16 queries sit in
  // one half skips the other half's key tiles (no bias fetch, no MFMA, no exp): [0, 13) or [12, 25) for the (8,7,7) window
  // split at token 196; the q-tile that straddles token 196 takes the whole range.  Bit-identical to the full range while the
  // row's logits spread by less than ~80 (the skipped scores must flush to zero in the full launch too; its row maximum could
  // otherwise come from the other half).
  auto tile_body = [&](auto t0_tag, auto t1_tag, const V8 qf_cur) __attribute__((always_inline)) {
    constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
    static_assert(T0 % 2 == 0 && T0 >= 0 && T1 <= NTD && T0 < T1, "key tiles pair up into 32-key PV steps");
    const int q0 = qt * 16;
    const u32x2* bd = dense + (size_t)qt * ATT_NT * 64;
    u32x2 braw[ATT_NT];
#pragma unroll
    for (int t = T0; t < T1; ++t) braw[t] = bd[t * 64];   // all bias tiles requested before anything waits
    f32x4 S[ATT_NT];
#pragma unroll
    for (int t = T0; t < T1; ++t)
      S[t] = (f32x4){Fp16::to_f32((uint16_t)(braw[t][0] & 0xffffu)), Fp16::to_f32((uint16_t)(braw[t][0] >> 16)),
                     Fp16::to_f32((uint16_t)(braw[t][1] & 0xffffu)), Fp16::to_f32((uint16_t)(braw[t][1] >> 16))};
    // score tile t = keys 16t..16t+15 in natural order: lane (query j, group g) then holds keys 16t+4g..+3, which is the
    // k order the transpose-read gives the V fragments
    V8 kfC = __builtin_bit_cast(V8, Ks[k_slot(16 * T0 + j, g)]), kfN = kfC;
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      if (t + 1 < T1) kfN = __builtin_bit_cast(V8, Ks[k_slot(16 * (t + 1 < T1 ? t + 1 : T0) + j, g)]);
      S[t] = E::mfma16(kfC, qf_cur, S[t]);
      kfC = kfN;
    }
    ATT_MARK(t_s);
    // row max: two chains of 3-input maxima (v_max3_f32: two new scores per instruction, 50 instead of 75 for the 100 scores)
    float mx = S[T0][0], mx1 = S[T0][2];
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      mx = fmaxf(fmaxf(mx, S[t][0]), S[t][1]);
      mx1 = fmaxf(fmaxf(mx1, S[t][2]), S[t][3]);
    }
    mx = fmaxf(mx, mx1);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // exponent arguments two at a time (v_pk_fma_f32: the kernel is VALU-issue bound, tools/ubench/pipe_share.hip)
    const f32x2 k2 = {kLog2e, kLog2e}, mb2 = {-mx * kLog2e, -mx * kLog2e};
    constexpr int S0 = T0 / 2, S1 = (T1 + 1) / 2;     // 32-key PV steps that hold a live tile
    uint32_t P[ATT_NT][2];
    if (T1 < 2 * S1) P[T1][0] = P[T1][1] = 0u;        // odd tile count: the step's second tile is skipped / padding keys 400..415
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      const f32x2 x0 = __builtin_elementwise_fma((f32x2){S[t][0], S[t][1]}, k2, mb2);
      const f32x2 x1 = __builtin_elementwise_fma((f32x2){S[t][2], S[t][3]}, k2, mb2);
      P[t][0] = E::pack2_raw(__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1]));
      P[t][1] = E::pack2_raw(__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1]));
    }
    ATT_MARK(t_x);
    // O^T = V^T P^T: the P registers are equally the B operand (lane (query j, g): keys 8g..8g+7 of the k-step) and the
    // transpose-read V fragment the A operand, so lane (query j, group g) ends up with features 4g..4g+3 (and 16+4g..) of ITS
    // query: 8-byte stores, one reciprocal per lane.  The ones-operand MFMA gives every lane its query's row sum.
    f32x4 O0 = {0.f, 0.f, 0.f, 0.f}, O1 = {0.f, 0.f, 0.f, 0.f}, Ls = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      const u32x4 pa = {P[2 * s][0], P[2 * s][1], P[2 * s + 1][0], P[2 * s + 1][1]};
      const V8 pf = __builtin_bit_cast(V8, pa);
      // k-step s = keys 32s..32s+31: elements 0-3 = keys 32s+4g+e (rows 0-15 of the subtile), 4-7 = keys 32s+16+4g+e
      const att_s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + (2 * s) * 128), a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + (2 * s) * 128 + 64);
      const att_s4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + (2 * s + 1) * 128), b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + (2 * s + 1) * 128 + 64);
      const V8 v0 = __builtin_bit_cast(V8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      const V8 v1 = __builtin_bit_cast(V8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      O0 = E::mfma16(v0, pf, O0);
      O1 = E::mfma16(v1, pf, O1);
      Ls = E::mfma16(ones, pf, Ls);
    }
    if (q0 + j < N) {
      const float inv = __builtin_amdgcn_rcpf(Ls[0]);       // the row sum is >= 1 (the row maximum contributes exp(0)): 1 ulp is plenty
      uint16_t* o = p.out + ((size_t)bw * N + q0 + j) * C + h * 32 + 4 * g;
      *reinterpret_cast<u32x2*>(o) = (u32x2){E::pack2(O0[0] * inv, O0[1] * inv), E::pack2(O0[2] * inv, O0[3] * inv)};
      *reinterpret_cast<u32x2*>(o + 16) = (u32x2){E::pack2(O1[0] * inv, O1[1] * inv), E::pack2(O1[2] * inv, O1[3] * inv)};
    }
    ATT_MARK(t_pv);
  };
  using TI0 = std::integral_constant<int, 0>;
  using TI12 = std::integral_constant<int, 12>;
  using TI13 = std::integral_constant<int, 13>;
  using TIN = std::integral_constant<int, NTD>;
  // depth-split window: host-checked geometry (N = 392, halves of 196 tokens): q-tiles 0..11 live in the first half, 13..24 in the
  // second, q-tile 12 (tokens 192..207) in both
  const bool dsplit = DSPLIT && p.dsplit_from >= 0 && w >= p.dsplit_from;         // wave-uniform
  while (qt < q_hi) {
    const int qt_next = take();
    const V8 qf_next = q_frag(qt_next);
    if (!DSPLIT || !dsplit || qt == 12) tile_body(TI0{}, TIN{}, qf);
    else if (qt < 12) tile_body(TI0{}, TI13{}, qf);
    else tile_body(TI12{}, TIN{}, qf);
#ifdef KVQ_ATT_TRACE
    ++n_tiles;
#endif
    qt = qt_next;
    qf = qf_next;
  }
#ifdef KVQ_ATT_TRACE
  if (tr) {
    __builtin_amdgcn_s_waitcnt(0);
    p.trace[blockIdx.x * 8 + 2] = __builtin_readcyclecounter();
    p.trace[blockIdx.x * 8 + 3] = t_s;
    p.trace[blockIdx.x * 8 + 4] = t_x;
    p.trace[blockIdx.x * 8 + 5] = t_pv;
    p.trace[blockIdx.x * 8 + 6] = n_tiles;
  }
#endif
}

template <typename E, bool FUSED, bool DSPLIT>
static int launch_attn_dense_v(const AttnDenseParams& p, hipStream_t st) {
  auto kern = window_attention_dense_kernel<E, FUSED, DSPLIT>;
  static LdsOptIn opt;
  constexpr int lds_req = ATT_D_LDS;
  if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), lds_req)) return rc;
  const int nclip = p.BW / p.nW, npair = p.n_types * p.nH;
  dim3 grid((unsigned)(8 * ceil_div(npair, 8) * nclip * (p.nW / p.n_types) * p.qsplit)), block(ATT_WAVES * 64);
  if (p.x_ln) grid.x = (unsigned)(8 * ceil_div(p.n_types, 8) * p.nH * nclip * (p.nW / p.n_types));     // XCDs take whole window types
  hipLaunchKernelGGL(kern, grid, block, lds_req, st, p);
  KVQ_CHECK_LAUNCH("window_attention_dense_kernel");
  return KVQ_OK;
}

template <typename E, bool FUSED = false>
static int launch_attn_dense(const AttnDenseParams& p, hipStream_t st) {
  return p.dsplit_from >= 0 ? launch_attn_dense_v<E, FUSED, true>(p, st) : launch_attn_dense_v<E, FUSED, false>(p, st);
}

}  // namespace kvq

static size_t dense_image_bytes(int n_types, int N, int num_heads) {
  return (size_t)n_types * num_heads * ((N + 15) / 16) * kvq::ATT_NT * 512;
}

extern "C" size_t kvq_attn_bias_dense_bytes(int n_types, int N, int num_heads) {
  if (n_types <= 0 || N < 1 || N > 400 || num_heads <= 0) return 0;
  // the image, then the builder's row maxima (fp32 [n_types][nH][N])
  return dense_image_bytes(n_types, N, num_heads) + (((size_t)n_types * num_heads * N * 4 + 255) & ~(size_t)255);
}

extern "C" int kvq_attn_bias_dense_build(const int32_t* tok, const float* rpb, const float* fpb, int table_len, int center,
                                         int nW, int N, int num_heads, int use_mask, void* out, float* max_abs, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(tok && rpb && out, KVQ_ERR_NULL, "kvq_attn_bias_dense_build: NULL pointer");
  KVQ_REQUIRE(kvq_attn_bias_dense_bytes(nW, N, num_heads) > 0 && table_len > 0, KVQ_ERR_SHAPE,
              "kvq_attn_bias_dense_build: bad shape nW=%d N=%d nH=%d", nW, N, num_heads);
  DenseBuildParams p{tok, rpb, fpb, table_len, center, nW, N, num_heads, use_mask, (uint16_t*)out, (unsigned*)max_abs, nullptr};
  const size_t lds = (size_t)table_len * 8 + (size_t)N * 12;
  KVQ_REQUIRE(lds <= 64 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_attn_bias_dense_build: table of %d entries does not fit the builder's LDS", table_len);
  hipLaunchKernelGGL(bias_dense_build_kernel, dim3((unsigned)num_heads, (unsigned)nW), dim3(256), lds, (hipStream_t)stream, p);
  KVQ_CHECK_LAUNCH("bias_dense_build_kernel");
  return KVQ_OK;
}

extern "C" int kvq_window_attention_dense(const uint16_t* qkv, const void* bias_dense, int n_types, int BW, int nW, int N,
                                          int num_heads, int dtype, uint16_t* out, void* stream) {
  return kvq_window_attention_dense_skip(qkv, bias_dense, n_types, BW, nW, N, num_heads, dtype, out, nullptr, stream);
}

extern "C" int kvq_window_attention_dense_skip(const uint16_t* qkv, const void* bias_dense, int n_types, int BW, int nW, int N,
                                               int num_heads, int dtype, uint16_t* out, const uint32_t* tile_skip, void* stream) {
  KvqAttnDenseArgs a{};
  a.qkv = qkv; a.bias_dense = bias_dense; a.n_types = n_types; a.BW = BW; a.nW = nW; a.N = N; a.num_heads = num_heads; a.dtype = dtype;
  a.out = out; a.tile_skip = tile_skip; a.dsplit_from = -1;
  return kvq_window_attention_dense_args(&a, stream);
}

extern "C" int kvq_window_attention_dense_args(const KvqAttnDenseArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->qkv && a->bias_dense && a->out, KVQ_ERR_NULL, "kvq_window_attention_dense: NULL pointer");
  const int BW = a->BW, nW = a->nW, N = a->N, num_heads = a->num_heads, n_types = a->n_types;
  KVQ_REQUIRE(BW > 0 && nW > 0 && BW % nW == 0 && num_heads > 0 && n_types > 0 && nW % n_types == 0, KVQ_ERR_SHAPE,
              "kvq_window_attention_dense: bad shape BW=%d nW=%d n_types=%d nH=%d", BW, nW, n_types, num_heads);
  KVQ_REQUIRE(N >= 1 && N <= 400, KVQ_ERR_UNSUPPORTED, "kvq_window_attention_dense: window of %d tokens unsupported (1..400)", N);
  KVQ_REQUIRE(((size_t)a->bias_dense & 7) == 0, KVQ_ERR_SHAPE, "kvq_window_attention_dense: bias_dense must be 8-byte aligned");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_window_attention_dense: dtype %d", a->dtype);
  KVQ_REQUIRE(a->dsplit_from < 0 || (N == 392 && a->dsplit_from < nW), KVQ_ERR_UNSUPPORTED,
              "kvq_window_attention_dense: depth-split windows need the (8,7,7) window (N = 392, halves of 196 tokens); got N=%d from=%d",
              N, a->dsplit_from);
  // 768 = 256 CUs x 3 resident workgroups: fill them when there are fewer (window, head, clip) units than that
  const int units = BW * num_heads, nqt = (N + 15) / 16;
  int qsplit = units >= 768 ? 1 : 768 / units;
  qsplit = qsplit > 4 ? 4 : qsplit;
  qsplit = qsplit > nqt ? nqt : qsplit;
  AttnDenseParams p{a->qkv, (const u32x2*)a->bias_dense, BW, nW, N, num_heads, n_types, qsplit, a->out, g_trace, g_trace_blocks, a->tile_skip,
                    a->dsplit_from < 0 ? -1 : a->dsplit_from};
  if (a->x_ln) {
    const int C = 32 * num_heads;
    KVQ_REQUIRE(a->w_qkv && a->b_qkv, KVQ_ERR_NULL, "kvq_window_attention_dense: x_ln without w_qkv / b_qkv");
    KVQ_REQUIRE(C == 96, KVQ_ERR_UNSUPPORTED, "kvq_window_attention_dense: the fused qkv projection is built for C = 96 (got %d)", C);
    p.qsplit = 1;      // one workgroup per (window, head) whatever the batch: which path a block takes (and with it the last bits of
                       // a clip's score) must not depend on how many clips share the launch
    KVQ_REQUIRE((((size_t)a->x_ln | (size_t)a->w_qkv | (size_t)a->b_qkv) & 15) == 0, KVQ_ERR_SHAPE, "kvq_window_attention_dense: x_ln / w_qkv / b_qkv must be 16-byte aligned");
    p.x_ln = a->x_ln; p.w_qkv = a->w_qkv; p.b_qkv = a->b_qkv; p.C_in = C; p.q_scale = a->q_scale;
    p.q_out = const_cast<uint16_t*>(a->qkv);
    return a->dtype == KVQ_DT_FP16 ? launch_attn_dense<Fp16, true>(p, (hipStream_t)stream) : launch_attn_dense<Bf16, true>(p, (hipStream_t)stream);
  }
  return a->dtype == KVQ_DT_FP16 ? launch_attn_dense<Fp16>(p, (hipStream_t)stream) : launch_attn_dense<Bf16>(p, (hipStream_t)stream);
}
