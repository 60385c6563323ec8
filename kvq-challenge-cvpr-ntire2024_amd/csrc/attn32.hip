// Window attention for head_dim 32 on gfx950 with the bias PRE-BUILT per (window type, head): 32 x 32 score blocks, running row maximum.
//
// Replaces WindowAttention3D.forward's core (swin_backbone.py:261-322).  The bias of a (window, head) depends only on the block's tables
// and the window's position in the clip, not on the clip: it is built once per weight set and plan geometry by bias32_build_kernel and
// streamed from HBM / L2 (2 B per score; one image per window TYPE: un-shifted windows that differ only in their depth index share one)
// while the kernel is bound by per-score VALU issue.  attn.hip keeps the exact per-score gather path (tables past max |bias| 16, or images
// that would dwarf the activations).  What the kernel is built around:
//   * S^T = K Q^T on v_mfma_f32_32x32x16 (two k-steps of 16 = head_dim 32): a lane holds 16 keys of ONE query (lane & 31) per 32-key
//     block — half the matrix-pipe issue time per flop of a 16 x 16 tiling and a quarter of its LDS fragment reads;
//   * the bias tile (fp16, row-max shifted: the entries that carry the probability mass sit next to 0, where fp16 resolves them to
//     <= 2^-11; the -100 shift mask and the "key >= N" exclusion, -60000, baked in) is widened, scaled by log2(e) and shifted by the row's
//     RUNNING maximum in one v_fma_mix_f32 per score — it is the MFMA's C operand, so the accumulator comes out as the exponent argument:
//     no subtract, no separate conversion.  q arrives scaled by head_dim^-0.5 * log2(e) (the qkv epilogue's scale): S is in log2 units;
//   * the running maximum is the flash-attention scheme with a deferred rescale: a block whose scores stay within 2^THR of the current
//     maximum is exponentiated as it is (P <= 2^THR: fp16 / bf16 hold that at full relative precision, and the normaliser is the sum of the
//     ROUNDED probabilities, so the scale cancels); only a growth past THR — or the row's first block — takes the rescale path (O, l and the
//     scores of the block move by the same 2^-d).  Masked scores (-100) and padding keys leave the exponential as zeros whatever the maximum;
//   * row sums by v_dot2c on the packed probabilities (one VALU per two scores); P V as O^T = V^T P^T with the packed P registers as the B
//     operand and V through the hardware transpose read (row-major V in LDS = a plain copy of the global rows);
//   * the key-block loop is software-pipelined inside the wave: the score MFMAs of block t+1 and the P V MFMAs of block t-1 are in the matrix
//     pipe while the VALU runs max / exp / pack of block t; bias tiles are requested two blocks ahead, the next q-block's ticket, q fragments
//     and first tile while the current one is computed.
// One workgroup (4 waves) per (window, head, clip[, q-part]), three per CU (52 KB of LDS): K (16-B chunks XOR-swizzled by (row >> 2) & 3)
// and V staged once by buffer-resource LDS-DMA (rows past N read zeros), 32-query blocks pulled from an LDS ticket.  FUSED (un-padded
// C = 96 stage): the workgroup computes its own q | k | v from the window's norm1 rows, k / v straight into the LDS images.
//
// Round 4 measured this body in three launch geometries (profiles/r04_attn_ab.txt): this one; a persistent form (one workgroup per CU, a
// loader wave streaming K | V into a three-slot LDS ring ahead of eight consumer waves) that wins alone on the chip with a warm image
// (71 vs 85-90 us at stage 0 un-fused) and loses inside the trunk, where the image is HBM-cold and the late stages hold few units; and the
// 16 x 16 tiling it replaces (attn.hip's dense kernel of rounds 2-3: 475 -> 449 us of attention per 4-clip step).
#include <stdlib.h>

#include "common.hpp"

#ifndef A32_ABL
#define A32_ABL 0           // diagnostic builds (tools/ubench/attn32_loop.hip), bit mask: 1 no bias stream, 2 no exponentials, 4 no P V MFMAs,
#endif                      // 8 no growth check, 16 no LDS fragment reads, 32 no v_fma_mix, 64 no row sums, 128 no row-maximum chain
#ifndef A32_SUM
#define A32_SUM 0           // row sums: 0 = v_dot2c on the packed probabilities, 1 = v_pk_add_f32 on the fp32 exponentials (round 5: loop bench 399 -> 392 cycles per block, standalone -1..3 %, 4-lane bench line unchanged; the error against the fp32 softmax grows 1.5x: not taken)
#endif
#if (A32_ABL & 2)
#define A32_EXP(x) ((x) * 0.001f + 1.0f)
#else
#define A32_EXP(x) __builtin_amdgcn_exp2f(x)
#endif

// timing probes of -DKVQ_DIAG builds (garbage scores; profiles/r06_traffic_probes.txt): compiled out of the product library
#ifdef KVQ_DIAG
#define A32_DIAG_FLAG(x) (x)
#else
#define A32_DIAG_FLAG(x) false
#endif

namespace kvq {

constexpr int A32_KB = 13;                         // 32-key blocks: 416 key positions (N <= 400 supported, 392 used)
constexpr int A32_ROWS = 400;                      // key rows the LDS images hold; rows 400..415 of the 13th block come from the zero block
constexpr int A32_K_BYTES = A32_ROWS * 64;         // 25 600: K rows (XOR-swizzled 16-B chunks), then as many of V (row-major)
constexpr int A32_SLOT = 2 * A32_K_BYTES;
constexpr int A32_WAVES_DEFAULT = 4;
constexpr int A32_OFF_ZERO = A32_SLOT;             // 1 KB of zeros, then the q-block ticket
constexpr int A32_OFF_CTR = A32_OFF_ZERO + 1024;
constexpr int A32_LDS = A32_OFF_CTR + 16;          // 52 240 B: three workgroups per CU
// The co-operative last q-block (round 6, SHORT launches: N = 385..392, 4 waves, no q-split).  13 q-blocks over 4 waves are 4 + 3 + 3 + 3: one
// wave works a fourth round on a block that holds 8 valid rows while three idle — and the workgroups of a launch run in lock-step, so every
// SIMD of the chip is down to one wave at once.  Instead the 12 full q-blocks go out 3 per wave and the LAST one is cut along the KEYS into four
// ranges (tickets 12..15: key blocks 0-3 | 4-6 | 7-9 | 10-12); the first three finishers leave their un-normalised (O, l, m) of the 8 rows
// in LDS, the fourth merges (the flash-attention merge: everything moves to the largest maximum) and stores.  The partials (1 KB each: 8 rows
// x 2 lane halves x 16 fp32) live where nothing that is READ FOR A USED VALUE lives: the zero block and the K image's rows 392..399 (in a SHORT
// launch the 13th key block only uses the accumulator rows of keys 384..391: what rows 392..415 hold reaches dead registers only; V's rows are
// NOT touched: P = 0 times a NaN would not be 0) and 1.5 KB behind the ticket.  52 240 + 1 744 B = 53 984: still three workgroups per CU.
constexpr int A32_COOP_PARTS = 4;
constexpr int A32_OFF_DONE = A32_OFF_CTR + 4;                          // arrivals of the key ranges
constexpr int A32_OFF_ML = A32_OFF_CTR + 16;                            // [3 slots][8 rows][nm, ls] fp32 = 192 B
constexpr int A32_OFF_PT = A32_OFF_ML + 192;                            // 3 x 512 B
constexpr int A32_LDS_COOP = A32_OFF_PT + 1536;                         // 53 984 B
// half `hh` (lane half) of partial slot `sl`: 8 rows x 64 B
__device__ __forceinline__ int a32_part_off(int sl, int hh) {
  const int u = 2 * sl + hh;                                            // 0, 1: the zero block; 2: K rows 392..399; 3..5: behind the ticket
  return u < 2 ? A32_OFF_ZERO + 512 * u : u == 2 ? 392 * 64 : A32_OFF_PT + 512 * (u - 3);
}
constexpr int A32_BDEPTH = 2;                      // bias tiles requested ahead of the block being multiplied (3 / 5 / 7 measured: no gain)
constexpr float A32_THR = 8.0f;                    // log2 units: a block is exponentiated against a maximum at most 2^8 too small
constexpr float A32_OFF = -60000.0f;               // padding keys: exp2 underflows to exactly 0

typedef __attribute__((ext_vector_type(4))) short a32_s4;
typedef __attribute__((address_space(3))) a32_s4* a32_tr_t;
typedef __attribute__((address_space(3))) void* a32_lds_t;

// A 32-query block keeps a row in lanes q and q + 32: v_permlane32_swap hands each half the other's value without an LDS round
// trip (new vdst = {own low half, partner's low half}, new vsrc = {partner's high half, own high half}: every lane sees both values)
__device__ __forceinline__ float a32_pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float a32_pair_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// One 32-query block against the key blocks [T0, T1) of the LDS images at `slot`.  Compile-time range: the score / probability /
// bias registers rotate through statically indexed sets.
// SHORT (round 6; N = 385..392, i.e. the (8,7,7) window): the 13th key block holds 8 valid keys, and they are registers 0..3 of both lane halves
// (key = 384 + (r & 3) + 8 (r >> 2) + 4 hi) — its bias widening, row maximum, exponentials, packs and row sums run on those four registers only
// (16 of the block's 56 VALU instructions; the MFMAs are whole tiles either way).  A KERNEL-level template argument: a launch holds one form of
// the body (round 5's run-time select between both forms in one kernel cost more in code size than the short block saved).
// The un-normalised state of a row after the key blocks [T0, T1): O (this lane's 16 features), the row sum of THIS lane's keys (the lane
// pair's halves still apart) and nm = -(the maximum the state is scaled by), log2 units.
struct A32State {
  f32x16 O;
  float ls, nm;
};

// normalise + store.  Lane (query q, half hi) holds features 8j + 4hi .. +3, j = 0..3: four 8-byte pieces of the row's 64 bytes,
// interleaved with the partner lane's.  Two v_permlane32_swap per dword pair hand lane q features 0..15 and lane q + 32 features
// 16..31: two 16-byte stores of 32 contiguous bytes per lane instead of four 8-byte ones (the store tail is issue-bound).
// `ls` = the ROW's sum (both halves).
template <typename E>
__device__ __forceinline__ void a32_finish(const f32x16& O, float ls, uint16_t* orow, const bool store) {
  const int hi = (threadIdx.x & 63) >> 5;
  const float inv = __builtin_amdgcn_rcpf(ls);          // >= 2^-THR-ish and finite: the row maximum contributes >= 2^-THR
  uint32_t pk[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    pk[j][0] = E::pack2(O[4 * j] * inv, O[4 * j + 1] * inv);
    pk[j][1] = E::pack2(O[4 * j + 2] * inv, O[4 * j + 3] * inv);
  }
  u32x4 w0, w1;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    // swap(vdst = piece j, vsrc = piece j + 2): low lanes end with {own piece j, partner's piece j}, high lanes with {partner's piece
    // j + 2, own piece j + 2} — in both cases consecutive features
    const auto a = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[2][d], false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(pk[1][d], pk[3][d], false, false);
    w0[d] = a[0]; w0[2 + d] = a[1];
    w1[d] = b[0]; w1[2 + d] = b[1];
  }
  if (store) {
    *reinterpret_cast<u32x4*>(orow + 16 * hi) = w0;
    *reinterpret_cast<u32x4*>(orow + 16 * hi + 8) = w1;
  }
}

template <typename E, int T0, int T1, bool SHORT = false, bool PARTIAL = false>
__device__ __forceinline__ void a32_qblock(const unsigned char* slot, const int zero, const u32x4* bd, const typename E::v8 qf0,
                                           const typename E::v8 qf1, const u32x4 (&pre)[2], uint16_t* orow, const bool store,
                                           A32State* part = nullptr) {
  using V8 = typename E::v8;
  static_assert(T0 >= 0 && T0 < T1 && T1 <= A32_KB, "key block range");
  const int lane = threadIdx.x & 63, q = lane & 31, hi = lane >> 5;
  const float kLog2e = 1.4426950408889634f;
  const int sw = (q >> 2) & 3;
  const u32x4* Ks = reinterpret_cast<const u32x4*>(slot);
  const int ka0 = q * 4 + (hi ^ sw), ka1 = q * 4 + ((2 + hi) ^ sw);          // + 128 per key block
  // the 13th block: rows 384..399 of the slot, then the zero block (`zero` = its offset from the slot, in 16-B units)
  const int kz0 = q < 16 ? ka0 + (A32_KB - 1) * 128 : zero + (q - 16) * 4 + (hi ^ sw), kz1 = q < 16 ? ka1 + (A32_KB - 1) * 128 : zero + (q - 16) * 4 + ((2 + hi) ^ sw);
  const a32_tr_t vtr = (a32_tr_t)(slot + A32_K_BYTES + (4 * hi + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
  const uint32_t one2 = (uint32_t)E::cvt(1.0f) * 0x10001u;

  constexpr int BR = A32_BDEPTH + 1;
  constexpr int LASTB = A32_KB - 1;
#define A32_NR(t) ((SHORT && (t) == LASTB) ? 4 : 16)      // live score registers of key block t
  u32x4 braw[BR][2];
  auto load_bias = [&](int t) __attribute__((always_inline)) {
#if (A32_ABL & 1)     // diagnostic: no bias stream (one tile, loaded once per q-block)
    if (t != T0) { braw[t % BR][0] = braw[T0 % BR][0]; braw[t % BR][1] = braw[T0 % BR][1]; return; }
#endif
    braw[t % BR][0] = bd[t * 128];
    if (A32_NR(t) > 8) braw[t % BR][1] = bd[t * 128 + 64];
  };
  // C operand of block t: bias * log2(e) - m, straight from the packed fp16 pairs (v_fma_mix_f32)
  auto mix = [&](int t, float nm, const f32x16& stale) __attribute__((always_inline)) -> f32x16 {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r >= A32_NR(t)) { c[r] = stale[r]; continue; }      // rows of padding keys, never read: whatever the destination registers hold (no moves)
      const uint32_t u = braw[t % BR][r >> 3][(r & 7) >> 1];
      const _Float16 hv = __builtin_bit_cast(_Float16, (uint16_t)((r & 1) ? (u >> 16) : (u & 0xffffu)));
      c[r] = (A32_ABL & 32) ? nm : __builtin_fmaf((float)hv, kLog2e, nm);
    }
    return c;
  };
  auto kfrag = [&](int t, int m) __attribute__((always_inline)) -> V8 {
    if (t == A32_KB - 1) return __builtin_bit_cast(V8, Ks[(m ? kz1 : kz0)]);          // key rows 400..415 come from the zero block
    return __builtin_bit_cast(V8, Ks[(m ? ka1 : ka0) + ((A32_ABL & 16) ? 0 : t) * 128]);
  };

  f32x16 S[2];
  uint32_t P[2][8];
  a32_s4 vf[2][4];                                                            // V fragments of block t: read during block t, used by P V one block later
  f32x16 O;
#pragma unroll
  for (int r = 0; r < 16; ++r) O[r] = 0.f;
  float ls = 0.f, ls1 = 0.f, nm = 0.f;                                        // nm = -(running maximum), log2 units
  f32x2 lv0 = {0.f, 0.f}, lv1 = {0.f, 0.f};                                   // A32_SUM == 1: the row sum as two packed fp32 chains

  braw[T0 % BR][0] = pre[0]; braw[T0 % BR][1] = pre[1];        // block T0: requested while the previous q-block was computed
#pragma unroll
  for (int a = 1; a < A32_BDEPTH; ++a)
    if (T0 + a < T1) load_bias(T0 + a);
  {
    const V8 k0 = kfrag(T0, 0), k1 = kfrag(T0, 1);
    const f32x16 c = mix(T0, 0.f, O);
    S[T0 & 1] = E::mfma32(k0, qf0, c);
    S[T0 & 1] = E::mfma32(k1, qf1, S[T0 & 1]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = T0; t <= T1; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    // ---- requests first: K fragments of block t+1, V fragments of block t, the bias of block t+2 ----
    V8 kn0, kn1;
    if (t + 1 < T1) { kn0 = kfrag(t + 1, 0); kn1 = kfrag(t + 1, 1); }
    if (t < T1 && !((A32_ABL & 16) && t > T0)) {
      vf[cur][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + t * 256);
      vf[cur][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + t * 256 + 64);
      if (t != A32_KB - 1) {                             // keys 400..415 are padding whatever N <= 400 is: their probabilities are zeros
        vf[cur][2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + t * 256 + 128);
        vf[cur][3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vtr + t * 256 + 192);
      }
    }
    if (t + A32_BDEPTH < T1) load_bias(t + A32_BDEPTH);
    // ---- P V of block t-1 in the matrix pipe under the row maximum of block t (two new scores per v_max3_f32) ----
    float mx = 0.f;
    if (t > T0 && !(A32_ABL & 4))
      O = E::mfma32(__builtin_bit_cast(V8, __builtin_shufflevector(vf[(A32_ABL & 16) ? (T0 & 1) : nxt][0], vf[(A32_ABL & 16) ? (T0 & 1) : nxt][1], 0, 1, 2, 3, 4, 5, 6, 7)),
                    __builtin_bit_cast(V8, (u32x4){P[nxt][0], P[nxt][1], P[nxt][2], P[nxt][3]}), O);
    if (t < T1) {
      mx = fmaxf(fmaxf(S[cur][0], S[cur][1]), S[cur][2]);
#pragma unroll
      for (int r = 3; r < ((A32_ABL & 128) ? 4 : A32_NR(t) - 1); r += 2) mx = fmaxf(fmaxf(mx, S[cur][r]), S[cur][r + 1]);
      mx = fmaxf(mx, S[cur][A32_NR(t) - 1]);
    }
    if (t > T0 && t - 1 != A32_KB - 1 && !(A32_ABL & 4))
      O = E::mfma32(__builtin_bit_cast(V8, __builtin_shufflevector(vf[(A32_ABL & 16) ? (T0 & 1) : nxt][2], vf[(A32_ABL & 16) ? (T0 & 1) : nxt][3], 0, 1, 2, 3, 4, 5, 6, 7)),
                    __builtin_bit_cast(V8, (u32x4){P[nxt][4], P[nxt][5], P[nxt][6], P[nxt][7]}), O);
    if (t == T1) break;
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (t == T0 || (!(A32_ABL & 8) && __builtin_amdgcn_ballot_w64(mx > A32_THR) != 0)) {
      // rescale path: the row's first block (exact maximum) or a growth past THR.  Everything still at the old maximum moves by
      // the same 2^-d: O and l (they hold blocks .. t-1 completely) and the scores of block t; block t+1's C operand is built below
      const float mr = a32_pair_max(mx);                            // the row lives in lanes q and q + 32
      const float d = (t == T0 || mr > A32_THR) ? mr : 0.f;
#pragma unroll
      for (int r = 0; r < A32_NR(t); ++r) S[cur][r] -= d;
      if (t != T0) {                       // the first block: O = l = 0, and 2^-d may be infinite (a masked row start)
        const float f = __builtin_amdgcn_exp2f(-d);
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] *= f;
        ls *= f;
        ls1 *= f;
        lv0 *= f;
        lv1 *= f;
      }
      nm -= d;
    }
    // ---- the score block t+1 in the matrix pipe under exp / pack / row sum of block t ----
    if (t + 1 < T1) {
      const f32x16 cn = mix(t + 1, nm, S[nxt]);
      S[nxt] = E::mfma32(kn0, qf0, cn);
    }
#if A32_SUM == 1
    // row sums on the fp32 exponentials, two per v_pk_add_f32, two chains (the v_dot2c form on the packed pairs cost 53 of the body's
    // 400 cycles per block in the loop bench).  The normaliser is then the sum of the UN-rounded probabilities: against the sum of the
    // rounded ones it differs by the mean of N independent 16-bit roundings, 2^-12 / sqrt(N) relative for fp16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 e = {A32_EXP(S[cur][2 * i]), A32_EXP(S[cur][2 * i + 1])};
      P[cur][i] = E::pack2_raw(e[0], e[1]);
      if (!(A32_ABL & 64) || i == 0) { if (i & 1) lv1 += e; else lv0 += e; }
    }
    if (t + 1 < T1) S[nxt] = E::mfma32(kn1, qf1, S[nxt]);
#pragma unroll
    for (int i = 4; i < 8; ++i) {
      const f32x2 e = {A32_EXP(S[cur][2 * i]), A32_EXP(S[cur][2 * i + 1])};
      P[cur][i] = E::pack2_raw(e[0], e[1]);
      if (!(A32_ABL & 64)) { if (i & 1) lv1 += e; else lv0 += e; }
    }
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) P[cur][i] = 2 * i < A32_NR(t) ? E::pack2_raw(A32_EXP(S[cur][2 * i]), A32_EXP(S[cur][2 * i + 1])) : 0u;
    if (t + 1 < T1) S[nxt] = E::mfma32(kn1, qf1, S[nxt]);
#pragma unroll
    for (int i = 4; i < 8; ++i) P[cur][i] = 2 * i < A32_NR(t) ? E::pack2_raw(A32_EXP(S[cur][2 * i]), A32_EXP(S[cur][2 * i + 1])) : 0u;
#pragma unroll
    for (int i = 0; i < ((A32_ABL & 64) ? 1 : A32_NR(t) / 2); i += 2) {          // two chains: v_dot2c accumulates in place
      ls = E::dot2(P[cur][i], one2, ls);
      ls1 = E::dot2(P[cur][i + 1], one2, ls1);
    }
#endif
    if (t + 1 < T1) {
      __builtin_amdgcn_sched_group_barrier(0x002, 16, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x002, 12, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  ls += ls1;
  if (A32_SUM == 1) ls = (lv0[0] + lv0[1]) + (lv1[0] + lv1[1]);
  ls = a32_pair_sum(ls);                                 // the row lives in lanes q and q + 32
  if (PARTIAL) {                                         // the key range of a co-operatively computed q-block: the caller merges the ranges
    part->O = O; part->ls = ls; part->nm = nm;
    return;
  }
  a32_finish<E>(O, ls, orow, store);
#undef A32_NR
}

struct Attn32Params {
  const uint16_t* qkv;         // [3][nH][BW*N][32], q scaled by head_dim^-0.5 * log2(e)
  const u32x4* image;          // [n_types*nH][ceil(N/32)][13][2][64 lanes] x 16 B (kvq_attn_bias32_build)
  int BW, nW, N, nH, n_types, qsplit;
  uint16_t* out;               // [BW*N][nH*32]
  const uint32_t* tile_skip;   // optional [nW]: bit t = rows 16t..16t+15 of the window are padding only
  int dsplit_from;             // >= 0: windows >= it are depth-split at token 196 of 392
  // fused qkv projection (x_ln != NULL)
  const uint16_t* x_ln;        // [BW*N][C] 16-bit, window order
  const uint16_t* w_qkv;       // [3C][C]
  const float* b_qkv;          // [3C]
  float q_scale;               // head_dim^-0.5 * log2(e)
  uint16_t* q_out;             // = the q third of qkv
  const uint32_t* pad_mask;    // optional [nW][13]: bit r & 31 of word r >> 5 = window row r is a padding row (k | v = b_qkv's thirds, q = 0)
  bool no_q_store;             // -DKVQ_DIAG timing probe (KVQ_NO_Q_STORE=1): the fused projection does not write q (garbage scores: the q scratch's write-back gone)
  bool image_one;              // -DKVQ_DIAG timing probe (KVQ_IMAGE_ONE_TYPE=1): every window reads the images of type 0 — garbage scores, the images' traffic gone
  bool coop_ok;                // no launch of this process splits a unit over workgroups (qsplit_max == 1): the co-operative last q-block may be chosen
};

// q | k | v of one (window, head) from the window's norm1 rows (C = 32 KS): D^T[feature][row] = W[feature][:] . x[row][:] on
// v_mfma_f32_16x16x32 — a lane ends up with 4 consecutive features of one row: + bias (q: x q_scale), the 16-bit rounding the qkv GEMM's
// epilogue applies, then 8 bytes into the K image (rows of 64 B, 16-B chunks XOR-swizzled by (row >> 2) & 3), the V image (plain rows) or
// the q scratch.  Wave w takes row tiles w, w + 4, ...; all six 16-feature column tiles (q0 q1 k0 k1 v0 v1) in one pass over the rows.
template <typename E, int KS, int NW>
__device__ __forceinline__ void a32_fused_qkv(const Attn32Params& p, unsigned char* slot, int bw, int h, int N) {
  using V8 = typename E::v8;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = KS * 32;
  const size_t Mtot = (size_t)p.BW * N;
  const uint16_t* xw = p.x_ln + (size_t)bw * N * C;
  uint16_t* qo = p.q_out + ((size_t)h * Mtot + (size_t)bw * N) * 32;
  V8 wf[6][KS];
  float bias[6][4];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int which = c >> 1, half = c & 1, frow = which * C + h * 32 + half * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[c][ks] = *reinterpret_cast<const V8*>(p.w_qkv + (size_t)(frow + j) * C + 32 * ks + 8 * g);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b_qkv + frow + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[c][r] = b4[r];
  }
  const int nrt = (N + 15) / 16;
  V8 xn[KS];
  {
    const int rowc = min(16 * wave + j, N - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xn[ks] = *reinterpret_cast<const V8*>(xw + (size_t)rowc * C + 32 * ks + 8 * g);
  }
#pragma unroll 1
  for (int rt = wave; rt < nrt; rt += NW) {
    const int row = 16 * rt + j;
    V8 xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = xn[ks];
    if (rt + NW < nrt) {                                   // the next tile's rows are requested before this one is multiplied
      const int rowc = min(row + 16 * NW, N - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xn[ks] = *reinterpret_cast<const V8*>(xw + (size_t)rowc * C + 32 * ks + 8 * g);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int which = c >> 1, half = c & 1;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = E::mfma16(wf[c][ks], xf[ks], acc);
      const float sc = which == 0 ? p.q_scale : 1.f;
      const u32x2 v = {E::pack2((acc[0] + bias[c][0]) * sc, (acc[1] + bias[c][1]) * sc),
                       E::pack2((acc[2] + bias[c][2]) * sc, (acc[3] + bias[c][3]) * sc)};
      if (row < N) {
        if (which == 0) {
          if (!A32_DIAG_FLAG(p.no_q_store)) *reinterpret_cast<u32x2*>(qo + (size_t)row * 32 + half * 16 + 4 * g) = v;
        } else if (which == 1) {                                   // features 16 half + 4g ..: chunk 2 half + (g >> 1), 8 bytes into it
          *reinterpret_cast<u32x2*>(slot + row * 64 + (((2 * half + (g >> 1)) ^ ((row >> 2) & 3)) << 4) + (g & 1) * 8) = v;
        } else {
          *reinterpret_cast<u32x2*>(slot + A32_K_BYTES + row * 64 + (16 * half + 4 * g) * 2) = v;
        }
      }
    }
  }
}

// NW = 4: three workgroups per CU (rounds 4-5).  NW = 8 (round 5 experiment, KVQ_ATTN_WAVES): one workgroup of eight waves per CU — the 13
// q-blocks of a unit spread over twice the waves (half the latency of a unit without re-staging K | V, as a q-split would), 52 KB of LDS per
// CU instead of 156; the register budget stays that of three waves per SIMD so that another kernel's waves fit beside it.
template <typename E, bool FUSED, bool DSPLIT, int NW = 4, bool SHORT = false, bool COOP = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 1) __attribute__((amdgpu_waves_per_eu(3, 3))) void window_attention32_kernel(Attn32Params p) {
  static_assert(!COOP || (SHORT && NW == 4), "the co-operative last q-block is built for SHORT launches of 4 waves");
  constexpr int A32_WAVES = NW;
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* ticket = reinterpret_cast<int*>(smem + A32_OFF_CTR);
  using V8 = typename E::v8;
  // Block order as attn.hip's dense kernel: the workgroups that share one (window type, head) bias run on the SAME XCD back to back
  // (workgroup b -> XCD b % 8); FUSED: an XCD takes whole window types, heads fastest (they read the same norm1 rows).
  const int nclip = p.BW / p.nW, nrep = p.nW / p.n_types, npair = p.n_types * p.nH, per_pair = nclip * nrep * p.qsplit;
  const int xcd = blockIdx.x & 7, slot_i = blockIdx.x >> 3;
  int pair, clip, rep, part;
  if (FUSED) {
    const int per_type = p.nH * nclip * nrep, wt_ = (slot_i / per_type) * 8 + xcd, sub = slot_i % per_type;
    if (wt_ >= p.n_types) return;
    const int h_ = sub % p.nH, rest = sub / p.nH;
    pair = wt_ * p.nH + h_; clip = rest % nclip; rep = rest / nclip; part = 0;
  } else {
    const int sub = slot_i % per_pair;
    pair = (slot_i / per_pair) * 8 + xcd;
    clip = sub % nclip; rep = (sub / nclip) % nrep; part = sub / (nclip * nrep);
  }
  if (pair >= npair) return;
  const int wt = pair / p.nH, h = pair - wt * p.nH, w = rep * p.n_types + wt, bw = clip * p.nW + w;
  const int tid = threadIdx.x, lane = tid & 63, N = p.N, nqb = (N + 31) >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t Mtot = (size_t)p.BW * N;
  const int C = p.nH * 32;
  const uint16_t* Qg = p.qkv + ((size_t)h * Mtot + (size_t)bw * N) * 32;

  if (tid < 64) *reinterpret_cast<u32x4*>(smem + A32_OFF_ZERO + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
  if (FUSED) {
    // rows N..399 of both images: zeros (the projection writes rows < N only)
    for (int i = tid; i < (A32_ROWS - N) * 8; i += A32_WAVES * 64)
      *reinterpret_cast<u32x4*>(smem + (i & 4 ? A32_K_BYTES : 0) + (N + (i >> 3)) * 64 + (i & 3) * 16) = (u32x4){0u, 0u, 0u, 0u};
    a32_fused_qkv<E, 3, NW>(p, smem, bw, h, N);
  } else {
    // K | V by buffer-resource LDS-DMA (rows past N read zeros): K with its 16-B chunks XOR-swizzled by (row >> 2) & 3, V as it lies
    const uint16_t* Kg = p.qkv + ((size_t)(1 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
    const uint16_t* Vg = p.qkv + ((size_t)(2 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Kg), 0, N * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Vg), 0, N * 64, 0x00020000);
    const unsigned vo_k = (unsigned)(lane >> 2) * 64u + (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u, vo_v = (unsigned)lane * 16u;
    for (int it = wave; it < A32_ROWS / 16; it += A32_WAVES) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (a32_lds_t)(smem + it * 1024), 16, vo_k, it * 1024, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (a32_lds_t)(smem + A32_K_BYTES + it * 1024), 16, vo_v, it * 1024, 0, 0);
    }
  }
  const int q_lo = part * nqb / p.qsplit, q_hi = (part + 1) * nqb / p.qsplit;
  if (tid == 0) {
    *ticket = q_lo;
    if (COOP) *reinterpret_cast<int*>(smem + A32_OFF_DONE) = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!FUSED && p.pad_mask) {
    // padded partition: the padding rows of the window are keys with k | v = qkv(0) = the bias (swin_backbone.py:416-449); their rows of
    // the qkv buffer were never written — the images get the 16-bit rounding of the bias here (thread = one 16-B chunk of a row)
    const uint32_t* pm = p.pad_mask + (size_t)w * A32_KB;
    const int c = tid & 3;
    const float* kb = p.b_qkv + C + h * 32 + 8 * c;
    const float* vb = p.b_qkv + 2 * C + h * 32 + 8 * c;
    const u32x4 kc = {E::pack2(kb[0], kb[1]), E::pack2(kb[2], kb[3]), E::pack2(kb[4], kb[5]), E::pack2(kb[6], kb[7])};
    const u32x4 vc = {E::pack2(vb[0], vb[1]), E::pack2(vb[2], vb[3]), E::pack2(vb[4], vb[5]), E::pack2(vb[6], vb[7])};
    for (int r = tid >> 2; r < N; r += A32_WAVES * 16)
      if ((pm[r >> 5] >> (r & 31)) & 1u) {
        *reinterpret_cast<u32x4*>(smem + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = kc;
        *reinterpret_cast<u32x4*>(smem + A32_K_BYTES + r * 64 + c * 16) = vc;
      }
    __syncthreads();
  }

  const int q = lane & 31, hi = lane >> 5;
  const uint32_t skip = p.tile_skip ? p.tile_skip[w] : 0u;
  const bool dsplit = DSPLIT && p.dsplit_from >= 0 && w >= p.dsplit_from;
  // COOP: tickets 0 .. nqb-2 are whole q-blocks, nqb-1 .. nqb+2 the four key ranges of the last one (a depth-split window's last q-block
  // sees 7 key blocks only: it stays whole)
  const bool coop = COOP && !dsplit && p.qsplit == 1;
  const int n_tickets = coop ? nqb - 1 + A32_COOP_PARTS : q_hi;
  auto block_of = [&](int t_) -> int { return coop && t_ >= nqb - 1 ? nqb - 1 : t_; };
  auto take = [&]() -> int {
    int t_, qb_;
    do {                                                        // q-blocks whose two 16-row tiles are padding only are passed over
      t_ = 0;
      if (lane == 0) t_ = atomicAdd(ticket, 1);
      t_ = __builtin_amdgcn_readfirstlane(t_);
      qb_ = block_of(t_);
    } while (t_ < n_tickets && ((skip >> (2 * qb_)) & 1u) && (((skip >> (2 * qb_ + 1)) & 1u) || 32 * qb_ + 16 >= N));
    return t_;
  };
  const u32x4* img = p.image + (size_t)(A32_DIAG_FLAG(p.image_one) ? h : pair) * nqb * (A32_KB * 128) + lane;
  // the ticket, the q fragments and the first bias tile of the NEXT q-block are requested while this one is computed
  struct Req { V8 qf0, qf1; u32x4 pre[2]; };
  auto request = [&](int t_, Req& r) __attribute__((always_inline)) {
    const int qb = block_of(t_), part = t_ - (nqb - 1);
    const int qrow = min(32 * qb + q, N - 1);
    r.qf0 = *reinterpret_cast<const V8*>(Qg + (size_t)qrow * 32 + 8 * hi);
    r.qf1 = *reinterpret_cast<const V8*>(Qg + (size_t)qrow * 32 + 16 + 8 * hi);
    if (!FUSED && p.pad_mask && ((p.pad_mask[(size_t)w * A32_KB + qb] >> q) & 1u)) {      // a padding row's q was never written: take zero
      r.qf0 = __builtin_bit_cast(V8, (u32x4){0u, 0u, 0u, 0u});
      r.qf1 = r.qf0;
    }
    const int t0 = (dsplit && qb > 6) ? 6 : (coop && part > 0) ? 1 + 3 * part : 0;        // first key block of what will run
    const u32x4* bd = img + (size_t)min(qb, nqb - 1) * (A32_KB * 128) + t0 * 128;
    r.pre[0] = bd[0]; r.pre[1] = bd[64];
  };
  auto run = [&](int t_, const Req& r) __attribute__((always_inline)) {
    const int qb = block_of(t_);
    const u32x4* bd = img + (size_t)qb * (A32_KB * 128);
    uint16_t* orow = p.out + ((size_t)bw * N + 32 * qb + q) * C + h * 32;
    const bool store = 32 * qb + q < N;
    const int zero = A32_OFF_ZERO >> 4;
    if (COOP && coop && t_ >= nqb - 1) {
      // one key range of the last q-block, then: park the partial state (the first three to arrive) or merge all four and store
      A32State st;
      const int part = t_ - (nqb - 1);
      if (part == 0) a32_qblock<E, 0, 4, false, true>(smem, zero, bd, r.qf0, r.qf1, r.pre, orow, store, &st);
      else if (part == 1) a32_qblock<E, 4, 7, false, true>(smem, zero, bd, r.qf0, r.qf1, r.pre, orow, store, &st);
      else if (part == 2) a32_qblock<E, 7, 10, false, true>(smem, zero, bd, r.qf0, r.qf1, r.pre, orow, store, &st);
      else a32_qblock<E, 10, A32_KB, SHORT, true>(smem, zero, bd, r.qf0, r.qf1, r.pre, orow, store, &st);
      // ranges 0..2 park their state in the slot of their INDEX and count themselves; range 3 (the last ticket: as a rule the last to finish)
      // waits for the three and merges in index order — the result does not depend on which wave ran which range or on who finished first
      int* done = reinterpret_cast<int*>(smem + A32_OFF_DONE);
      float* ml = reinterpret_cast<float*>(smem + A32_OFF_ML);
      const bool row_ok = q < 8;                                   // rows 384 + q of the window: at most 8 are tokens
      if (part < A32_COOP_PARTS - 1) {
        if (row_ok) {
          float* po = reinterpret_cast<float*>(smem + a32_part_off(part, hi) + q * 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(po + 4 * j) = (f32x4){st.O[4 * j], st.O[4 * j + 1], st.O[4 * j + 2], st.O[4 * j + 3]};
          if (hi == 0) *reinterpret_cast<f32x2*>(ml + (part * 8 + q) * 2) = (f32x2){st.nm, st.ls};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        if (lane == 0) while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < A32_COOP_PARTS - 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float nmin = st.nm;
        float onm[A32_COOP_PARTS - 1], ols[A32_COOP_PARTS - 1];
#pragma unroll
        for (int sl = 0; sl < A32_COOP_PARTS - 1; ++sl) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(ml + (sl * 8 + (q & 7)) * 2);
          onm[sl] = v[0]; ols[sl] = v[1];
          nmin = fminf(nmin, v[0]);
        }
        // O_p is scaled by 2^-m_p = 2^nm_p: everything moves to the largest maximum m = -nmin by 2^(nmin - nm_p) <= 1; ranges in index order
        float ls = 0.f;
        f32x16 O;
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) O[r2] = 0.f;
#pragma unroll
        for (int sl = 0; sl < A32_COOP_PARTS - 1; ++sl) {
          const float f = __builtin_amdgcn_exp2f(nmin - onm[sl]);
          ls = fmaf(ols[sl], f, ls);
          const float* po = reinterpret_cast<const float*>(smem + a32_part_off(sl, hi) + (q & 7) * 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(po + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) O[4 * j + e] = fmaf(v[e], f, O[4 * j + e]);
          }
        }
        {
          const float f = __builtin_amdgcn_exp2f(nmin - st.nm);
          ls = fmaf(st.ls, f, ls);
#pragma unroll
          for (int r2 = 0; r2 < 16; ++r2) O[r2] = fmaf(st.O[r2], f, O[r2]);
        }
        a32_finish<E>(O, ls, orow, store && row_ok);
      }
      return;
    }
    // depth-split window (N = 392, halves of 196 tokens): q-blocks 0..5 see key blocks 0..6, 7..12 see 6..12, q-block 6 all of them
    if (dsplit && qb != 6) {
      if (qb < 6) a32_qblock<E, 0, 7>(smem, zero, bd, r.qf0, r.qf1, r.pre, orow, store);
      else a32_qblock<E, 6, A32_KB, SHORT>(smem, zero, bd, r.qf0, r.qf1, r.pre, orow, store);
    } else {
      a32_qblock<E, 0, A32_KB, SHORT>(smem, zero, bd, r.qf0, r.qf1, r.pre, orow, store);
    }
  };
  // ONE inlined body per kind of ticket (rounds 4-5 unrolled the loop twice to rotate two request sets; the copy below is 16 moves per q-block)
  Req cur, nxt;
  int ta = take();
  if (ta < n_tickets) request(ta, cur);
  while (ta < n_tickets) {
    const int tn = take();
    if (tn < n_tickets) request(tn, nxt);
    run(ta, cur);
    ta = tn;
    cur = nxt;
  }
}

// The image builder (one workgroup per (window type, head)): attn.hip's bias arithmetic — idx = code_q - code_k + center,
// b = fma(gate, r - f, f), the shift mask REPLACES it by -100 — minus the row maximum over the un-masked keys, fp16, in the
// kernel's accumulator layout: [qb][kb][half][lane (q = lane & 31, hi = lane >> 5)][8]: register r = 8 half + e of the
// 32 x 32 score block = key 32 kb + (r & 3) + 8 (r >> 2) + 4 hi.
struct Bias32BuildParams {
  const int32_t* tok;
  const float* rpb;
  const float* fpb;
  int table_len, center, nW, N, nH, use_mask;
  uint16_t* out;
  unsigned* max_abs;
};

__global__ __launch_bounds__(256) void bias32_build_kernel(Bias32BuildParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  f32x2* tab = reinterpret_cast<f32x2*>(bsm);
  int2* tokL = reinterpret_cast<int2*>(bsm + (size_t)p.table_len * 8);
  float* rmax = reinterpret_cast<float*>(bsm + (size_t)p.table_len * 8 + (size_t)p.N * 8);
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
  for (int qq = tid; qq < N; qq += 256) {
    const int2 tq = tokL[qq];
    float mx = -INFINITY;
    for (int key = 0; key < N; ++key) {
      bool masked;
      const float b = value(tq, key, &masked);
      if (!masked) { mx = fmaxf(mx, b); big = fmaxf(big, fabsf(b)); }
    }
    rmax[qq] = mx;
  }
  if (p.max_abs) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = fmaxf(big, __shfl_xor(big, o));
    if ((tid & 63) == 0) atomicMax(p.max_abs, __float_as_uint(big));
  }
  __syncthreads();
  const int nqb = (N + 31) >> 5, lane = tid & 63, wave = tid >> 6, ql = lane & 31, hi = lane >> 5;
  uint16_t* img = p.out + ((size_t)w * p.nH + h) * nqb * A32_KB * 1024;
  for (int tile = wave; tile < nqb * A32_KB * 2; tile += 4) {
    const int half = tile & 1, blk = tile >> 1, qb = blk / A32_KB, kb = blk - qb * A32_KB, qq = 32 * qb + ql;
    const int2 tq = tokL[qq < N ? qq : N - 1];
    const float shift = rmax[qq < N ? qq : N - 1];
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * half + e, key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float b = 0.f;
      if (key >= N) {
        b = A32_OFF;
      } else if (qq < N) {
        bool masked;
        b = value(tq, key, &masked);
        if (masked) b = -100.0f;
        b -= shift;
      }
      v[e] = b;
    }
    *reinterpret_cast<u32x4*>(img + (size_t)tile * 512 + lane * 8) =
        (u32x4){Fp16::pack2_raw(v[0], v[1]), Fp16::pack2_raw(v[2], v[3]), Fp16::pack2_raw(v[4], v[5]), Fp16::pack2_raw(v[6], v[7])};
  }
}

template <typename E, bool FUSED, bool DSPLIT, int NW, bool SHORT, bool COOP = false>
static int launch_attn32_nw(const Attn32Params& p, hipStream_t st) {
  auto kern = window_attention32_kernel<E, FUSED, DSPLIT, NW, SHORT, COOP>;
  // KVQ_ATTN_LDS_PAD (A/B knob): extra dynamic LDS bytes per workgroup — 2048 makes it two workgroups per CU instead of three
  static const int lds_pad = getenv("KVQ_ATTN_LDS_PAD") ? atoi(getenv("KVQ_ATTN_LDS_PAD")) : 0;
  const int lds_bytes = (COOP ? A32_LDS_COOP : A32_LDS) + lds_pad;
  LdsOptIn opt;
  if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), lds_bytes)) return rc;
  const int nclip = p.BW / p.nW, nrep = p.nW / p.n_types, npair = p.n_types * p.nH;
  unsigned grid = (unsigned)(8 * ceil_div(npair, 8) * nclip * nrep * p.qsplit);
  if (FUSED) grid = (unsigned)(8 * ceil_div(p.n_types, 8) * p.nH * nclip * nrep);      // XCDs take whole window types
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, st, p);
  KVQ_CHECK_LAUNCH("window_attention32_kernel");
  return KVQ_OK;
}

template <typename E, bool FUSED, bool DSPLIT>
static int launch_attn32(const Attn32Params& p, hipStream_t st) {
  // KVQ_ATTN_WAVES: 4 (default) | 8 | 0 = eight waves where the launch holds fewer than 768 units (stages 2-3 at 4 clips)
  static const int nw_env = getenv("KVQ_ATTN_WAVES") ? atoi(getenv("KVQ_ATTN_WAVES")) : 4;
  const bool eight = nw_env == 8 || (nw_env == 0 && (long)p.BW * p.nH < 768);
  // KVQ_ATTN_SHORT=0: the generic 13th key block for every N (A/B runs; results agree to the last bit: the skipped registers are padding keys)
  static const bool short_ok = getenv("KVQ_ATTN_SHORT") ? atoi(getenv("KVQ_ATTN_SHORT")) != 0 : true;
  // KVQ_ATTN_COOP=0: the last q-block as one ticket (rounds 4-5); results differ by the fp32 rounding of the four-way merge on its <= 8 rows
  // The co-operative last q-block is OFF by default: built, pinned (tests/test_gpu_kernels.py::test_window_attention32_cooperative_last_qblock)
  // and measured in round 6 (profiles/r06_attn_coop_ab.txt).  Alone on the chip it shortens the launches that are ONE round of workgroups
  // (stages 2-3, their duration is a workgroup's lifetime: 33.7 -> 31.7, 26.3 -> 24.1 us) and lengthens those of several rounds, whose
  // workgroups run out of step anyway (stage 0: 95 -> 104 us: four range prologues + the merge instead of one q-block); on the 4-lane line
  // both choices are level with the plain form (400.4 / 400.1 / 400.2 videos/s, same box, alternating).  KVQ_ATTN_COOP=1: where one clip
  // holds <= 96 (window, head) units; 2: every SHORT launch.  A window's rows differ between the forms by the fp32 rounding of the merge,
  // so the choice follows nW x nH (the units of ONE clip), never the batch.
  static const int coop_env = getenv("KVQ_ATTN_COOP") ? atoi(getenv("KVQ_ATTN_COOP")) : 0;
  if (short_ok && !eight && p.N > 384 && p.N <= 392) {
    const bool coop = coop_env >= 2 || (coop_env == 1 && p.nW * p.nH <= 96);
    if (coop && p.coop_ok && p.qsplit == 1) return launch_attn32_nw<E, FUSED, DSPLIT, 4, true, true>(p, st);
    return launch_attn32_nw<E, FUSED, DSPLIT, 4, true>(p, st);
  }
  return eight ? launch_attn32_nw<E, FUSED, DSPLIT, 8, false>(p, st) : launch_attn32_nw<E, FUSED, DSPLIT, 4, false>(p, st);
}

template <typename E>
static int launch_attn32_e(const Attn32Params& p, hipStream_t st) {
  if (p.x_ln) return p.dsplit_from >= 0 ? launch_attn32<E, true, true>(p, st) : launch_attn32<E, true, false>(p, st);
  return p.dsplit_from >= 0 ? launch_attn32<E, false, true>(p, st) : launch_attn32<E, false, false>(p, st);
}

}  // namespace kvq

extern "C" size_t kvq_attn_bias32_bytes(int n_types, int N, int num_heads) {
  if (n_types <= 0 || N < 1 || N > 400 || num_heads <= 0) return 0;
  return (size_t)n_types * num_heads * ((N + 31) / 32) * kvq::A32_KB * 2048 + 2048;     // + one block: the kernel requests block t0 + 1 unconditionally
}

extern "C" int kvq_attn_bias32_build(const int32_t* tok, const float* rpb, const float* fpb, int table_len, int center,
                                          int nW, int N, int num_heads, int use_mask, void* out, float* max_abs, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(tok && rpb && out, KVQ_ERR_NULL, "kvq_attn_bias32_build: NULL pointer");
  KVQ_REQUIRE(kvq_attn_bias32_bytes(nW, N, num_heads) > 0 && table_len > 0, KVQ_ERR_SHAPE,
              "kvq_attn_bias32_build: bad shape nW=%d N=%d nH=%d", nW, N, num_heads);
  KVQ_REQUIRE(((size_t)out & 15) == 0, KVQ_ERR_SHAPE, "kvq_attn_bias32_build: out must be 16-byte aligned");
  Bias32BuildParams p{tok, rpb, fpb, table_len, center, nW, N, num_heads, use_mask, (uint16_t*)out, (unsigned*)max_abs};
  const size_t lds = (size_t)table_len * 8 + (size_t)N * 12;
  KVQ_REQUIRE(lds <= 64 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_attn_bias32_build: table of %d entries does not fit the builder's LDS", table_len);
  hipLaunchKernelGGL(bias32_build_kernel, dim3((unsigned)num_heads, (unsigned)nW), dim3(256), lds, (hipStream_t)stream, p);
  KVQ_CHECK_LAUNCH("bias32_build_kernel");
  return KVQ_OK;
}

extern "C" int kvq_window_attention32(const KvqAttnDenseArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->qkv && a->bias_dense && a->out, KVQ_ERR_NULL, "kvq_window_attention32: NULL pointer");
  const int BW = a->BW, nW = a->nW, N = a->N, num_heads = a->num_heads, n_types = a->n_types;
  KVQ_REQUIRE(BW > 0 && nW > 0 && BW % nW == 0 && num_heads > 0 && n_types > 0 && nW % n_types == 0, KVQ_ERR_SHAPE,
              "kvq_window_attention32: bad shape BW=%d nW=%d n_types=%d nH=%d", BW, nW, n_types, num_heads);
  KVQ_REQUIRE(N >= 1 && N <= 400, KVQ_ERR_UNSUPPORTED, "kvq_window_attention32: window of %d tokens unsupported (1..400)", N);
  KVQ_REQUIRE(((size_t)a->bias_dense & 15) == 0, KVQ_ERR_SHAPE, "kvq_window_attention32: the bias image must be 16-byte aligned");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_window_attention32: dtype %d", a->dtype);
  KVQ_REQUIRE(a->dsplit_from < 0 || (N == 392 && a->dsplit_from < nW), KVQ_ERR_UNSUPPORTED,
              "kvq_window_attention32: depth-split windows need the (8,7,7) window (N = 392); got N=%d from=%d", N, a->dsplit_from);
  const int units = BW * num_heads, nqb = (N + 31) / 32;
  int qsplit = units >= 768 ? 1 : 768 / units;        // 768 = 256 CUs x 3 resident workgroups
  // Round 5: no q-split by default.  Splitting a (window, head) unit over 2-4 workgroups fills the chip when the launch is alone on it
  // (stages 2-3: 384 / 192 units) and re-stages K | V in every part; in the 4-lane bench line the un-split form is +1.2 % (397.6 -> 402.2
  // videos/s, three alternating pairs, profiles/r05_qsplit_streams_ab.txt): what counts there is CU x time.  KVQ_ATTN_QSPLIT_MAX=4: rounds 3-4.
  static const int qsplit_max = getenv("KVQ_ATTN_QSPLIT_MAX") ? atoi(getenv("KVQ_ATTN_QSPLIT_MAX")) : (latency_mode() ? 4 : 1);      // (results do not depend on it)
  qsplit = qsplit > qsplit_max ? qsplit_max : qsplit;
  qsplit = qsplit < 1 ? 1 : qsplit;                   // (KVQ_ATTN_QSPLIT_MAX=0 must not give an empty grid)
  qsplit = qsplit > nqb ? nqb : qsplit;
  Attn32Params p{a->qkv, (const u32x4*)a->bias_dense, BW, nW, N, num_heads, n_types, qsplit, a->out, a->tile_skip,
                     a->dsplit_from < 0 ? -1 : a->dsplit_from};
#ifdef KVQ_DIAG
  { static const bool one = getenv("KVQ_IMAGE_ONE_TYPE") && atoi(getenv("KVQ_IMAGE_ONE_TYPE")) == 1; p.image_one = one; }
  { static const bool nq = getenv("KVQ_NO_Q_STORE") && atoi(getenv("KVQ_NO_Q_STORE")) == 1; p.no_q_store = nq; }
#endif
  p.coop_ok = qsplit_max == 1;       // (a q-split that follows the batch would make the form, hence a window's rounding, follow the batch)
  if (a->pad_mask) {
    KVQ_REQUIRE(a->b_qkv && !a->x_ln, KVQ_ERR_NULL, "kvq_window_attention32: pad_mask needs b_qkv (and excludes the fused projection)");
    p.pad_mask = a->pad_mask; p.b_qkv = a->b_qkv;
  }
  if (a->x_ln) {
    const int C = 32 * num_heads;
    KVQ_REQUIRE(a->w_qkv && a->b_qkv, KVQ_ERR_NULL, "kvq_window_attention32: x_ln without w_qkv / b_qkv");
    KVQ_REQUIRE(C == 96, KVQ_ERR_UNSUPPORTED, "kvq_window_attention32: the fused qkv projection is built for C = 96 (got %d)", C);
    KVQ_REQUIRE((((size_t)a->x_ln | (size_t)a->w_qkv | (size_t)a->b_qkv) & 15) == 0, KVQ_ERR_SHAPE,
                "kvq_window_attention32: x_ln / w_qkv / b_qkv must be 16-byte aligned");
    p.qsplit = 1;       // one workgroup per (window, head) whatever the batch: a block's path must not depend on what shares the launch
    p.x_ln = a->x_ln; p.w_qkv = a->w_qkv; p.b_qkv = a->b_qkv; p.q_scale = a->q_scale; p.q_out = const_cast<uint16_t*>(a->qkv);
  }
  hipStream_t st = (hipStream_t)stream;
  return a->dtype == KVQ_DT_FP16 ? launch_attn32_e<Fp16>(p, st) : launch_attn32_e<Bf16>(p, st);
}
