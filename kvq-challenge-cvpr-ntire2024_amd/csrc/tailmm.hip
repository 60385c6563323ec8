// The fused post-attention launch for WIDE rows (C = 384: stage 2 of Swin-T/S; C = 512: stage 2 of Swin-B, template CF = 4, numbers
// below are for C = 384) as a register-blocked GEMM chain on
// v_mfma_f32_32x32x16 (round 1's token-per-lane 16x16x32 launch for C = 384, tail16.hip, was retired in round 3).
//
// Same computation (swin_backbone.py:479-516): x += proj(attn) (window-reverse / roll-back / crop through the row map);
// x += fc2(GELU(fc1(norm2(x)))) [+ the next block's norm1 in ITS window order].
//
// Why not token-per-lane as for C <= 192: at C = 384 that design feeds every 16-cycle MFMA with its own 1 KB weight fragment from LDS
// and has ONE wave per SIMD, so its floor is the wave's instruction stream: 2592 MFMAs x 17 ticks + 2592 fragment reads + GELU, additive on
// gfx950 (tools/ubench/pipe_share.hip: VALU and LDS issue of a wave do not hide under that wave's own MFMAs beyond ~4 cycles
// per 16x16x32 / ~11 per 32x32x16).  Here a workgroup is still 64 tokens x 4 waves, but a wave owns a FEATURE slice for all
// 64 tokens instead of a token slice for all features:
//   * D[feature][token] = W[feature][k] . X[token][k]: weights are the A operand, activations the B operand, 32x32x16 tiles;
//     a wave's k-step is 3 weight fragments + 2 activation fragments for 6 MFMAs (fc1: 2 + 2 for 4) — 1296 MFMAs of 32
//     cycles per wave instead of 2592 of 16, and 5 LDS reads per 192 MFMA cycles instead of 12;
//   * a wave reads only ITS feature rows' weights, so each wave streams its own fragment list (648 x 1 KB, consumption
//     order, packed by tailmm_pack_kernel) global -> VGPR through a private REGISTER ring with the compiler's counted vmcnt —
//     no LDS round trip, no workgroup barrier per fragment; the stream runs 20 fragments ahead across all phases;
//   * activations go through LDS as ready-made B fragments: the attention rows by LDS-DMA (row gather), norm2's output and
//     the GELU output written by their producers straight from the accumulator layout (a lane holds one token and, per tile
//     and register quad pair, 8 k values = one 16-byte fragment slot; the k order inside a 16-step is permuted accordingly in
//     the packed weights): 3 + 2 x 6 workgroup barriers per launch;
//   * D keeps a token per lane (column) — residual, LayerNorm statistics (in-lane + lane^32 + a 4-wave exchange through
//     LDS), bias and GELU are register arithmetic as before.
// LDS: 48 KB activation tile (attention rows, then norm2 rows) + 32 KB GELU chunk + 12 KB parameters.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "tail.hpp"

// The 42 instantiations of the kernel take ~5 minutes in one translation unit: the build (_build.py) compiles this file FOUR times with
// -DKVQ_TAILMM_PART=0..3 — part 0 holds the host side (packing, dispatch) and the C = 384 product form, parts 1-3 the other forms —
// and links the four objects; without the macro (study builds, a plain `hipcc -c`) everything is one unit.
#ifdef KVQ_TAILMM_PART
#define MM_PART_HERE(k) (KVQ_TAILMM_PART == (k))
#else
#define MM_PART_HERE(k) 1
#endif

#ifndef MM_ABL
#define MM_ABL 0            // study builds (tools/tailmm_variant.sh), bit mask: 1 no activation-fragment LDS reads past a phase's first body,
#endif                      // 2 no weight stream past the first ring fill, 8 GELU replaced by a pack, 16 fc1's MFMAs skipped, 32 fc2's MFMAs skipped
                            // (profiles/r06_tailmm_study.txt: where a C = 384 workgroup's cycles go)

namespace kvq {

typedef __attribute__((address_space(3))) void* mm_lds_t;
typedef __attribute__((address_space(1))) const void* mm_gbl_t;

// Geometry by CF = C / 128 = 32-feature tiles per wave: 3 (C = 384: stage 2 of Swin-T / -S) or 4 (C = 512: stage 2 of Swin-B,
// register ring only — its 64 KB activation tile leaves no room for an LDS weight ring), and by HC = hidden units per MLP chunk:
//   HC = 256: a wave's fc1 slice is 64 units (two tiles): 512 VGPRs, 94 KB of LDS, ONE workgroup per CU (rounds 2-4);
//   HC = 128: 32 units (one tile) and a 12-fragment ring: <= 256 VGPRs, 77 KB of LDS at C = 384 — TWO workgroups per CU, i.e. two
//             independent instruction streams per SIMD: one workgroup's serial phases (row loads, LayerNorm exchanges, GELU, barriers,
//             the store tail) run under the other's MFMAs (round 5; the per-CU weight stream per token is unchanged).
//   TT = 4 (round 6): 128 tokens per workgroup — every weight fragment of the register ring feeds FOUR MFMAs (token tiles) instead of two, which
//             halves the L1 -> VGPR weight stream per MFMA (1 KB per two 32x32x16 is 64 B / clk per CU, the path's limit, at full matrix rate).
//             192 + 64 accumulator registers: ONE wave per SIMD, 142 KB of LDS at C = 384 (HC = 128).
template <int CF, int HC_ = 256, int TT_ = 2>
struct MMc {
  static constexpr int TT = TT_, TOK = 32 * TT_;                        // token tiles of 32, tokens per workgroup
  static constexpr int C = 128 * CF, H = 4 * C, W = 32 * CF;            // channels, hidden units, features per wave
  static constexpr int HC = HC_, KS_H = HC / 16, HT = HC / 128;         // hidden chunk, its k-steps, fc1 tiles per wave
  static constexpr int NCH = H / HC, KS_C = C / 16;                     // hidden chunks, k-steps over C
  static constexpr int NF_PROJ = KS_C * CF, NF_FC1 = KS_C * HT, NF_FC2 = KS_H * CF;
  static constexpr int NF = NF_PROJ + NCH * (NF_FC1 + NF_FC2);          // fragments per wave: 648 / 1152
  static constexpr int OFF_X = 0;                                       // [KS_C k-steps][2 token tiles][64 lanes][16 B] = 48 / 64 KB
  static constexpr int OFF_G = OFF_X + KS_C * TT * 1024;                 // [KS_H][2][64][16 B] = 32 / 16 KB
  static constexpr int OFF_PRM = OFF_G + KS_H * TT * 1024;               // b1[H] g2[C] b2n[C] proj_b[C] b2[C] fp32
  static constexpr int PRM_FLOATS = H + 4 * C;
  static constexpr int OFF_RED = OFF_PRM + PRM_FLOATS * 4;              // [4 waves][64 tokens] fp32
  static constexpr int LDS = OFF_RED + 4 * TOK * 4;
  static constexpr int WG_PER_CU = HC == 128 && 2 * LDS <= 163840 ? 2 : 1;
  // register budget in waves per SIMD: the HC = 128 forms up to C = 512 keep to 256 VGPRs so that another workgroup (C = 384: of this launch;
  // C = 512, 97 KB of LDS: of another lane's launch) fits beside them; C = 768 needs 192 accumulator registers and takes the whole file
  static constexpr int REG_WAVES = HC == 128 && CF <= 4 && TT == 2 ? 2 : 1;
#ifndef KVQ_TAILMM_KU1
#define KVQ_TAILMM_KU1 1
#endif
  static constexpr int KU1 = HC == 128 ? KVQ_TAILMM_KU1 : 1;            // k-steps per loop body where a wave owns ONE weight tile (fc1 at HC = 128)
  static constexpr bool PIPE = HC != 128;          // the MLP chunks software-pipelined in the wave (fc1 of chunk c + 1 ahead of fc2 of chunk c): the HC = 256 forms
  // packed image: 4 waves x NF KB of fragments, then fp32 parameters b1 | g2 | b2n | proj_b | b2
  static constexpr size_t PACK_FRAG_BYTES = (size_t)4 * NF * 1024;
  // register ring: every phase consumes a multiple of VR_R fragments (72 | 48 | 48 of 24; 128 | 64 | 64 of 16 — 32 slots at C = 512
  // spill: 128 + 64 accumulator registers are there already; HC = 128: 72 | 24 | 24 of 12)
  static constexpr int VR_R = HC == 128 ? (CF == 3 ? 12 : CF >= 4 ? 16 : 8) : (CF == 3 ? 24 : 16), VR_PF = VR_R - 4;      // (C = 256: 32 | 32 | 32 fragments per phase)
  static_assert(NF_PROJ % VR_R == 0 && NF_FC1 % VR_R == 0 && NF_FC2 % VR_R == 0, "every phase starts at register slot 0");
  static_assert(LDS <= 163840, "LDS");
};

#if MM_PART_HERE(0)
// KVQ_TAILMM_HC=256 takes rounds 2-4's one-workgroup-per-CU form at C = 384 (A/B runs); read once — the packed image and the launch
// must agree.  C = 512 (Swin-B stage 2, the C5 line) takes HC = 128 too: 2 x 97 KB of LDS do not fit, so it is still one workgroup of
// this launch per CU, but at 256 registers and no spills (HC = 256 at C = 512: 512 registers, 8 / 17 / 115 spilled by MODE) a workgroup
// of another lane's launch fits beside it — C5 22.98 -> 23.46 videos/s, same box, alternating (profiles/r05_hc512_ab.txt);
// KVQ_TAILMM_HC512=256 is the old form.  C = 256 keeps HC = 256 (not on any benchmarked path).
static int tailmm_hc(int C) {
  static const int env = getenv("KVQ_TAILMM_HC") ? atoi(getenv("KVQ_TAILMM_HC")) : (latency_mode() ? 256 : 128);
  static const int env512 = getenv("KVQ_TAILMM_HC512") ? atoi(getenv("KVQ_TAILMM_HC512")) : (latency_mode() ? 256 : 128);
  if (C == 512) return env512 == 128 ? 128 : 256;
  return C == 384 && env == 128 ? 128 : 256;
}

// KVQ_TAILMM_TOK=128 (round 6): the 128-token workgroup at C = 384, HC = 128 (same packed image: the fragment lists do not depend on it)
static int tailmm_tok(int C) {
  static const int env = getenv("KVQ_TAILMM_TOK") ? atoi(getenv("KVQ_TAILMM_TOK")) : 64;
  return C == 384 && env == 128 ? 128 : 64;
}

// C = 768 (round 5: stage 3 of Swin-T / -S; CF = 6, HC = 128, one workgroup per CU: 137 KB of LDS, 192 accumulator registers per wave).  At
// 4 clips it is a launch of 49 workgroups that takes 193-254 us where the GEMM / LayerNorm launches it replaces take 115 us ALONE on the
// chip — and the 4-lane bench line gains 4.5 % (370.9 / 373.3 / 368.3 -> 389.5 / 386.4 / 386.1 videos/s, same box, alternating;
// profiles/r05_tail768_ab.txt): in the mix what a launch costs is CU x time, not its latency, and 49 busy CUs for 250 us are a third of
// five thin launches spread over the chip.  KVQ_TAILMM_768=0: the GEMM chain.
static bool tailmm_768() {
  static const bool on = getenv("KVQ_TAILMM_768") ? atoi(getenv("KVQ_TAILMM_768")) != 0 : !latency_mode();
  return on;
}
bool tailmm_supported(int C, int hidden) { return (C == 256 || C == 384 || C == 512 || (C == 768 && tailmm_768())) && hidden == 4 * C; }
size_t tailmm_pack_bytes(int C, int hidden) {
  if (!tailmm_supported(C, hidden)) return 0;
  const size_t frag = C == 384 ? (tailmm_hc(C) == 128 ? MMc<3, 128>::PACK_FRAG_BYTES : MMc<3>::PACK_FRAG_BYTES)
                      : C == 512 ? (tailmm_hc(C) == 128 ? MMc<4, 128>::PACK_FRAG_BYTES : MMc<4>::PACK_FRAG_BYTES) : C == 768 ? MMc<6, 128>::PACK_FRAG_BYTES : MMc<2>::PACK_FRAG_BYTES;
  return frag + (((size_t)(hidden + 4 * C) * 4 + 255) & ~(size_t)255);
}

// k offset inside a 16-step of element e of fragment slot group g when the B operand is written from accumulators: a lane
// (token, half) holds, per 32-row tile, rows 8q + 4 half + i; quads (q even, q odd) of one 16-row half are one slot
__host__ __device__ inline int mm_kperm(int g, int e) { return e < 4 ? 4 * g + e : 8 + 4 * g + (e - 4); }

template <int CF, int HC>
__global__ void tailmm_pack_kernel(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b, const float* n2w,
                                   const float* n2b, const float* b1, const float* b2, unsigned char* out) {
  using K = MMc<CF, HC>;
  constexpr int MM_C = K::C, MM_H = K::H, MM_NCH = K::NCH, MM_NF = K::NF, MM_NF_PROJ = K::NF_PROJ, MM_NF_FC1 = K::NF_FC1, MM_NF_FC2 = K::NF_FC2;
  constexpr int MM_HC = K::HC, HT = K::HT;
  constexpr int MM_PACK_PRM_FLOATS = K::PRM_FLOATS;
  constexpr size_t MM_PACK_FRAG_BYTES = K::PACK_FRAG_BYTES;
  const long gi = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_slots = (long)4 * MM_NF * 64;
  if (gi < n_slots) {
    const int lane = (int)(gi & 63), i = lane & 31, g = lane >> 5;
    const int f = (int)((gi >> 6) % MM_NF), w = (int)((gi >> 6) / MM_NF);
    uint16_t* o = reinterpret_cast<uint16_t*>(out + gi * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint16_t v;
      if (f < MM_NF_PROJ) {                                  // proj: natural k (B fragments come straight from the attention rows)
        const int ks = f / CF, ft = f % CF;
        v = wp[(size_t)(K::W * w + 32 * ft + i) * MM_C + 16 * ks + 8 * g + e];
      } else {
        // consumption order behind proj: fc1(0); then fc1(c+1), fc2(c) for c = 0..4; then fc2(5) — the wave's software pipeline
        const int r = f - MM_NF_PROJ;
        int c, q;
        if (!K::PIPE) {                                       // not software-pipelined: fc1(c), fc2(c), fc1(c + 1), ...
          c = r / (MM_NF_FC1 + MM_NF_FC2); q = r % (MM_NF_FC1 + MM_NF_FC2);
        } else if (r < MM_NF_FC1) { c = 0; q = r; }
        else {
          const int r2 = r - MM_NF_FC1, blk = r2 / (MM_NF_FC1 + MM_NF_FC2), o2 = r2 % (MM_NF_FC1 + MM_NF_FC2);
          if (blk >= MM_NCH - 1) { c = MM_NCH - 1; q = MM_NF_FC1 + (r2 - (MM_NCH - 1) * (MM_NF_FC1 + MM_NF_FC2)); }
          else if (o2 < MM_NF_FC1) { c = blk + 1; q = o2; }
          else { c = blk; q = o2; }
        }
        if (q < MM_NF_FC1) {                                 // fc1 rows of chunk c, k = channel in accumulator order
          const int ks = q / HT, ft = q % HT;
          v = w1[(size_t)(MM_HC * c + 32 * HT * w + 32 * ft + i) * MM_C + 16 * ks + mm_kperm(g, e)];
        } else {                                             // fc2: all C outputs, k = hidden unit of chunk c in accumulator order
          const int q2 = q - MM_NF_FC1, ks = q2 / CF, ft = q2 % CF;
          v = w2[(size_t)(K::W * w + 32 * ft + i) * MM_H + MM_HC * c + 16 * ks + mm_kperm(g, e)];
        }
      }
      o[e] = v;
    }
  } else if (gi < n_slots + MM_PACK_PRM_FLOATS) {
    const int q = (int)(gi - n_slots);
    float v;
    if (q < MM_H) v = b1[q];
    else if (q < MM_H + MM_C) v = n2w[q - MM_H];
    else if (q < MM_H + 2 * MM_C) v = n2b[q - MM_H - MM_C];
    else if (q < MM_H + 3 * MM_C) v = proj_b[q - MM_H - 2 * MM_C];
    else v = b2[q - MM_H - 3 * MM_C];
    reinterpret_cast<float*>(out + MM_PACK_FRAG_BYTES)[q] = v;
  }
}

template <int CF, int HC>
static void launch_pack(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b, const float* n2w, const float* n2b,
                        const float* b1, const float* b2, unsigned char* out, hipStream_t st) {
  const long total = (long)4 * MMc<CF, HC>::NF * 64 + MMc<CF, HC>::PRM_FLOATS;
  hipLaunchKernelGGL((tailmm_pack_kernel<CF, HC>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wp, w1, w2, proj_b, n2w, n2b, b1, b2, out);
}

int tailmm_pack(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b, const float* n2w, const float* n2b,
                const float* b1, const float* b2, int C, int hidden, unsigned char* out, hipStream_t st) {
  if (C == 512 && tailmm_hc(C) == 128) launch_pack<4, 128>(wp, w1, w2, proj_b, n2w, n2b, b1, b2, out, st);
  else if (C == 512) launch_pack<4, 256>(wp, w1, w2, proj_b, n2w, n2b, b1, b2, out, st);
  else if (C == 768) launch_pack<6, 128>(wp, w1, w2, proj_b, n2w, n2b, b1, b2, out, st);
  else if (C == 256) launch_pack<2, 256>(wp, w1, w2, proj_b, n2w, n2b, b1, b2, out, st);
  else if (tailmm_hc(C) == 128) launch_pack<3, 128>(wp, w1, w2, proj_b, n2w, n2b, b1, b2, out, st);
  else launch_pack<3, 256>(wp, w1, w2, proj_b, n2w, n2b, b1, b2, out, st);
  KVQ_CHECK_LAUNCH("tailmm_pack_kernel");
  return KVQ_OK;
}

// q | k | v of the NEXT block from this launch (round 5): its qkv weight as a fourth fragment list per wave — [which = q, k, v][k-step]
// [feature tile], the wave's feature slice of each third, k in accumulator order (the norm1 rows reach LDS from the accumulators, like
// norm2's) — consumed behind the MLP by the same register ring.
template <int CF>
__global__ void tailmm_qkv_pack_kernel(const uint16_t* wq, unsigned char* out) {
  using K = MMc<CF>;
  const long gi = (long)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int NFQ = 3 * K::NF_PROJ;
  if (gi >= (long)4 * NFQ * 64) return;
  const int lane = (int)(gi & 63), i = lane & 31, g = lane >> 5;
  const int f = (int)((gi >> 6) % NFQ), w = (int)((gi >> 6) / NFQ);
  const int which = f / K::NF_PROJ, r = f % K::NF_PROJ, ks = r / CF, ft = r % CF;
  uint16_t* o = reinterpret_cast<uint16_t*>(out + gi * 16);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = wq[(size_t)(which * K::C + K::W * w + 32 * ft + i) * K::C + 16 * ks + mm_kperm(g, e)];
}

size_t tailmm_qkv_pack_bytes(int C, int hidden) {
  if (!tailmm_supported(C, hidden)) return 0;
  const int CF = C / 128;
  return (size_t)4 * 3 * (C / 16) * CF * 1024;
}

int tailmm_qkv_pack(const uint16_t* qkv_w, int C, int hidden, unsigned char* out, hipStream_t st) {
  KVQ_REQUIRE(tailmm_supported(C, hidden), KVQ_ERR_UNSUPPORTED, "kvq_block_tail_qkv_pack: C=%d hidden=%d", C, hidden);
  const long total = (long)tailmm_qkv_pack_bytes(C, hidden) / 16;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (C == 512) hipLaunchKernelGGL(tailmm_qkv_pack_kernel<4>, grid, block, 0, st, qkv_w, out);
  else if (C == 768) hipLaunchKernelGGL(tailmm_qkv_pack_kernel<6>, grid, block, 0, st, qkv_w, out);
  else if (C == 256) hipLaunchKernelGGL(tailmm_qkv_pack_kernel<2>, grid, block, 0, st, qkv_w, out);
  else hipLaunchKernelGGL(tailmm_qkv_pack_kernel<3>, grid, block, 0, st, qkv_w, out);
  KVQ_CHECK_LAUNCH("tailmm_qkv_pack_kernel");
  return KVQ_OK;
}

#endif  // MM_PART_HERE(0)

// The weight stream: a wave's weight fragments are PRIVATE to it, and plain VGPR loads stream as fast as LDS-DMA
// (tools/ubench/l2_stream.hip), so the fragments go global -> VGPR and are the MFMA A operand as they arrive: a REGISTER ring of
// VR_R fragments per wave, VR_PF in flight (80 KB per CU at C = 384), no weight reads from LDS.  Slots are compile-time: every phase
// consumes a multiple of VR_R fragments, so phase-local position q lives in slot q % VR_R.  Plain loads, not inline asm: the
// compiler's own vmcnt bookkeeping is exact in the unrolled phases (s_waitcnt vmcnt(19 / 18) in the steady state), and an inline-asm
// load is unsafe under this register pressure (the allocator splits the live range of a value it believes ready; the late data lands
// in a register handed on).  (Round 2 also carried an LDS-ring form of the stream and ablation builds of it — no weight stream 65 us,
// no MFMAs 57, neither 43 of 78 — removed in round 3: 79 -> 73 us with the next norm1, 73 -> 64 without, bit-identical.)
template <int N> struct TokVec { float v[N]; __device__ __forceinline__ float operator[](int i) const { return v[i]; } };

template <typename E, int MODE, int CF = 3, int HC = 256, int TT = 2>      // MODE 0: x only; 1: + the next block's norm1 rows; 2: + the next block's q | k | v
__global__ __launch_bounds__(256, (MMc<CF, HC, TT>::REG_WAVES)) void block_tailmm_kernel(TailParams p) {
  constexpr bool EMIT = MODE == 1, QKV = MODE == 2;
  using K = MMc<CF, HC, TT>;
  constexpr int MM_TOK = K::TOK;
  using TV = TokVec<TT>;                             // one value per token tile of the lane
  constexpr int MM_C = K::C, MM_H = K::H, MM_NCH = K::NCH, MM_KS_C = K::KS_C, MM_NF = K::NF, VR_R = K::VR_R, VR_PF = K::VR_PF, FW = K::W;
  constexpr int MM_HC = K::HC, MM_KS_H = K::KS_H, HT = K::HT;
  constexpr int MM_OFF_X = K::OFF_X, MM_OFF_G = K::OFF_G, MM_OFF_PRM = K::OFF_PRM, MM_OFF_RED = K::OFF_RED;
  constexpr int MM_PRM_FLOATS = K::PRM_FLOATS;
  constexpr size_t MM_PACK_FRAG_BYTES = K::PACK_FRAG_BYTES;
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E::v8;
  constexpr int C = MM_C;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* wsrc = p.pack + (size_t)wave * MM_NF * 1024 + lane * 16;    // this wave's fragment list
  float* prm = reinterpret_cast<float*>(lds + MM_OFF_PRM);
  float* red = reinterpret_cast<float*>(lds + MM_OFF_RED);
  const float* gprm = reinterpret_cast<const float*>(p.pack + MM_PACK_FRAG_BYTES);
#ifdef KVQ_TAIL_TRACE   // diagnostic build only (tools/tail_trace.py): per-workgroup shader-clock stamps of wave 0
  const bool tr = p.trace && tid == 0 && (int)blockIdx.x < p.trace_blocks;
  unsigned long long wait_dma = 0, wait_bar = 0, t_gelu = 0, t_fc2 = 0, t_mark = 0;
#define MM_T0() t_mark = __builtin_readcyclecounter()
#define MM_T1(acc_) { const unsigned long long t1_ = __builtin_readcyclecounter(); acc_ += t1_ - t_mark; t_mark = t1_; }
#define MM_STAMP(i) if (tr) p.trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter()
#define MM_BARRIER() { const unsigned long long t0_ = __builtin_readcyclecounter(); __syncthreads(); wait_bar += __builtin_readcyclecounter() - t0_; }
#else
#define MM_STAMP(i)
#define MM_T0()
#define MM_T1(acc_)
#define MM_BARRIER() __syncthreads()
#endif
  MM_STAMP(0);

  // ---- the weight stream: `issued` fragments of this wave's list have been requested ----
  // (past the end of the list the LAST fragment is requested again, into a slot nobody reads any more: the vmcnt arithmetic
  // stays uniform and the loop bodies stay branch-free)
  int issued = 0;
  typename E::v8 wr[VR_R];
  constexpr int NFQ = QKV ? 3 * K::NF_PROJ : 0, NF_ALL = MM_NF + NFQ;      // QKV: the next block's qkv fragments follow the list
  const unsigned char* wq = QKV ? p.qkv_pack + (size_t)wave * NFQ * 1024 + lane * 16 : nullptr;
  auto vload = [&](int slot) __attribute__((always_inline)) {        // list position `issued` -> register slot (compile-time after unrolling)
    const int src = issued < NF_ALL ? issued : NF_ALL - 1;
    const unsigned char* a = QKV && src >= MM_NF ? wq + (size_t)(src - MM_NF) * 1024 : wsrc + (size_t)src * 1024;
    wr[slot] = *reinterpret_cast<const typename E::v8*>(a);
    ++issued;
  };

  // ---- this workgroup's rows: token tt*32 + j of 64, window order -> token of the residual stream ----
  long orig[TT], arow[TT];
  bool live[TT];
  int tloc_[TT], tb_[TT];
  const long nrows = p.gather ? p.n_tok : p.M;     // window rows, or (gather) tokens: no work on padding rows
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const long row = (long)blockIdx.x * MM_TOK + 32 * tt + j;
    const long rc = row < nrows ? row : nrows - 1;
    arow[tt] = rc;
    int tb, tloc;
    const unsigned rcu = (unsigned)rc;                 // 32-bit unsigned divisions: row counts are int32 at the boundary
    if (p.gather) {
      tb = (int)(rcu / (unsigned)p.out_rows);
      tloc = (int)(rcu - (unsigned)tb * (unsigned)p.out_rows);
      arow[tt] = (long)tb * p.map_rows + p.gather[tloc];
    } else if (p.map) {
      tb = (int)(rcu / (unsigned)p.map_rows);
      tloc = p.map[rcu - (unsigned)tb * (unsigned)p.map_rows];
    } else {
      tb = (int)(rcu / (unsigned)p.out_rows);
      tloc = (int)(rcu - (unsigned)tb * (unsigned)p.out_rows);
    }
    live[tt] = row < nrows && tloc >= 0;
    tloc = tloc < 0 ? 0 : tloc;
    tloc_[tt] = tloc; tb_[tt] = tb;
    orig[tt] = (long)tb * p.out_rows + tloc;
  }
  // ---- attention rows -> B fragments [k-step][token tile] by LDS-DMA (lane (j, half) fetches row j's k 16ks + 8 half ..+7) ----
  {
    for (int fr = wave; fr < MM_KS_C * TT; fr += 4) {
      const int ks = fr / TT, tt = fr % TT;
      long row = arow[0];
#pragma unroll
      for (int u = 1; u < TT; ++u) row = tt == u ? arow[u] : row;
      __builtin_amdgcn_global_load_lds((mm_gbl_t)(p.attn + (size_t)row * C + 16 * ks + 8 * half), (mm_lds_t)(lds + MM_OFF_X + fr * 1024), 16, 0, 0);
    }
    for (int q = wave; q < (MM_PRM_FLOATS * 4) / 1024; q += 4)      // b1 | g2 | b2n | proj_b | b2: 12 KB
      __builtin_amdgcn_global_load_lds((mm_gbl_t)((const unsigned char*)gprm + q * 1024 + lane * 16), (mm_lds_t)(lds + MM_OFF_PRM + q * 1024), 16, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < VR_PF; ++q) vload(q);          // VR_PF fragments of the weight list in flight from here on
  // ---- accumulators = x + proj bias: tile (ft, tt), register r <-> feature 96 wave + 32 ft + (r&3) + 8 (r>>2) + 4 half ----
  // all 24 row pieces of a lane are requested before anything waits (they queue behind the DMA requests above: one drain)
  f32x16 acc[CF][TT];
  {
    f32x4 xv[CF][4][TT];
    if (p.x16) {      // fp16 residual stream (round 6): 8 bytes per piece, raw in the first two registers until everything has landed
#pragma unroll
      for (int ft = 0; ft < CF; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            const uint64_t u = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint16_t*>(p.x) + (size_t)orig[tt] * C + FW * wave + 32 * ft + 8 * q + 4 * half);
            xv[ft][q][tt][0] = __uint_as_float((uint32_t)u); xv[ft][q][tt][1] = __uint_as_float((uint32_t)(u >> 32));
          }
    } else {
#pragma unroll
      for (int ft = 0; ft < CF; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            xv[ft][q][tt] = *reinterpret_cast<const f32x4*>(p.x + (size_t)orig[tt] * C + FW * wave + 32 * ft + 8 * q + 4 * half);
    }
    __builtin_amdgcn_sched_barrier(0);
    // everything requested so far has landed (the row loads were issued last: vmcnt(0) covers the DMA before them too)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MM_STAMP(1);
    if (p.x16) {
      typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
#pragma unroll
      for (int ft = 0; ft < CF; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            const f16x4_t hv = __builtin_bit_cast(f16x4_t, (uint64_t)__float_as_uint(xv[ft][q][tt][0]) | ((uint64_t)__float_as_uint(xv[ft][q][tt][1]) << 32));
            xv[ft][q][tt] = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
          }
    }
#pragma unroll
    for (int ft = 0; ft < CF; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 pb = *reinterpret_cast<const f32x4*>(prm + MM_H + 2 * MM_C + FW * wave + 32 * ft + 8 * q + 4 * half);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ft][tt][4 * q + i] = xv[ft][q][tt][i] + pb[i];
      }
  }

  // One GEMM phase: nk k-steps, NA weight tiles per wave; the weight fragments ARE the registers wr[.] (phase-local list position
  // q lives in slot q % VR_R), activation fragments come from `bbuf` ([k-step][2][1 KB]); mm(ft, tt, a, b) issues one MFMA.
  // A loop body covers KU k-steps (fc1 has only 4 MFMAs per k-step: two are paired); the activation fragments of body s+1 are read
  // and NR new weight fragments requested BETWEEN the MFMAs of body s, one at a time: a wave's LDS / VMEM issue hides under its own
  // MFMAs only ~10 cycles at a time (tools/ubench/pipe_share.hip) — a burst of 5 reads in front of 6 MFMAs does not hide at all.
  auto gemm_phase = [&](auto na_tag, auto nk_tag, const unsigned char* bbuf, auto&& mm, auto&& between) __attribute__((always_inline)) {
    // k-steps per body.  One weight tile per wave (fc1 at HC = 128) is ONE MFMA per activation fragment read and a body of TT MFMAs; bodies of
    // KU1 = 2 / 4 k-steps (the fragments requested 128 / 256 cycles ahead) were measured in round 6 and change nothing — alone on the chip or
    // on the 4-lane line (profiles/r06_tailmm_study.txt): the phase does not wait for its LDS reads.  -DKVQ_TAILMM_KU1=2 builds them.
    constexpr int NA = decltype(na_tag)::value, nk = decltype(nk_tag)::value, KU = NA == 2 ? 2 : NA == 1 ? K::KU1 : 1, NR = NA * KU;
    constexpr int NM = TT * NA * KU;                              // MFMAs per body
    static_assert(nk % KU == 0, "k-steps per body");
    V8 b[2][KU][TT];                     // activation fragments of two bodies: the one in use and the one being read
    auto rdb = [&](int buf, int ks, int q) __attribute__((always_inline)) {
      const int k = q / TT, tt = q % TT;
      b[buf][k][tt] = *reinterpret_cast<const V8*>(bbuf + ((ks + k) * TT + tt) * 1024 + lane * 16);
    };
    auto body_vr = [&](int s, bool more) __attribute__((always_inline)) {
      const int u = (s / KU) & 1, base = (s / KU) * NR;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int k = m / (TT * NA), t = (m % (TT * NA)) / TT, tt = m % TT;
        mm(t, tt, wr[(base + k * NA + t) % VR_R], b[u][k][tt]);
        if (more && m < TT * KU && !((MM_ABL & 1) && s >= 2 * KU)) rdb(u ^ 1, s + KU, m);
        if (m >= NM - NR && !(MM_ABL & 2)) vload((base + VR_PF + (m - (NM - NR))) % VR_R);
        between(s * TT * NA + m);                       // VALU work of the caller, placed between two MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll
    for (int q = 0; q < TT * KU; ++q) rdb(0, 0, q);
#pragma unroll
    for (int s = 0; s + KU < nk; s += KU) body_vr(s, true);
    body_vr(nk - KU, false);
  };
  using KC = std::integral_constant<int, MM_KS_C>;
  using KH = std::integral_constant<int, MM_KS_H>;
  auto nothing = [](int) {};
  using T2 = std::integral_constant<int, HT>;          // fc1 tiles per wave
  using T3 = std::integral_constant<int, CF>;          // weight tiles per wave in proj / fc2

  // ---- proj: acc (= x + bias) += Wp . attn^T -------------------------------------------------------------------------
  gemm_phase(T3{}, KC{}, lds + MM_OFF_X, [&](int ft, int tt, V8 a, V8 b) { acc[ft][tt] = E::mfma32(a, b, acc[ft][tt]); }, nothing);

  // ---- LayerNorm statistics of a token over the 4 waves' slices: in-lane + lane^32 + LDS exchange; two passes -----------
  auto token_sums = [&](auto&& term) __attribute__((always_inline)) {       // term(ft, tt, r) -> sum over all 384 features, per tt
    TV s;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float v = 0.f;
#pragma unroll
      for (int ft = 0; ft < CF; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) v += term(ft, tt, r);
      v += __shfl_xor(v, 32);
      s.v[tt] = v;
    }
    __syncthreads();                               // the previous use of `red` has been read by everybody
    if (half == 0) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) red[wave * MM_TOK + 32 * tt + j] = s.v[tt];
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      s.v[tt] = (red[0 * MM_TOK + 32 * tt + j] + red[1 * MM_TOK + 32 * tt + j]) + (red[2 * MM_TOK + 32 * tt + j] + red[3 * MM_TOK + 32 * tt + j]);
    return s;
  };
  // mean and 1 / sqrt(var + eps) of the lane's tokens from the accumulators (two passes, as the LayerNorm launch)
  auto token_stats = [&](TV& mean, TV& rstd) __attribute__((always_inline)) {
    const TV sum = token_sums([&](int ft, int tt, int r) { return acc[ft][tt][r]; });
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) mean.v[tt] = sum.v[tt] * (1.0f / (float)C);
    const TV sq = token_sums([&](int ft, int tt, int r) { const float d = acc[ft][tt][r] - mean.v[tt]; return d * d; });
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) rstd.v[tt] = rsqrtf(sq.v[tt] / (float)C + p.eps);
  };
  // (acc - mean) * rstd * gamma + beta, 16-bit, written as B fragments: tile ft covers k-steps kbase + 2 ft + {0, 1}
  auto write_norm = [&](const TV& mean, const TV& rstd, const float* gam, const float* bet, unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int ft = 0; ft < CF; ++ft)
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {               // quad pair (2 hp, 2 hp + 1) = 16 features = one k-step
        const int f0 = FW * wave + 32 * ft + 16 * hp + 4 * half;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam + f0), g1 = *reinterpret_cast<const f32x4*>(gam + f0 + 8);
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(bet + f0), e1 = *reinterpret_cast<const f32x4*>(bet + f0 + 8);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          float y[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {                // two fma per value: (x rstd - mean rstd) g + b
            const float nmr = -mean[tt] * rstd[tt];
            y[i] = fmaf(fmaf(acc[ft][tt][8 * hp + i], rstd[tt], nmr), g0[i], e0[i]);
            y[4 + i] = fmaf(fmaf(acc[ft][tt][8 * hp + 4 + i], rstd[tt], nmr), g1[i], e1[i]);
          }
          const u32x4 w = {E::pack2(y[0], y[1]), E::pack2(y[2], y[3]), E::pack2(y[4], y[5]), E::pack2(y[6], y[7])};
          const int ks = 2 * CF * wave + 2 * ft + hp;
          *reinterpret_cast<u32x4*>(dst + (ks * TT + tt) * 1024 + lane * 16) = w;
        }
      }
  };
  {
    TV mean, rstd;
    token_stats(mean, rstd);
    // every wave is past its last read of the attention tile (two barriers ago): norm2 rows take its place
    write_norm(mean, rstd, prm + MM_H, prm + MM_H + MM_C, lds + MM_OFF_X);
  }
  // acc becomes the fc2 accumulator: x_mid + fc2 bias
#pragma unroll
  for (int ft = 0; ft < CF; ++ft)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(prm + MM_H + 3 * MM_C + FW * wave + 32 * ft + 8 * q + 4 * half);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ft][tt][4 * q + i] += b2[i];
    }
  MM_BARRIER();                                      // norm2 rows complete
  MM_STAMP(2);

  // ---- MLP over 6 chunks of 256 hidden units: fc1 (wave: 64 units x 64 tokens) -> GELU -> LDS -> fc2 partial.  Software
  // pipeline: fc1 of chunk c+1 runs first, then the GELU of chunk c+1 is evaluated BETWEEN the MFMAs of fc2(chunk c) — one pair
  // per three MFMAs — so that the VALU stream no longer stops the weight stream ----------
  f32x16 hacc[HT][TT];
  u32x4 gp[HT][2][TT];                                // GELU outputs of a chunk, packed: [weight tile][k-step half][token tile]
  auto fc1 = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int ft = 0; ft < HT; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(prm + MM_HC * c + 32 * HT * wave + 32 * ft + 8 * q + 4 * half);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int i = 0; i < 4; ++i) hacc[ft][tt][4 * q + i] = b1[i];
      }
    gemm_phase(T2{}, KC{}, lds + MM_OFF_X, [&](int ft, int tt, V8 a, V8 b) {
      if (MM_ABL & 16) { asm volatile("" :: "v"(a), "v"(b)); return; }
      hacc[ft][tt] = E::mfma32(a, b, hacc[ft][tt]); }, nothing);
  };
  auto gelu_pair = [&](int pi) __attribute__((always_inline)) {      // pi = ((ft * 2 + hp) * TT + tt) * 4 + i, 0..8 TT HT - 1
    const int i = pi & 3, tt = (pi >> 2) % TT, hp = ((pi >> 2) / TT) & 1, ft = (pi >> 2) / (2 * TT);
    const int r = 8 * hp + 2 * (i & 1) + 4 * (i >> 1);              // pairs (r, r+1): i = 0,1 -> quad 2hp; i = 2,3 -> quad 2hp+1
    uint32_t w = (MM_ABL & 8) ? E::pack2(hacc[ft][tt][r], hacc[ft][tt][r + 1]) : gelu_pack2<E>(hacc[ft][tt][r], hacc[ft][tt][r + 1]);
    asm volatile("" : "+v"(w));                        // pins the evaluation where it is placed
    gp[ft][hp][tt][i] = w;
  };
  auto write_gelu = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ft = 0; ft < HT; ++ft)
#pragma unroll
      for (int hp = 0; hp < 2; ++hp)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
          *reinterpret_cast<u32x4*>(lds + MM_OFF_G + ((2 * HT * wave + 2 * ft + hp) * TT + tt) * 1024 + lane * 16) = gp[ft][hp][tt];
  };
  if constexpr (!K::PIPE) {
    // two workgroups per CU: the other workgroup's MFMAs run under this one's GELU, so the chunks are NOT software-pipelined in the
    // wave (fc1(c + 1) ahead of fc2(c) keeps acc + hacc + the packed GELU outputs + the ring + two bodies of fragments live at
    // once: 64 registers past the 256 of two waves per SIMD, spilled and reloaded in the loop)
    for (int c = 0; c < MM_NCH; ++c) {
      MM_T0();
      fc1(c);
      MM_T1(wait_dma);                                 // (trace builds: slot 5 = cycles in fc1, slot 7 = in the GELU, slot 6 = in barriers)
#pragma unroll
      for (int pi = 0; pi < 8 * TT * HT; ++pi) gelu_pair(pi);
      MM_T1(t_gelu);
      if (c > 0) MM_BARRIER();                         // everybody has finished fc2 of chunk c - 1: its GELU rows may go
      write_gelu();
      MM_BARRIER();                                    // GELU rows of chunk c complete
      MM_T0();
      gemm_phase(T3{}, KH{}, lds + MM_OFF_G, [&](int ft, int tt, V8 a, V8 b) {
        if (MM_ABL & 32) { asm volatile("" :: "v"(a), "v"(b)); return; }
        acc[ft][tt] = E::mfma32(a, b, acc[ft][tt]); }, nothing);
      MM_T1(t_fc2);
    }
  } else {
    static_assert(TT == 2, "the software-pipelined chunks are the 64-token form");
    fc1(0);
  #pragma unroll
    for (int pi = 0; pi < 16 * HT; ++pi) gelu_pair(pi);
    write_gelu();
    MM_BARRIER();                                      // GELU rows of chunk 0 complete
    for (int c = 0; c < MM_NCH; ++c) {
      const bool more = c + 1 < MM_NCH;
      if (more) {
        fc1(c + 1);
        gemm_phase(T3{}, KH{}, lds + MM_OFF_G, [&](int ft, int tt, V8 a, V8 b) { acc[ft][tt] = E::mfma32(a, b, acc[ft][tt]); },
                   [&](int m) { if (m % CF == 0) gelu_pair(m / CF); });      // 16 HT pairs over the 16 HT CF MFMAs
        MM_BARRIER();                                  // everybody has finished fc2 of chunk c: its GELU rows may go
        write_gelu();
        MM_BARRIER();                                  // GELU rows of chunk c + 1 complete
      } else {
        gemm_phase(T3{}, KH{}, lds + MM_OFF_G, [&](int ft, int tt, V8 a, V8 b) { acc[ft][tt] = E::mfma32(a, b, acc[ft][tt]); }, nothing);
      }
    }
  }

  MM_STAMP(3);
  // ---- write the residual stream back; optionally the next block's norm1 rows in ITS window order ----------------------
  if (p.x16) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      if (live[tt]) {
        uint16_t* xr = reinterpret_cast<uint16_t*>(p.x) + (size_t)orig[tt] * C + FW * wave + 4 * half;
#pragma unroll
        for (int ft = 0; ft < CF; ++ft)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<u32x2*>(xr + 32 * ft + 8 * q) = (u32x2){Fp16::pack2(acc[ft][tt][4 * q], acc[ft][tt][4 * q + 1]), Fp16::pack2(acc[ft][tt][4 * q + 2], acc[ft][tt][4 * q + 3])};
      }
  } else
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
    if (live[tt]) {
      float* xr = p.x + (size_t)orig[tt] * C + FW * wave + 4 * half;
#pragma unroll
      for (int ft = 0; ft < CF; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(xr + 32 * ft + 8 * q) = (f32x4){acc[ft][tt][4 * q], acc[ft][tt][4 * q + 1], acc[ft][tt][4 * q + 2], acc[ft][tt][4 * q + 3]};
    }
  if (QKV) {
    // ---- the next block's q | k | v (swin_backbone.py:252-260 of block b + 1): norm1 rows -> B fragments in the activation tile (every
    // wave is past its last read of the norm2 rows: the barrier behind the last fc1), then three passes of the proj-shaped GEMM phase
    // over the wave's feature slice of q, k and v; a 32-feature tile is one head.  Rows leave head-major in the next block's window order.
    TV mean, rstd;
    token_stats(mean, rstd);
    write_norm(mean, rstd, p.nn_w, p.nn_b, lds + MM_OFF_X);
    long drow[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) drow[tt] = (long)tb_[tt] * p.next_rows + p.next_dst[tloc_[tt]];
    MM_BARRIER();                                      // norm1 rows complete
#pragma unroll 1
    for (int which = 0; which < 3; ++which) {
#pragma unroll
      for (int ft = 0; ft < CF; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 qb = *reinterpret_cast<const f32x4*>(p.qkv_b + which * C + FW * wave + 32 * ft + 8 * q + 4 * half);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ft][tt][4 * q + i] = qb[i];
        }
      gemm_phase(T3{}, KC{}, lds + MM_OFF_X, [&](int ft, int tt, V8 a, V8 b) { acc[ft][tt] = E::mfma32(a, b, acc[ft][tt]); }, nothing);
      const float sc = which == 0 ? p.q_scale : 1.f;
      // as the norm1 rows below: the lane pair of a token swaps 8-byte pieces, lane `half` then owns head dims 8 (2 t + half) .. + 7
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int ft = 0; ft < CF; ++ft) {
          uint16_t* o = p.qkv_out + ((size_t)(which * p.num_heads + CF * wave + ft) * p.qkv_rows + (size_t)drow[tt]) * 32;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            uint32_t pk[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int q = 2 * t + u;
              pk[u][0] = E::pack2(acc[ft][tt][4 * q] * sc, acc[ft][tt][4 * q + 1] * sc);
              pk[u][1] = E::pack2(acc[ft][tt][4 * q + 2] * sc, acc[ft][tt][4 * q + 3] * sc);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            if (live[tt]) *reinterpret_cast<u32x4*>(o + 8 * (2 * t + half)) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
          }
        }
    }
  }
  if (EMIT) {
    TV mean, rstd;
    token_stats(mean, rstd);
    // 16 bytes per lane: the lane pair (half = 0 | 1) of a token exchanges the 8-byte pieces of (q, q + 1) by v_permlane32_swap, lane
    // `half` then owns features 8 (2 t + half) .. + 7 of a tile — half the row-divergent store instructions
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const long drow = (long)tb_[tt] * p.next_rows + p.next_dst[tloc_[tt]];
      uint16_t* o = p.next_ln + (size_t)drow * C + FW * wave;
#pragma unroll
      for (int ft = 0; ft < CF; ++ft)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          uint32_t pk[2][2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int q = 2 * t + u;
            const int f0 = FW * wave + 32 * ft + 8 * q + 4 * half;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.nn_w + f0), be = *reinterpret_cast<const f32x4*>(p.nn_b + f0);
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = fmaf(fmaf(acc[ft][tt][4 * q + i], rstd[tt], -mean[tt] * rstd[tt]), gm[i], be[i]);
            pk[u][0] = E::pack2(y[0], y[1]);
            pk[u][1] = E::pack2(y[2], y[3]);
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
          if (live[tt]) *reinterpret_cast<u32x4*>(o + 32 * ft + 8 * (2 * t + half)) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
        }
    }
  }
#ifdef KVQ_TAIL_TRACE
  if (tr) {
    __builtin_amdgcn_s_waitcnt(0);
    p.trace[blockIdx.x * 8 + 5] = wait_dma;
    p.trace[blockIdx.x * 8 + 6] = wait_bar;
    p.trace[blockIdx.x * 8 + 7] = t_fc2;             // (the GELU's VALU work has no memory dependence: the compiler places it behind the clock read)
  }
#endif
  MM_STAMP(4);
}

template <typename E, int CF, int HC = 256, int TT = 2>
static int launch_mm_cf(const TailParams& p, hipStream_t st) {
  constexpr int LDS = MMc<CF, HC, TT>::LDS;
  dim3 grid((unsigned)ceil_div(p.gather ? p.n_tok : p.M, MMc<CF, HC, TT>::TOK)), block(256);
  auto go = [&](auto k) -> int {
    LdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(k), LDS)) return rc;
    hipLaunchKernelGGL(k, grid, block, LDS, st, p);
    return KVQ_OK;
  };
  int rc;
  if (p.qkv_out) rc = go(block_tailmm_kernel<E, 2, CF, HC, TT>);
  else if (p.next_ln) rc = go(block_tailmm_kernel<E, 1, CF, HC, TT>);
  else rc = go(block_tailmm_kernel<E, 0, CF, HC, TT>);
  if (rc) return rc;
  KVQ_CHECK_LAUNCH("block_tailmm_kernel");
  return KVQ_OK;
}

// the forms, grouped by translation unit (see the top of the file)
int tailmm_launch_part0(int form, const TailParams& p, int dtype, hipStream_t st);
int tailmm_launch_part1(int form, const TailParams& p, int dtype, hipStream_t st);
int tailmm_launch_part2(int form, const TailParams& p, int dtype, hipStream_t st);
int tailmm_launch_part3(int form, const TailParams& p, int dtype, hipStream_t st);
enum { MM_F_C512_H128, MM_F_C512_H256, MM_F_C768, MM_F_C256, MM_F_C384_T128, MM_F_C384_H128, MM_F_C384_H256 };
#define MM_GO(CF, HC, TT) (dtype == KVQ_DT_FP16 ? launch_mm_cf<Fp16, CF, HC, TT>(p, st) : launch_mm_cf<Bf16, CF, HC, TT>(p, st))
#ifndef KVQ_TAILMM_FOCUS
#if MM_PART_HERE(0)
int tailmm_launch_part0(int form, const TailParams& p, int dtype, hipStream_t st) { return MM_GO(3, 128, 2); }                                 // C = 384: the C2 line's form
#endif
#if MM_PART_HERE(1)
int tailmm_launch_part1(int form, const TailParams& p, int dtype, hipStream_t st) { return form == MM_F_C384_H256 ? MM_GO(3, 256, 2) : MM_GO(2, 256, 2); }
#endif
#if MM_PART_HERE(2)
int tailmm_launch_part2(int form, const TailParams& p, int dtype, hipStream_t st) { return form == MM_F_C512_H128 ? MM_GO(4, 128, 2) : MM_GO(4, 256, 2); }
#endif
#if MM_PART_HERE(3)
int tailmm_launch_part3(int form, const TailParams& p, int dtype, hipStream_t st) { return form == MM_F_C768 ? MM_GO(6, 128, 2) : MM_GO(3, 128, 4); }
#endif
#endif

#if MM_PART_HERE(0)
int tailmm_geometry_code(int C, int hidden) {
  if (!tailmm_supported(C, hidden)) return 0;
  const int hc = C == 768 ? 128 : (C == 384 || C == 512) ? tailmm_hc(C) : 256;
  return (hc / 128) * 10 + (hc == 128 && tailmm_tok(C) == 128 ? 4 : 2);
}

int tailmm_launch(const TailParams& p, int C, int dtype, hipStream_t st) {
  KVQ_REQUIRE(tailmm_supported(C, p.hidden), KVQ_ERR_UNSUPPORTED, "kvq_block_tail: C=%d hidden=%d", C, p.hidden);
#ifdef KVQ_TAILMM_FOCUS     // compile-time study builds (register / ISA inspection of one form in seconds instead of minutes), e.g. 3102: CF 3, HC 128, TT 2
  return launch_mm_cf<Fp16, KVQ_TAILMM_FOCUS / 1000, (KVQ_TAILMM_FOCUS / 10) % 100 * 128 / 10, KVQ_TAILMM_FOCUS % 10>(p, st);
#else
  if (C == 512) return tailmm_launch_part2(tailmm_hc(C) == 128 ? MM_F_C512_H128 : MM_F_C512_H256, p, dtype, st);
  if (C == 768) return tailmm_launch_part3(MM_F_C768, p, dtype, st);
  if (C == 256) return tailmm_launch_part1(MM_F_C256, p, dtype, st);
  if (tailmm_hc(C) == 128 && tailmm_tok(C) == 128) return tailmm_launch_part3(MM_F_C384_T128, p, dtype, st);
  if (tailmm_hc(C) == 128) return tailmm_launch_part0(MM_F_C384_H128, p, dtype, st);
  return tailmm_launch_part1(MM_F_C384_H256, p, dtype, st);
#endif
}
#endif  // MM_PART_HERE(0)

}  // namespace kvq
