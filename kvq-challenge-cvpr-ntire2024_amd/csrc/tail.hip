// Fused post-attention half of a SwinTransformerBlock3D (gfx950), one launch per block:
//
//   x  <- x + window_reverse(roll(proj(attn_out)))                (swin_backbone.py:323, :472-488, :509)
//   x  <- x + fc2(GELU(fc1(norm2(x))))                            (:490-491, :514, Mlp :84-87)
//   [ ln_next <- window_partition(roll(norm1_next(x))) ]          (:416-449 of the NEXT block)
//
// Everything here is row-local, so the kernel is written TOKEN-PER-LANE: all GEMMs are computed transposed,
//   Out^T[channel][token] = W[channel][k] * In^T[k][token],
// with the weights as the MFMA A operand (from LDS) and 32 tokens as the 32 columns of a 32x32x16 MFMA.  In the
// C/D layout a lane then owns ONE token (column lane&31) and the channel rows (r&3) + 8*(r>>2) + 4*(lane>>5) of
// every 32-channel tile: the residual add, LayerNorm statistics (in-lane sums + one exchange with lane^32), bias
// and GELU all happen in registers, and an accumulator tile converts to the B operand of the next GEMM by a
// 16-bit pack alone — the k order of that operand is a fixed permutation of the channels, which is folded into
// the weights once on the host side of the boundary (kvq_block_tail_pack).  The hidden activations (4C per
// token), the proj output and norm2's output never exist in memory: per token the launch reads attn_out (2C B)
// and x (4C B) and writes x (4C B) [+ 2C B for the next norm1] instead of 48C B over five launches.
//
// Weights stream through an LDS ring by LDS-DMA as "panels" (32 x C 16-bit: 32 output channels of proj, the W1
// rows of 32 hidden units, or the W2 columns of 32 hidden units) already laid out fragment-major (one 1 KB
// wave-load = one k-step of A fragments, lane-linear => conflict-free ds_read_b128); a ring item is 2 panels.
// One counted vmcnt wait + one raw barrier per item, as in gemm.hip.
//
// The MLP is bound by the GELU's VALU work, not by MFMA, so its loop is software-pipelined over 32-unit chunks:
// iteration j issues the MFMAs of fc2(chunk j) and fc1(chunk j+2) and, between them, evaluates GELU(chunk j+1)
// — the matrix pipe runs under the VALU stream of the SAME wave (two fc1 accumulators and two packed GELU
// outputs rotate).  Ring item j is therefore {W2 panel j, W1 panel j+2}.
#include <stdlib.h>

#include "common.hpp"
#include "tail.hpp"

namespace kvq {


typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__host__ __device__ constexpr int tail_slot_bytes(int C) { return 128 * C; }            // 2 panels of 32 x C 16-bit
__host__ __device__ constexpr int tail_proj_items(int C) { return (C / 32 + 1) / 2; }
// workgroups per CU for NW waves of 32 tokens each: C=96 fits 3 waves per SIMD (<= 168 VGPRs), wider rows 2
__host__ __device__ constexpr int tail_bpc(int C, int NW) { return (C == 96 ? 12 : 8) / NW; }
__host__ __device__ constexpr int tail_ring(int C, int NW) {                            // ring slots: what LDS allows, <= 4
  return (163840 / tail_bpc(C, NW) - 8192) / tail_slot_bytes(C) >= 4 ? 4 : 3;
}
// Row staging (round 5): the x / attention / norm1 rows of a wave travel global <-> LDS as 128-byte row segments (8 rows per wave-load:
// 8 cache lines per access instead of 32 lane-rows) through a 4 KB tile per wave and are transposed there to the token-per-lane layout.
// Prologue: the tile lives in the LAST ring slot (empty until the first item is consumed) + TAIL_STAGE_EXTRA bytes behind the ring where
// a slot is smaller than the four tiles; epilogue: in the slots the ring has left.
constexpr int TAIL_STG = 4096;
__host__ __device__ constexpr int tail_stage_extra(int C, int NW) { return NW * TAIL_STG > tail_slot_bytes(C) ? NW * TAIL_STG - tail_slot_bytes(C) : 0; }
static size_t tail_items(int C, int hidden) { return (size_t)tail_proj_items(C) + 1 + hidden / 32; }
static size_t tail_param_bytes(int C, int hidden) { return (((size_t)(4 * C + hidden) * 4) + 4095) & ~(size_t)4095; }

// ---- weight image ---------------------------------------------------------------------------------------------
// panel = 64*C bytes = C/16 "fragment rows" of 64 lanes x 16 B.  Fragment row f of a panel holds, for lane
// (m = lane&31, h = lane>>5), the 8 k-values an MFMA A operand needs at that lane.
//   proj panel i (i < C/32):  f = s (k-step):                Wp[32i+m][16s + 8h + e]            (e = 0..7)
//   W1 panel j (j < hid/32):  f = s:                         W1[32j+m][chan(s,h,e)]
//   W2 panel j:               f = 2i + t (t = 0,1):          W2[32i+m][32j + 8(2t + (e>>2)) + 4h + (e&3)]
//   chan(s,h,e) = 32(s>>1) + 8(2(s&1) + (e>>2)) + 4h + (e&3)   — the channel an accumulator register r = 8(s&1)+e
//   of tile s>>1 holds in lane half h (C/D layout of v_mfma_f32_32x32x16).
// Items (2 panels each): proj panels in order (zero-padded to a whole item); {W1_0, W1_1}; then for every j
// {W2_j, W1_{j+2}} (zero where j+2 is past the end).  After the items: fp32 [proj_b C][norm2_w C][norm2_b C]
// [fc2_b C][fc1_b hidden], padded to 4 KB.
__global__ void tail_pack_kernel(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b,
                                 const float* n2w, const float* n2b, const float* b1, const float* b2, int C, int hidden,
                                 unsigned char* out, long n_chunks, long n_par) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CM = C / 32, KS = C / 16, NJ = hidden / 32, first_mlp = tail_proj_items(C) * 2;
  if (g < n_chunks) {
    const int panel = (int)(g / (KS * 64)), rem = (int)(g % (KS * 64));
    const int f = rem >> 6, lane = rem & 63, m = lane & 31, h = lane >> 5;
    int kind = 0, idx = 0;             // 0 zero, 1 proj, 2 W1, 3 W2
    if (panel < CM) {
      kind = 1; idx = panel;
    } else if (panel >= first_mlp) {
      const int q = panel - first_mlp;
      if (q < 2) {
        kind = 2; idx = q;
      } else if (((q - 2) & 1) == 0) {
        kind = 3; idx = (q - 2) >> 1;
      } else if (((q - 2) >> 1) + 2 < NJ) {
        kind = 2; idx = ((q - 2) >> 1) + 2;
      }
    }
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint16_t val = 0;
      if (kind == 1) {
        val = wp[(size_t)(32 * idx + m) * C + 16 * f + 8 * h + e];
      } else if (kind == 2) {
        const int ch = 32 * (f >> 1) + 8 * (2 * (f & 1) + (e >> 2)) + 4 * h + (e & 3);
        val = w1[(size_t)(32 * idx + m) * C + ch];
      } else if (kind == 3) {
        const int i = f >> 1, t = f & 1;
        val = w2[(size_t)(32 * i + m) * hidden + 32 * idx + 8 * (2 * t + (e >> 2)) + 4 * h + (e & 3)];
      }
      v[e] = val;
    }
    uint16_t* o = reinterpret_cast<uint16_t*>(out + g * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e];
  } else if (g < n_chunks + n_par) {
    const int q = (int)(g - n_chunks);
    float val = 0.f;
    if (q < C) val = proj_b[q];
    else if (q < 2 * C) val = n2w[q - C];
    else if (q < 3 * C) val = n2b[q - 2 * C];
    else if (q < 4 * C) val = b2[q - 3 * C];
    else if (q < 4 * C + hidden) val = b1[q - 4 * C];
    reinterpret_cast<float*>(out + n_chunks * 16)[q] = val;
  }
}

// ---- the kernel -----------------------------------------------------------------------------------------------
// NW waves of 32 tokens share one weight ring.  Measured (B = 4 clips, stage 0 / stage 1, us per launch): NW = 4
// 82 / 76, NW = 6 136 / -, NW = 8 - / 76, NW = 12 124 / -: the L2 -> LDS weight stream (1.7 KB per token at NW = 4) is
// NOT the bound — the wider barrier domain costs more than the halved stream saves — so workgroups stay at 4 waves,
// 3 (C = 96, <= 168 VGPRs) or 2 of them per CU.
template <typename E, int CM, int NW, int MODE, bool STAGED>      // MODE 0: x only; 1: + the next block's norm1 rows; 2: + the next block's q | k | v
__global__ __launch_bounds__(64 * NW, tail_bpc(32 * CM, NW)) void block_tail_kernel(TailParams p) {
  constexpr bool EMIT = MODE != 0, QKV = MODE == 2;
  fp16_saturate_mode();
  constexpr int C = 32 * CM, KS = 2 * CM, PANEL = 64 * C, SLOT = 2 * PANEL, LPW = SLOT / 1024 / NW;
  constexpr int NST = tail_ring(C, NW), NPI = tail_proj_items(C);
  static_assert(SLOT % (1024 * NW) == 0, "an item is a whole number of 1 KB loads per wave");
  static_assert(NST == 3 || NST == 4, "wait ladder below");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E::v8;
  constexpr int XTRA = STAGED ? tail_stage_extra(C, NW) : 0, PRM_OFF = NST * SLOT + XTRA;      // the lane-per-row form asks for no staging bytes
  float* prm = reinterpret_cast<float*>(lds + PRM_OFF);
  const float* s_pb = prm;
  const float* s_g2 = prm + C;
  const float* s_b2n = prm + 2 * C;
  const float* s_fb2 = prm + 3 * C;
  const float* s_fb1 = prm + 4 * C;

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NJ = p.hidden >> 5, NI = NPI + 1 + NJ;
  constexpr int NQI = QKV ? (3 * C / 32) / 2 : 0;      // QKV: the next block's qkv weight as 3C / 32 more panels (W1 layout), two per item, in p.qkv_pack
  const int NI_ALL = NI + NQI;
  const int NQ = ((4 * C + p.hidden) * 4 + 1023) >> 10;              // 1 KB wave-loads that carry the parameters
  float* s_nn = reinterpret_cast<float*>(lds + PRM_OFF + NQ * 1024);   // [norm1_next_w C][norm1_next_b C] (EMIT)
#ifdef KVQ_TAIL_TRACE   // diagnostic build only: the stamps cost registers and scheduling freedom
  const bool tr = p.trace && tid == 0 && (int)blockIdx.x < p.trace_blocks;
#define KVQ_STAMP(i) if (tr) p.trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter()
#else
#define KVQ_STAMP(i)
#endif
  KVQ_STAMP(0);

  // oldest in the queue: the next block's norm1 vectors (registers -> LDS below) and the fp32 parameters (DMA)
  f32x4 nn_reg = {0.f, 0.f, 0.f, 0.f};
  if (EMIT && tid < C / 2) nn_reg = *reinterpret_cast<const f32x4*>((tid < C / 4 ? p.nn_w : p.nn_b - C) + 4 * tid);
  {
    const unsigned char* src = p.pack + (size_t)NI * SLOT;
    for (int q = wave; q < NQ; q += NW)      // uneven per wave is fine: these are OLDER than every counted item
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + q * 1024 + lane * 16),
                                       (lds_ptr_t)(lds + PRM_OFF + q * 1024), 16, 0, 0);
  }
  // this lane's token: row = a window row (through `map` to its token) or, with `gather`, a token (through it to its window row)
  const long row = (long)blockIdx.x * (32 * NW) + wave * 32 + (lane & 31);
  const long nrows = p.gather ? p.n_tok : p.M;
  long rc = row < nrows ? row : nrows - 1;
  int tb, tloc;
  // 32-bit unsigned divisions (row counts are int32 at the boundary): a 64-bit one is ~100 VALU instructions of a VALU-bound launch
  const unsigned rcu = (unsigned)rc;
  if (p.gather) {
    tb = (int)(rcu / (unsigned)p.out_rows);
    tloc = (int)(rcu - (unsigned)tb * (unsigned)p.out_rows);
    rc = (long)tb * p.map_rows + p.gather[tloc];           // the attention row of this token
  } else if (p.map) {
    tb = (int)(rcu / (unsigned)p.map_rows);
    tloc = p.map[rcu - (unsigned)tb * (unsigned)p.map_rows];
  } else {
    tb = (int)(rcu / (unsigned)p.out_rows);
    tloc = (int)(rcu - (unsigned)tb * (unsigned)p.out_rows);
  }
  const bool live = row < nrows && tloc >= 0;
  tloc = tloc < 0 ? 0 : tloc;
  const long orig = (long)tb * p.out_rows + tloc;
  V8 bx[KS];
  f32x16 acc[CM];
  // row-major role of this lane in a staged 128-byte-per-row tile: row 8 g + rrow (g = 0..3), 16-byte piece rp; piece p of row r sits
  // in slot 8 r + (p ^ ((r >> 1) & 7)) — both the row-major and the token-per-lane accesses are bank-conflict-free
  const int rrow = lane >> 3, rp = lane & 7, tj = lane & 31;
  auto stg_at = [](unsigned char* base, int row, int piece) __attribute__((always_inline)) -> unsigned char* {
    return base + ((row * 8 + (piece ^ ((row >> 1) & 7))) << 4);
  };
  if (!STAGED) {
    const uint16_t* ar = p.attn + (size_t)rc * C + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) bx[s] = *reinterpret_cast<const V8*>(ar + 16 * s);
    // acc = x + proj bias (from the global image: LDS is not populated yet).  From here to the end of the MLP the
    // accumulator tiles are only READ element-wise (norm2) or written by MFMA: element-wise updates in between
    // would make the register allocator shuffle / spill the 16-register tuples.
    const float* xr = p.x + (size_t)orig * C + 4 * h;
    const float* pbg = reinterpret_cast<const float*>(p.pack + (size_t)NI * SLOT) + 4 * h;
    if (p.x16) {
      // fp16 residual stream (round 6): 8 bytes per lane and quad instead of 16; every piece requested before the first is widened
      typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
      const uint16_t* xr16 = reinterpret_cast<const uint16_t*>(p.x) + (size_t)orig * C + 4 * h;
      f16x4_t u[CM][4];
#pragma unroll
      for (int i = 0; i < CM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) u[i][q] = *reinterpret_cast<const f16x4_t*>(xr16 + 32 * i + 8 * q);
#pragma unroll
      for (int i = 0; i < CM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(pbg + 32 * i + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = (float)u[i][q][e] + b[e];
        }
    } else {
#pragma unroll
      for (int i = 0; i < CM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 32 * i + 8 * q);
          const f32x4 b = *reinterpret_cast<const f32x4*>(pbg + 32 * i + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = v[e] + b[e];
        }
    }
  }

  auto issue = [&](int item, int slot) {
    unsigned char* dst = lds + slot * SLOT;
    const unsigned char* src = QKV && item >= NI ? p.qkv_pack + (size_t)(item - NI) * SLOT : p.pack + (size_t)item * SLOT;
#pragma unroll
    for (int l = 0; l < LPW; ++l) {
      const int q = l * NW + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + q * 1024 + lane * 16), (lds_ptr_t)(dst + q * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue(i, i);
  if (EMIT && tid < C / 2) *reinterpret_cast<f32x4*>(s_nn + 4 * tid) = nn_reg;
  if (STAGED) {
    // the wave's 4 KB tile in the last ring slot (+ the extra bytes behind the ring): free until the first item has been consumed — the
    // first next_item() below fills that slot only after its barrier, which every wave reaches with its prologue behind it
    unsigned char* stg = lds + (NST - 1) * SLOT + wave * TAIL_STG;
    int rowx[4], rowa[4];          // residual-stream row / attention row of the tile rows 8 g + rrow
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      rowx[g] = __shfl((int)orig, 8 * g + rrow);
      rowa[g] = __shfl((int)rc, 8 * g + rrow);
    }
    // ---- attention rows (16-bit): 64 channels = 128 bytes of a row per tile = 4 k-steps; C % 64 == 32: a last half tile ----
    constexpr int AT = C / 64, AH = (C % 64) / 32;
    u32x4 va[AT + AH][4];
#pragma unroll
    for (int t = 0; t < AT + AH; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        va[t][g] = (u32x4){0u, 0u, 0u, 0u};
        if (t < AT || rp < 4) va[t][g] = *reinterpret_cast<const u32x4*>(p.attn + (size_t)rowa[g] * C + 64 * t + 8 * rp);
      }
    // ---- residual rows (fp32): 32 channels = 128 bytes per tile ----
    const float* pbg = reinterpret_cast<const float*>(p.pack + (size_t)NI * SLOT) + 4 * h;
    constexpr int XB = CM > 3 ? 3 : CM;                 // tiles requested at once (12 x 16 B per lane in flight)
#pragma unroll
    for (int i0 = 0; i0 < CM; i0 += XB) {
      f32x4 vx[XB][4];
#pragma unroll
      for (int i = i0; i < i0 + XB && i < CM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          vx[i - i0][g] = *reinterpret_cast<const f32x4*>(p.x + (size_t)rowx[g] * C + 32 * i + 4 * rp);
        }
      if (i0 == 0) {
#pragma unroll
        for (int t = 0; t < AT + AH; ++t) {
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x4*>(stg_at(stg, 8 * g + rrow, rp)) = va[t][g];
#pragma unroll
          for (int k = 0; k < (t < AT ? 4 : 2); ++k)
            bx[4 * t + k] = __builtin_bit_cast(V8, *reinterpret_cast<const u32x4*>(stg_at(stg, tj, 2 * k + h)));
        }
      }
#pragma unroll
      for (int i = i0; i < i0 + XB && i < CM; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(stg_at(stg, 8 * g + rrow, rp)) = vx[i - i0][g];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(stg_at(stg, tj, 2 * q + h));
          const f32x4 b = *reinterpret_cast<const f32x4*>(pbg + 32 * i + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = v[e] + b[e];
        }
      }
    }
  }
  int it = 0, slot = 0;
  // item `it` visible to every wave; everybody has left item it-1, whose slot takes item it+NST-1
#ifdef KVQ_TAIL_TRACE
  unsigned long long wait_dma = 0, wait_bar = 0;
#endif
  auto next_item = [&]() -> const unsigned char* {
    const int younger = NI_ALL - 1 - it;
#ifdef KVQ_TAIL_TRACE
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    if (NST == 4 && younger >= 2) wait_vmcnt<2 * LPW>();
    else if (younger >= 1) wait_vmcnt<LPW>();
    else wait_vmcnt<0>();
#ifdef KVQ_TAIL_TRACE
    const unsigned long long t1 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef KVQ_TAIL_TRACE
    wait_dma += t1 - t0;
    wait_bar += __builtin_readcyclecounter() - t1;
#endif
    const int fill = slot == 0 ? NST - 1 : slot - 1;
    if (it + NST - 1 < NI_ALL) issue(it + NST - 1, fill);
    const unsigned char* st = lds + slot * SLOT + lane * 16;
    ++it;
    slot = slot + 1 == NST ? 0 : slot + 1;
    return st;
  };

  // ---- proj: acc (= x) += Wp . attn^T ------------------------------------------------------------------------
  const unsigned char* st = nullptr;
#pragma unroll
  for (int i = 0; i < CM; ++i) {
    if (i % 2 == 0) st = next_item();
    if (i == 0) { KVQ_STAMP(1); }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const V8 a = *reinterpret_cast<const V8*>(st + (i % 2) * PANEL + s * 1024);
      acc[i] = E::mfma32(a, bx[s], acc[i]);
      if (s % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // keeps the fragment loads from piling up in registers
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- norm2 in registers: two-pass statistics, biased variance, eps inside the rsqrt (as ln.hip) -----------
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 4) s += (acc[i][r] + acc[i][r + 1]) + (acc[i][r + 2] + acc[i][r + 3]);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[i][r] - mean;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq / (float)C + p.eps);
    float mean_n = mean;
    asm volatile("" : "+v"(mean_n));   // an opaque copy: CSE with the variance pass would keep C/2 differences live (spills)
    const float nmr = -mean_n * rstd;
    // normalised row -> B operands of fc1 (k order = accumulator order, see tail_pack_kernel); x1 stays in acc
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(s_g2 + 32 * i + 8 * q + 4 * h);
        const f32x4 be = *reinterpret_cast<const f32x4*>(s_b2n + 32 * i + 8 * q + 4 * h);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaf(fmaf(acc[i][4 * q + e], rstd, nmr), g[e], be[e]);      // two fma per value: (x rstd - mean rstd) g + b
        u32x4 w = __builtin_bit_cast(u32x4, bx[2 * i + (q >> 1)]);
        w[2 * (q & 1)] = E::pack2(y[0], y[1]);
        w[2 * (q & 1) + 1] = E::pack2(y[2], y[3]);
        bx[2 * i + (q >> 1)] = __builtin_bit_cast(V8, w);
        if (q & 1) __builtin_amdgcn_sched_barrier(0);
      }
  }
  __builtin_amdgcn_sched_barrier(0);     // nothing of the MLP prologue (bias / fragment loads) is hoisted into norm2
  KVQ_STAMP(2);

  // ---- MLP, software-pipelined over chunks of 32 hidden units ------------------------------------------------
  // fc1 accumulator of a chunk, initialised with its bias
  auto h_init = [&](f32x16& ha, int j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(s_fb1 + 32 * j + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) ha[4 * q + e] = b[e];
    }
  };
  // GELU of accumulator registers 2g, 2g+1 -> dword g of the packed B operand pair
  auto gelu_pair = [&](const f32x16& ha, u32x4 (&hb)[2], int g) {
    uint32_t w = gelu_pack2<E>(ha[2 * g], ha[2 * g + 1]);
    asm volatile("" : "+v"(w));      // pins the evaluation HERE (between two MFMAs): IR-level sinking would otherwise
    hb[g >> 2][g & 3] = w;           // move the whole GELU next to its first use, after the MFMA stream
  };
  f32x16 haA, haB;
  u32x4 hbA[2], hbB[2];
  // prologue item {W1_0, W1_1}: fc1 of chunks 0 and 1; GELU(chunk 0) runs under chunk 1's MFMAs
  st = next_item();
  h_init(haA, 0);
  h_init(haB, 1);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    haA = E::mfma32(*reinterpret_cast<const V8*>(st + s * 1024), bx[s], haA);
    if (s % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    haB = E::mfma32(*reinterpret_cast<const V8*>(st + PANEL + s * 1024), bx[s], haB);
    if (s % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int g = 0; g < 8; ++g) gelu_pair(haA, hbA, g);

  // one pipeline step: Y += W2_j . hb_cur ; ha_cur <- fc1(chunk j+2) ; hb_nxt <- GELU(ha_nxt)   (ha_nxt = chunk j+1)
  auto step = [&](int j, f32x16& ha_cur, u32x4 (&hb_cur)[2], f32x16& ha_nxt, u32x4 (&hb_nxt)[2], bool do_h,
                  bool do_g) __attribute__((always_inline)) {
    const unsigned char* sp = next_item();
    const V8 b0 = __builtin_bit_cast(V8, hb_cur[0]), b1 = __builtin_bit_cast(V8, hb_cur[1]);
    if (do_h) h_init(ha_cur, j + 2);
    // 2*KS MFMAs (the item is their 2*KS A fragments in order) and 8 GELU pairs, interleaved in program order: MFMA k,
    // then the k-th share of the VALU work; fragment k+1 is fetched before MFMA k is issued
    const int nk = do_h ? 2 * KS : KS;
    V8 a_nxt = *reinterpret_cast<const V8*>(sp);
#pragma unroll
    for (int k = 0; k < 2 * KS; ++k) {
      const V8 a = a_nxt;
      if (k + 1 < nk) a_nxt = *reinterpret_cast<const V8*>(sp + (k + 1) * 1024);
      if (k < KS) {                                   // fc2: tile k>>1, k-step k&1 of this chunk
        acc[k >> 1] = E::mfma32(a, (k & 1) ? b1 : b0, acc[k >> 1]);
      } else if (do_h) {                              // fc1 of chunk j+2
        ha_cur = E::mfma32(a, bx[k - KS], ha_cur);
      }
      if (do_g) {
#pragma unroll
        for (int g = (8 * k) / (2 * KS); g < (8 * (k + 1)) / (2 * KS); ++g) gelu_pair(ha_nxt, hb_nxt, g);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int j = 0;
  for (; j + 2 < NJ; j += 2) {
    step(j, haA, hbA, haB, hbB, true, true);
    step(j + 1, haB, hbB, haA, hbA, true, true);
  }
  step(j, haA, hbA, haB, hbB, false, true);          // chunk NJ-2: nothing left to start
  step(j + 1, haB, hbB, haA, hbA, false, false);     // chunk NJ-1
  KVQ_STAMP(3);

  // ---- + fc2 bias; write the residual stream back; optionally the next block's norm1 in ITS window order ------
#pragma unroll
  for (int i = 0; i < CM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 fb = *reinterpret_cast<const f32x4*>(s_fb2 + 32 * i + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][4 * q + e] += fb[e];
      if (q & 1) __builtin_amdgcn_sched_barrier(0);
    }
  // STAGED: the wave's tile in slots the ring has left — the last item sits in slot (NI - 1) % NST, every other slot is free (the barrier
  // of the last next_item() lies behind every wave's last read of the item before it); launch_tail checks that four tiles fit
  const int s_last = (NI - 1) % NST;
  unsigned char* stg2 = lds + ((NST - 1 - s_last) * SLOT + XTRA >= NW * TAIL_STG ? (s_last + 1) * SLOT : 0) + wave * TAIL_STG;
  if (STAGED) {
    int rowx[4];                   // residual-stream row of the tile rows 8 g + rrow, < 0: not stored (re-derived: nothing of the prologue stays live)
#pragma unroll
    for (int g = 0; g < 4; ++g) rowx[g] = __shfl(live ? (int)orig : -1, 8 * g + rrow);
#pragma unroll
    for (int i = 0; i < CM; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(stg_at(stg2, tj, 2 * q + h)) = (f32x4){acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(stg_at(stg2, 8 * g + rrow, rp));
        if (rowx[g] >= 0) *reinterpret_cast<f32x4*>(p.x + (size_t)rowx[g] * C + 32 * i + 4 * rp) = t;
      }
    }
  } else if (live && p.x16) {
    uint16_t* xr = reinterpret_cast<uint16_t*>(p.x) + (size_t)orig * C + 4 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<u32x2*>(xr + 32 * i + 8 * q) =
            (u32x2){Fp16::pack2(acc[i][4 * q], acc[i][4 * q + 1]), Fp16::pack2(acc[i][4 * q + 2], acc[i][4 * q + 3])};
  } else if (live) {
    float* xr = p.x + (size_t)orig * C + 4 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(xr + 32 * i + 8 * q) =
            (f32x4){acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
  }
  if (EMIT) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 4) s += (acc[i][r] + acc[i][r + 1]) + (acc[i][r + 2] + acc[i][r + 3]);
    s += __shfl_xor(s, 32);
    const float mu = s / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[i][r] - mu;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 32);
    const float rs = rsqrtf(sq / (float)C + p.eps);
    float mu_n = mu;
    asm volatile("" : "+v"(mu_n));
    const float nmr2 = -mu_n * rs;
    // 16 bytes per lane: the lane pair (h = 0 | 1) of a token exchanges the 8-byte pieces of (q, q + 1) by v_permlane32_swap, lane h
    // then owns channels 8 (2 t + h) .. + 7 of a tile — half the row-divergent store instructions (one row per cycle in the addresser)
    const long drow = (long)tb * p.next_rows + p.next_dst[tloc];
    if (QKV) {
      // ---- the next block's q | k | v (swin_backbone.py:252-260 of block b + 1): its norm1 row becomes the B operand in registers (k order
      // = accumulator order, as norm2's), then one 32-channel panel = one head of q, k or v at a time from 3C / 32 more ring panels ----
#pragma unroll
      for (int i = 0; i < CM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(s_nn + 32 * i + 8 * q + 4 * h);
          const f32x4 be = *reinterpret_cast<const f32x4*>(s_nn + C + 32 * i + 8 * q + 4 * h);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = fmaf(fmaf(acc[i][4 * q + e], rs, nmr2), g[e], be[e]);
          u32x4 w = __builtin_bit_cast(u32x4, bx[2 * i + (q >> 1)]);
          w[2 * (q & 1)] = E::pack2(y[0], y[1]);
          w[2 * (q & 1) + 1] = E::pack2(y[2], y[3]);
          bx[2 * i + (q >> 1)] = __builtin_bit_cast(V8, w);
          if (q & 1) __builtin_amdgcn_sched_barrier(0);
        }
      __builtin_amdgcn_sched_barrier(0);
      const int nH = C / 32;
#pragma unroll 1
      for (int j = 0; j < 3 * C / 32; j += 2) {
        const unsigned char* sp = next_item();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pj = j + u, which = pj / nH, head = pj - which * nH;
          f32x16 ha;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.qkv_b + 32 * pj + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) ha[4 * q + e] = b[e];
          }
#pragma unroll
          for (int s2 = 0; s2 < KS; ++s2) {
            ha = E::mfma32(*reinterpret_cast<const V8*>(sp + u * PANEL + s2 * 1024), bx[s2], ha);
            if (s2 % 4 == 3) __builtin_amdgcn_sched_barrier(0);
          }
          const float sc = which == 0 ? p.q_scale : 1.f;
          uint16_t* o = p.qkv_out + ((size_t)(which * nH + head) * p.qkv_rows + (size_t)drow) * 32;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            uint32_t pk[2][2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
              const int q = 2 * t + v;
              pk[v][0] = E::pack2(ha[4 * q] * sc, ha[4 * q + 1] * sc);
              pk[v][1] = E::pack2(ha[4 * q + 2] * sc, ha[4 * q + 3] * sc);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            if (live) *reinterpret_cast<u32x4*>(o + 8 * (2 * t + h)) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
          }
        }
      }
    }
    uint16_t* o = p.next_ln + (size_t)drow * C;
    int rowd[4];
    if (STAGED) {
#pragma unroll
      for (int g = 0; g < 4; ++g) rowd[g] = __shfl(live ? (int)drow : -1, 8 * g + rrow);
    }
#pragma unroll
    for (int i = 0; i < (QKV ? 0 : CM); ++i) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t pk[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = 2 * t + u;
          const f32x4 g = *reinterpret_cast<const f32x4*>(s_nn + 32 * i + 8 * q + 4 * h);
          const f32x4 be = *reinterpret_cast<const f32x4*>(s_nn + C + 32 * i + 8 * q + 4 * h);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = fmaf(fmaf(acc[i][4 * q + e], rs, nmr2), g[e], be[e]);
          pk[u][0] = E::pack2(y[0], y[1]);
          pk[u][1] = E::pack2(y[2], y[3]);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        const u32x4 piece = {s0[0], s1[0], s0[1], s1[1]};
        if (STAGED) *reinterpret_cast<u32x4*>(stg_at(stg2, tj, 4 * (i & 1) + 2 * t + h)) = piece;      // 128-byte row tile = channel tiles (i, i + 1)
        else if (live) *reinterpret_cast<u32x4*>(o + 32 * i + 8 * (2 * t + h)) = piece;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (STAGED && ((i & 1) || i == CM - 1)) {          // a row tile is complete: 8 rows x 128 (or a last 64) bytes per wave-store
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(stg_at(stg2, 8 * g + rrow, rp));
          if (rowd[g] >= 0 && ((i & 1) || rp < 4)) *reinterpret_cast<u32x4*>(p.next_ln + (size_t)rowd[g] * C + 64 * (i >> 1) + 8 * rp) = v;
        }
      }
    }
  }
#ifdef KVQ_TAIL_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  if (tr) {
    p.trace[blockIdx.x * 8 + 5] = wait_dma;
    p.trace[blockIdx.x * 8 + 6] = wait_bar;
  }
#endif
  KVQ_STAMP(4);
}

template <typename E, int CM, int NW>
static int launch_tail(const TailParams& p, hipStream_t st) {
  constexpr int C = 32 * CM, NST = tail_ring(C, NW), SLOT = tail_slot_bytes(C), XTRA = tail_stage_extra(C, NW);
  // KVQ_TAIL_STAGED=1: the x / attention / norm1 rows travel as 128-byte row segments through an LDS transpose (round 5, the review's
  // "coalesced row traffic"; bit-identical results).  Built, tested and measured — and OFF by default: the C = 96 launches take 97.6 /
  // 88.5 us against 93.1 / 87.5, the C = 192 ones 86.3 / 77.6 against 87.4 / 80.3, the 4-lane C2 line 355.7 / 358.6 against 360.0 /
  // 360.7 videos/s (profiles/r05_tail_staged_ab.txt): a quarter of the row-divergent global accesses per wave buys nothing, so the
  // lane-per-row accesses are NOT what these launches wait for (round 4's reading of the merge launch does not carry over).
  static const bool staged_on = getenv("KVQ_TAIL_STAGED") && atoi(getenv("KVQ_TAIL_STAGED")) == 1;
  // the epilogue's four tiles need NW * 4 KB of ring slots (+ the extra bytes) that do not hold the last item
  const int NI = tail_proj_items(C) + 1 + p.hidden / 32, s_last = (NI - 1) % NST;
  const bool staged = staged_on && !p.x16 && ((NST - 1 - s_last) * SLOT + XTRA >= NW * TAIL_STG || s_last * SLOT >= NW * TAIL_STG);
  const size_t lds = (size_t)NST * SLOT + (staged ? XTRA : 0) + ((((size_t)(4 * C + p.hidden) * 4) + 1023) & ~(size_t)1023) + (size_t)2 * C * 4;
  KVQ_REQUIRE(lds <= (size_t)163840 / tail_bpc(C, NW), KVQ_ERR_UNSUPPORTED, "kvq_block_tail: %zu B of LDS", lds);
  dim3 grid((unsigned)ceil_div(p.gather ? p.n_tok : p.M, 32 * NW)), block(64 * NW);
  auto go = [&](auto k) -> int {
    KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, block, lds, st, p);
    return KVQ_OK;
  };
  int rc;
  if (p.qkv_out) {
    KVQ_REQUIRE(!staged && (3 * C / 32) % 2 == 0, KVQ_ERR_UNSUPPORTED, "kvq_block_tail: q | k | v emission with staged rows / an odd panel count");
    rc = go(block_tail_kernel<E, CM, NW, 2, false>);
  } else if (p.next_ln) rc = staged ? go(block_tail_kernel<E, CM, NW, 1, true>) : go(block_tail_kernel<E, CM, NW, 1, false>);
  else rc = staged ? go(block_tail_kernel<E, CM, NW, 0, true>) : go(block_tail_kernel<E, CM, NW, 0, false>);
  if (rc) return rc;
  KVQ_CHECK_LAUNCH("block_tail_kernel");
  return KVQ_OK;
}

template <typename E>
static int launch_tail_e(const TailParams& p, int C, hipStream_t st) {
  switch (C) {      // 4 waves per workgroup (8 / 12 measured in round 1: no gain)
    case 96: return launch_tail<E, 3, 4>(p, st);
    case 128: return launch_tail<E, 4, 4>(p, st);
    case 192: return launch_tail<E, 6, 4>(p, st);
    default: break;
  }
  KVQ_REQUIRE(false, KVQ_ERR_UNSUPPORTED, "kvq_block_tail: C=%d not in {96,128,192}", C);
}

}  // namespace kvq

// C = 256 / 384 / 512: csrc/tailmm.hip (feature-sliced 32x32x16 GEMM chain, weights through a register ring);
// C = 96 / 128 / 192: the token-per-lane launch of this file
static bool use_tailmm(int C, int hidden) { return kvq::tailmm_supported(C, hidden); }

extern "C" int kvq_block_tail_supported(int C, int hidden) {
  if (use_tailmm(C, hidden)) return 1;
  // hidden/32 even and >= 4: the MLP pipeline rotates two accumulators
  return (C == 96 || C == 128 || C == 192) && hidden % 64 == 0 && hidden >= 128 ? 1 : 0;
}

extern "C" size_t kvq_block_tail_pack_bytes(int C, int hidden) {
  if (!kvq_block_tail_supported(C, hidden)) return 0;
  if (use_tailmm(C, hidden)) return kvq::tailmm_pack_bytes(C, hidden);
  return kvq::tail_items(C, hidden) * kvq::tail_slot_bytes(C) + kvq::tail_param_bytes(C, hidden);
}

extern "C" int kvq_block_tail_pack(const void* proj_w, const float* proj_b, const float* norm2_w, const float* norm2_b,
                                   const void* fc1_w, const float* fc1_b, const void* fc2_w, const float* fc2_b, int C,
                                   int hidden, void* pack, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(proj_w && proj_b && norm2_w && norm2_b && fc1_w && fc1_b && fc2_w && fc2_b && pack, KVQ_ERR_NULL,
              "kvq_block_tail_pack: NULL pointer");
  KVQ_REQUIRE(kvq_block_tail_supported(C, hidden), KVQ_ERR_UNSUPPORTED, "kvq_block_tail_pack: C=%d hidden=%d", C, hidden);
  if (use_tailmm(C, hidden))
    return tailmm_pack((const uint16_t*)proj_w, (const uint16_t*)fc1_w, (const uint16_t*)fc2_w, proj_b, norm2_w, norm2_b, fc1_b,
                       fc2_b, C, hidden, (unsigned char*)pack, (hipStream_t)stream);
  const long n_chunks = (long)tail_items(C, hidden) * tail_slot_bytes(C) / 16;
  const long n_par = (long)tail_param_bytes(C, hidden) / 4;
  const long total = n_chunks + n_par;
  hipLaunchKernelGGL(tail_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)proj_w, (const uint16_t*)fc1_w, (const uint16_t*)fc2_w, proj_b, norm2_w, norm2_b,
                     fc1_b, fc2_b, C, hidden, (unsigned char*)pack, n_chunks, n_par);
  KVQ_CHECK_LAUNCH("tail_pack_kernel");
  return KVQ_OK;
}

namespace kvq {
// qkv weight as 3C / 32 panels of 32 output channels in the W1 panel layout (k = channel in accumulator order), two panels per ring item
__global__ void tail_qkv_pack_kernel(const uint16_t* wq, int C, unsigned char* out, long n_chunks) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_chunks) return;
  const int KS = C / 16;
  const int panel = (int)(g / (KS * 64)), rem = (int)(g % (KS * 64));
  const int f = rem >> 6, lane = rem & 63, m = lane & 31, h = lane >> 5;
  uint16_t* o = reinterpret_cast<uint16_t*>(out + g * 16);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = 32 * (f >> 1) + 8 * (2 * (f & 1) + (e >> 2)) + 4 * h + (e & 3);
    o[e] = wq[(size_t)(32 * panel + m) * C + ch];
  }
}
}  // namespace kvq

extern "C" size_t kvq_block_tail_qkv_pack_bytes(int C, int hidden) {
  if (use_tailmm(C, hidden)) return kvq::tailmm_qkv_pack_bytes(C, hidden);
  if (!kvq_block_tail_supported(C, hidden) || (3 * C / 32) % 2) return 0;
  return (size_t)(3 * C / 32) * 64 * C;           // 3C / 32 panels of 64 C bytes
}

extern "C" int kvq_block_tail_qkv_pack(const void* qkv_w, int C, int hidden, void* pack, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(qkv_w && pack, KVQ_ERR_NULL, "kvq_block_tail_qkv_pack: NULL pointer");
  KVQ_REQUIRE(kvq_block_tail_qkv_pack_bytes(C, hidden) > 0, KVQ_ERR_UNSUPPORTED, "kvq_block_tail_qkv_pack: C=%d hidden=%d", C, hidden);
  if (use_tailmm(C, hidden)) return tailmm_qkv_pack((const uint16_t*)qkv_w, C, hidden, (unsigned char*)pack, (hipStream_t)stream);
  const long n_chunks = (long)kvq_block_tail_qkv_pack_bytes(C, hidden) / 16;
  hipLaunchKernelGGL(tail_qkv_pack_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv_w, C,
                     (unsigned char*)pack, n_chunks);
  KVQ_CHECK_LAUNCH("tail_qkv_pack_kernel");
  return KVQ_OK;
}

extern "C" int kvq_block_tail(const KvqBlockTailArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->attn && a->x && a->pack, KVQ_ERR_NULL, "kvq_block_tail: NULL pointer");
  KVQ_REQUIRE(kvq_block_tail_supported(a->C, a->hidden), KVQ_ERR_UNSUPPORTED, "kvq_block_tail: C=%d hidden=%d", a->C,
              a->hidden);
  KVQ_REQUIRE(a->M > 0 && a->out_rows > 0 && (!a->scatter_map || a->map_rows > 0), KVQ_ERR_SHAPE, "kvq_block_tail: bad rows");
  KVQ_REQUIRE(!a->next_ln || (a->next_norm_w && a->next_norm_b && a->next_dst && a->next_rows > 0), KVQ_ERR_NULL,
              "kvq_block_tail: next_ln without its norm / map");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_block_tail: dtype %d", a->dtype);
  TailParams p{};
  p.attn = (const uint16_t*)a->attn; p.x = a->x; p.x16 = a->x_f16; p.map = a->scatter_map; p.map_rows = a->map_rows; p.out_rows = a->out_rows;
  p.M = a->M; p.hidden = a->hidden; p.pack = (const unsigned char*)a->pack;
  p.nn_w = a->next_norm_w; p.nn_b = a->next_norm_b; p.next_dst = a->next_dst; p.next_ln = (uint16_t*)a->next_ln;
  p.next_rows = a->next_rows; p.eps = a->eps; p.trace = g_trace; p.trace_blocks = g_trace_blocks;
  if (a->qkv_out) {
    KVQ_REQUIRE(kvq_block_tail_qkv_pack_bytes(a->C, a->hidden) > 0, KVQ_ERR_UNSUPPORTED, "kvq_block_tail: C=%d cannot emit q | k | v", a->C);
    KVQ_REQUIRE(!a->next_ln && a->next_qkv_pack && a->next_qkv_b && a->next_norm_w && a->next_norm_b && a->next_dst && a->next_rows > 0 &&
                    a->num_heads * 32 == a->C, KVQ_ERR_NULL, "kvq_block_tail: qkv_out needs next_qkv_pack / next_qkv_b / next norm / map, heads of 32, and no next_ln");
    const long nb = a->attn_gather ? a->M / a->map_rows : (a->scatter_map ? a->M / a->map_rows : a->M / a->out_rows);
    p.qkv_pack = (const unsigned char*)a->next_qkv_pack; p.qkv_b = a->next_qkv_b; p.qkv_out = (uint16_t*)a->qkv_out;
    p.q_scale = a->q_scale; p.num_heads = a->num_heads; p.qkv_rows = nb * a->next_rows;
  }
  if (a->attn_gather) {
    KVQ_REQUIRE(a->map_rows > 0 && a->M % a->map_rows == 0, KVQ_ERR_SHAPE, "kvq_block_tail: attn_gather needs M = n_batch * map_rows");
    p.gather = a->attn_gather; p.n_tok = a->M / a->map_rows * a->out_rows; p.map = nullptr;
  }
  if (use_tailmm(a->C, a->hidden)) return tailmm_launch(p, a->C, a->dtype, (hipStream_t)stream);
  return a->dtype == KVQ_DT_FP16 ? launch_tail_e<Fp16>(p, a->C, (hipStream_t)stream)
                                 : launch_tail_e<Bf16>(p, a->C, (hipStream_t)stream);
}
