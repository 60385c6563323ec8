// 16-bit-operand MFMA GEMM with fused epilogues for the Swin-3D trunk (gfx950).
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]      A [M][K], W [N][K] (nn.Linear layout) bf16|fp16, fp32 acc
//
// Replaces the reference's nn.Linear calls on the hot path (swin_backbone.py:254 qkv, :323 proj,
// :84-87 fc1/fc2, :553 reduction, :726 patch-embed conv as an im2col GEMM) together with what
// surrounds them: bias, exact GELU, the q scale + head split (:255-260), window_reverse + inverse
// roll + crop + residual add (:472-488, :509, :514).
//
// Structure: 256 threads = 4 waves in a 2x2 grid; block tile (64*MI) x (64*NI), wave tile
// (32*MI) x (32*NI) built from v_mfma_f32_32x32x16_{bf16,f16}; K streamed through a 4-slice LDS ring
// by LDS-DMA with counted waits (see the kernel comment).
#include <stdlib.h>

#include <algorithm>

#include "gemm_common.hpp"

namespace kvq {

// what out-of-image / K-padding chunks of an implicit-GEMM A tile are fetched from
__device__ __attribute__((aligned(16))) const uint32_t kvq_zero_chunk[4] = {0u, 0u, 0u, 0u};

#ifndef KVQ_GEMM_NST
#define KVQ_GEMM_NST 3      // slices in the LDS ring: 48 KB per 128x128 block -> 3 blocks/CU (measured: 4 -> 3.66 ms,
                            // 3 -> 3.53 ms, 2 -> 3.50 ms per step: co-residency beats DMA depth)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

template <typename V>
__device__ __forceinline__ void lds_read_b128(V& dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void gemm_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K loop: an NST-deep LDS ring of BK=32 slices filled by LDS-DMA (global_load_lds_dwordx4: 16 B per lane,
// no VGPR round trip, no ds_write), NST-1 slices in flight while one is multiplied.  The waits are COUNTED
// (s_waitcnt vmcnt((NST-2)*NL): only the slice about to be read must have landed) and the barrier is the raw
// s_barrier, so the younger slices' DMAs stay in flight across it; one barrier per slice covers both the
// RAW (slice t landed for every wave) and the WAR (everybody finished slice t-1 before its buffer is
// refilled with slice t+NST-1).  LDS rows are 64 B (no padding: the DMA image is lane-linear); the 16-B chunk
// c of row r lives at chunk c ^ ((r>>2)&3) — applied on the SOURCE address of the DMA and on the
// fragment read — which makes the 16-lane service groups of ds_read_b128 hit 16 distinct 16-B slots.
// BK = 32: LDS rows of 64 B, 16-B chunk c of row r at chunk c ^ ((r>>2)&3).  BK = 64 (the "one big tile per CU"
// variants for long-K shapes): rows of 128 B, chunk c at c ^ ((r>>1)&7) — in both, the 16 lanes of a ds_read_b128
// service group land on 16 distinct 16-B slots.
// (Measured in round 2 and dropped: eight waves in a 2x4 grid over the same tile, rings of 6 / 8 slices, fragment reads of slice t+1
// under the MFMAs of slice t — all within -8..0 % on the lone-workgroup shapes; what those shapes needed was the wider tile of
// gemm256.hip.)
template <typename E, int MI, int NI, int BK, int EPI, int NST = KVQ_GEMM_NST, bool IMPL = false, bool WALK = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int WN = 2;                                                  // 2 x 2 wave grid
  static_assert(IMPL || !WALK, "WALK is a mode of the implicit-GEMM instantiation");
  fp16_saturate_mode();
  static_assert(BK == 32 || BK == 64, "ring slices are 32 or 64 deep");
  static_assert(NST >= 2 && NST <= 8, "ring of 2 .. 8 slices");
  constexpr int NT = 128 * WN;                                         // threads
  constexpr int BM = 64 * MI, BN = 32 * NI * WN;
  constexpr int RB = BK * 2, CH = RB / 16, KK = BK / 16;             // row bytes, 16-B chunks per row, MFMA k-steps per slice
  constexpr int A_BYTES = BM * RB, ST_BYTES = (BM + BN) * RB;
  constexpr int A_PER = BM * CH / NT, B_PER = BN * CH / NT, NL = A_PER + B_PER;   // DMAs per thread per slice
  static_assert((BM * CH) % NT == 0 && (BN * CH) % NT == 0, "whole DMA rounds");
  auto swz = [](int row) { return BK == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E::v8;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8, each XCD has its own L2.  Give
  // every XCD a CONTIGUOUS range of the logical tile index (N fastest), so the N-tiles that re-read one A
  // row-panel share an L2 instead of fetching it once per XCD (bijective also when nwg % 8 != 0).
  const int nbn = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, qd = nwg >> 3, rm = nwg & 7;
  const int lid0 = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
  const int ntile = nwg / p.ksplit, ksl = lid0 / ntile, lid = lid0 - ksl * ntile;     // K range index (slowest), tile
  const int bm = lid / nbn, bn = lid % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // DMA assignment: 16-B chunk q = i*256 + tid of the slice image: row q>>2, physical chunk q&3
  const uint16_t* a_src[A_PER];
  const uint16_t* b_src[B_PER];
  int a_t0[IMPL ? A_PER : 1], a_y0[IMPL ? A_PER : 1], a_x0[IMPL ? A_PER : 1], a_c[IMPL ? A_PER : 1];   // IMPL: pixel origin, chunk column
  int4 a_tap[IMPL ? A_PER : 1];                                                                          // IMPL: tap of the next slice to issue
  // IMPL without a table: a table entry is a per-lane VMEM load in the same in-order queue as the slice DMAs, and the s_waitcnt
  // vmcnt(0) in front of its use drains both slices in flight at every step (1x1x1 over 3072 channels: 60 us as a table-driven
  // conv against 40 us as the plain GEMM it is).  Uniform walk state of the next slice to issue:
  // (a compile-time mode: with both in one kernel the table path's pending tap registers make the walk path wait as well)
  constexpr bool walk = WALK;
  int u_dd = 0, u_dh = 0, u_dw = 0, u_c8 = 0;
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int q = i * NT + tid, row = q / CH, c = (q % CH) ^ swz(row);
    if (IMPL) {
      int m = min(m0 + row, p.M - 1);
      const int wo = m % p.cWo; m /= p.cWo;
      const int ho = m % p.cHo; m /= p.cHo;
      const int dq = m % p.cDo, b = m / p.cDo;
      a_t0[i] = dq * p.csd - p.cpd; a_y0[i] = ho * p.csh - p.cph; a_x0[i] = wo * p.csw - p.cpw; a_c[i] = c;
      // element offset of (b, t0, y0, x0, 0): may point before the image; only ever dereferenced with an in-bounds tap added
      a_src[i] = p.A + ((((long)b * p.cD + a_t0[i]) * p.cH + a_y0[i]) * p.cW + a_x0[i]) * (long)p.cC;
      if (walk) {
        a_src[i] += 8 * c;
        a_tap[i] = (int4){0, 0, 0, 0};
      } else {
        a_tap[i] = p.taps[(ksl * (p.K / BK) / p.ksplit) * CH + c];
      }
    } else {
      size_t ar = (size_t)min(m0 + row, p.M - 1);
      if (p.a_gather) {
        const int bq = (int)(ar / p.a_rows);
        ar = (size_t)bq * p.a_phys_rows + p.a_gather[ar - (size_t)bq * p.a_rows];
      }
      a_src[i] = p.A + ar * p.K + c * 8;
    }
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int q = i * NT + tid, row = q / CH, c = (q % CH) ^ swz(row);
    b_src[i] = p.W + (size_t)min(n0 + row, p.N - 1) * p.K + c * 8;
  }
  const int kt0 = ksl * (p.K / BK) / p.ksplit, nk = (ksl + 1) * (p.K / BK) / p.ksplit - kt0;      // this workgroup's k-slices
  // walk: per-lane source of the next slice and its advance per slice (0 for rows whose tap falls outside the image: they keep
  // reading the zero chunk).  Both only change when the TAP changes (every C / 32 slices): with one workgroup per CU — the late
  // layers: 100-196 tiles — nothing hides the issue path, and per-slice compares / selects / 64-bit address sums cost what they cost
  const uint16_t* w_cur[IMPL ? A_PER : 1];
  int w_step[IMPL ? A_PER : 1];
  auto retap = [&]() {
    const int toff = ((u_dd * p.cH + u_dh) * p.cW + u_dw) * p.cC + 8 * u_c8;       // wave-uniform
    const bool tap_ok = u_dd < p.ckd;                                              // past the last tap: the zero padding of K
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const bool ok = tap_ok && (unsigned)(a_t0[i] + u_dd) < (unsigned)p.cD && (unsigned)(a_y0[i] + u_dh) < (unsigned)p.cH &&
                      (unsigned)(a_x0[i] + u_dw) < (unsigned)p.cW;
      w_cur[i] = ok ? a_src[i] + toff : reinterpret_cast<const uint16_t*>(kvq_zero_chunk);
      w_step[i] = ok ? BK : 0;
    }
  };
  if (walk) {
    const int C8 = p.cC >> 3;
    int tap = (kt0 * CH) / C8;
    u_c8 = kt0 * CH - tap * C8;
    u_dw = tap % p.ckw; tap /= p.ckw;
    u_dh = tap % p.ckh; u_dd = tap / p.ckh;
    retap();
  }
  auto issue = [&](int kl) {                      // kl: slice index inside the range; ring slot kl % NST
    const int kt = kt0 + kl;
    unsigned char* st = lds + (kl % NST) * ST_BYTES;
    if (IMPL) {
      // issue() is called for consecutive slices: the tap of THIS slice was fetched during the previous call (its L2
      // latency would otherwise sit in front of every DMA), the next slice's is requested now
      if (walk) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)w_cur[i], (lds_ptr_t)(st + (i * NT + wave * 64) * 16), 16, 0, 0);
          w_cur[i] += w_step[i];
        }
        u_c8 += CH;
        if (u_c8 >= (p.cC >> 3)) {
          u_c8 = 0;
          if (++u_dw == p.ckw) {
            u_dw = 0;
            if (++u_dh == p.ckh) { u_dh = 0; ++u_dd; }
          }
          retap();
        }
      } else {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          const int4 t = a_tap[i];
          const bool ok = t.w >= 0 && (unsigned)(a_t0[i] + t.x) < (unsigned)p.cD && (unsigned)(a_y0[i] + t.y) < (unsigned)p.cH &&
                          (unsigned)(a_x0[i] + t.z) < (unsigned)p.cW;
          const uint16_t* src = ok ? a_src[i] + t.w : reinterpret_cast<const uint16_t*>(kvq_zero_chunk);
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(st + (i * NT + wave * 64) * 16), 16, 0, 0);
          a_tap[i] = p.taps[min(kt + 1, p.K / BK - 1) * CH + a_c[i]];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + kt * BK), (lds_ptr_t)(st + (i * NT + wave * 64) * 16), 16,
                                         0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b_src[i] + kt * BK),
                                       (lds_ptr_t)(st + A_BYTES + (i * NT + wave * 64) * 16), 16, 0, 0);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;


  const bool tr = p.trace && tid == 0 && (int)blockIdx.x < p.trace_blocks;
  (void)bn;
  if (tr) p.trace[blockIdx.x * 8 + 0] = __builtin_readcyclecounter();
#pragma unroll
  for (int kl = 0; kl < NST - 1; ++kl)
    if (kl < nk) issue(kl);
  const int frow = lane & 31, fkg = lane >> 5;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)lds;          // LDS byte address of the ring
  // per-lane fragment byte offsets inside a slice (swizzled), one per kk
  int a_off[MI][KK], b_off[NI][KK];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * 32 * MI + i * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) a_off[i][kk] = row * RB + (((kk * 2 + fkg) ^ swz(row)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = wn * 32 * NI + j * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) b_off[j][kk] = A_BYTES + row * RB + (((kk * 2 + fkg) ^ swz(row)) << 4);
  }
  // vmcnt wait for "slice kt has landed": the slices issued behind it (at most `depth`) may still be in flight
  constexpr int PER = NL + ((IMPL && !walk) ? A_PER : 0);    // table-driven IMPL: every issue also carries A_PER tap loads
  auto wait_landed = [&](int kt, auto depth_tag) {
    constexpr int DEPTH = decltype(depth_tag)::value;
    const int younger = nk - 1 - kt;
    if (DEPTH >= 1 && younger >= DEPTH) gemm_wait_vmcnt<DEPTH * PER>();
    else if (DEPTH > 5 && younger == 5) gemm_wait_vmcnt<5 * PER>();
    else if (DEPTH > 4 && younger == 4) gemm_wait_vmcnt<4 * PER>();
    else if (DEPTH > 3 && younger == 3) gemm_wait_vmcnt<3 * PER>();
    else if (DEPTH > 2 && younger == 2) gemm_wait_vmcnt<2 * PER>();
    else if (DEPTH > 1 && younger == 1) gemm_wait_vmcnt<PER>();
    else gemm_wait_vmcnt<0>();
  };
  for (int kt = 0; kt < nk; ++kt) {
    // slice kt must have landed; up to two younger slices stay in flight
    // (up to NST - 2 younger slices stay in flight; table-driven IMPL: every issue also carries A_PER tap loads)
    wait_landed(kt, std::integral_constant<int, NST - 2>{});
    __builtin_amdgcn_s_barrier();
    if (tr && kt == 0) p.trace[blockIdx.x * 8 + 1] = __builtin_readcyclecounter();
    if (kt + NST - 1 < nk) issue(kt + NST - 1);
    const unsigned char* st = lds + (kt % NST) * ST_BYTES;
    // Fragments of k-step kk+1 are requested before the MFMAs of k-step kk (double-buffered registers).  The reads are
    // issued from inline asm with COUNTED waits: LDS returns in order, so "at most MI+NI outstanding" = k-step kk has
    // landed while k-step kk+1 is still in flight.  (Left to the compiler, every k-step waits lgkmcnt(0): with one or two
    // waves per SIMD — the big-tile variants — nothing else hides that round trip.)
    const unsigned sbase = lds_base + (kt % NST) * ST_BYTES;
    V8 af[2][MI], bfr[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) lds_read_b128(af[0][i], sbase + a_off[i][0]);
#pragma unroll
    for (int j = 0; j < NI; ++j) lds_read_b128(bfr[0][j], sbase + b_off[j][0]);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) {
#pragma unroll
        for (int i = 0; i < MI; ++i) lds_read_b128(af[(kk + 1) & 1][i], sbase + a_off[i][kk + 1]);
#pragma unroll
        for (int j = 0; j < NI; ++j) lds_read_b128(bfr[(kk + 1) & 1][j], sbase + b_off[j][kk + 1]);
      }
      // the wait is tied to the registers it releases, so no use can be scheduled above it
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (kk + 1 < KK) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[kk & 1][i]) : "n"(MI + NI) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[kk & 1][i])::"memory");
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(bfr[kk & 1][j]));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = E::mfma32(af[kk & 1][i], bfr[kk & 1][j], acc[i][j]);   // columns >= N multiply clamped rows: discarded below
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();   // everybody is done with the ring before the epilogue slabs overwrite it
  if (tr) p.trace[blockIdx.x * 8 + 2] = __builtin_readcyclecounter();

  // ---- epilogue (gemm_common.hpp): per-wave LDS transpose, float4 math, 16-byte stores
  GemmEpilogue<E, EPI, NI> ep;
  ep.init(p, reinterpret_cast<float*>(lds) + wave * GemmEpilogue<E, EPI, NI>::SLAB_FLOATS, n0 + wn * 32 * NI, lane);
#pragma unroll
  for (int i = 0; i < MI; ++i)
    ep.tile(p, m0 + wm * 32 * MI + i * 32, ksl, [&](int j, int r) { return acc[i][j][r]; });
  if (tr) {
    p.trace[blockIdx.x * 8 + 3] = __builtin_readcyclecounter();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    p.trace[blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hwid;
  }
}

template <typename E, int MI, int NI, int BK, int EPI, int NST = KVQ_GEMM_NST, bool IMPL = false, bool WALK = false>
static int launch_one(const GemmParams& p_in, hipStream_t st) {
  constexpr int WN = 2;
  GemmParams p = p_in;
  p.ksplit = p.ksplit < 1 ? 1 : p.ksplit;
  constexpr int BM = 64 * MI, BN = 32 * NI * WN;
  constexpr size_t main_bytes = NST * (BM + BN) * BK * 2;               // ring of 2*BK-byte rows
  constexpr size_t epi_bytes = 2 * WN * 32 * (32 * NI) * sizeof(float);     // one fp32 slab per wave
  constexpr size_t lds_bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  auto kern = gemm_kernel<E, MI, NI, BK, EPI, NST, IMPL, WALK>;
  static LdsOptIn opt;            // > 64 KiB of LDS needs the opt-in attribute (once per instantiation and device)
  if (lds_bytes > 64 * 1024)
    if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), (int)lds_bytes)) return rc;
  dim3 grid(ceil_div(p.M, BM) * ceil_div(p.N, BN) * p.ksplit), block(128 * WN);
  hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, p);
  KVQ_CHECK_LAUNCH("gemm_kernel");
  return KVQ_OK;
}

// ---- split-K, second launch: out = epilogue(sum_s partial[s]) — the S partials are added in index order (bit-reproducible),
// then bias / activation / identity / residual scatter exactly as the GEMM's own epilogue does.  A thread owns 8 columns of a row.
template <typename E>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, const float* partial, int S, int epi) {
  fp16_saturate_mode();
  const long chunks = (long)p.M * (p.N / 8);
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= chunks) return;
  const int m = (int)(idx / (p.N / 8)), n = (int)(idx % (p.N / 8)) * 8;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.f;
  for (int s2 = 0; s2 < S; ++s2) {
    const f32x4* src = reinterpret_cast<const f32x4*>(partial + ((size_t)s2 * p.M + m) * p.N + n);
    const f32x4 a = src[0], b = src[1];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] += a[k]; v[4 + k] += b[k]; }
  }
  if (p.bias) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.bias + n), b = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] += a[k]; v[4 + k] += b[k]; }
  }
  if (epi == KVQ_EPI_STORE_F32 || epi == KVQ_EPI_RESID_F32) {
    long orow = m;
    if (epi == KVQ_EPI_RESID_F32 && p.scatter_map) {
      const int b = m / p.map_rows, s3 = p.scatter_map[m - b * p.map_rows];
      if (s3 < 0) return;
      orow = (long)b * p.out_rows + s3;
    }
    f32x4* o = reinterpret_cast<f32x4*>(p.out_f32 + (size_t)orow * p.N + n);
    f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
    if (epi == KVQ_EPI_RESID_F32) { x0 += o[0]; x1 += o[1]; }
    o[0] = x0; o[1] = x1;
    return;
  }
  if (epi == KVQ_EPI_GELU_BF16) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = gelu_fast(v[k]);
  } else if (epi == KVQ_EPI_QGELU_BF16) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] / (1.f + __expf(-1.702f * v[k]));
  } else if (epi == KVQ_EPI_RELU_BF16) {
    if (p.resid_h) {
      const u32x4 rr = *reinterpret_cast<const u32x4*>(p.resid_h + (size_t)m * p.N + n);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += E::to_f32((uint16_t)(rr[k] & 0xffffu));
        v[2 * k + 1] += E::to_f32((uint16_t)(rr[k] >> 16));
      }
    }
    if (p.resid_f32) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.resid_f32 + (size_t)m * p.N + n);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.resid_f32 + (size_t)m * p.N + n + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += a[k]; v[4 + k] += b[k]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    if (p.out_f32) {
      f32x4* o = reinterpret_cast<f32x4*>(p.out_f32 + (size_t)m * p.N + n);
      o[0] = (f32x4){v[0], v[1], v[2], v[3]};
      o[1] = (f32x4){v[4], v[5], v[6], v[7]};
    }
  }
  *reinterpret_cast<u32x4*>(p.out_h + (size_t)m * (p.ldc ? p.ldc : p.N) + p.col_off + n) =
      (u32x4){E::pack2(v[0], v[1]), E::pack2(v[2], v[3]), E::pack2(v[4], v[5]), E::pack2(v[6], v[7])};
}

// S for a launch of `tiles` output tiles walking nk ring slices of bk each: few tiles on 256 CUs with a long K loop are split
// until about two workgroups per CU exist, keeping >= 768 of K per workgroup (below that the partial-tile traffic and the second
// launch cost more than the idle CUs: measured on the trunk's stage-2 merge, K = 1536 over 150 tiles, 25 -> 35 us split in two).
// The 64-deep variants hold a CU's whole LDS (one workgroup per CU): more than 256 workgroups would only add a second, half-empty
// round (fc2 of stage 3, K = 3072 over 150 tiles: 43.6 us un-split, 43.8 split in three) — they stay whole.
static int splitk_factor(long tiles, int nk, int bk) {
  // 196 tiles (res4 of SlowFast's slow pathway) stay whole since the implicit GEMM walks its taps: 1x3x3 / 256 channels 36 us whole,
  // 40 us split in two; 100 tiles (res5): 58 us whole, 38 us split in five
  if (bk != 32 || tiles > 160 || nk < 48) return 1;
  int S = (int)(512 / tiles);
  S = S > 8 ? 8 : S;
  while (S > 1 && nk / S < 24) --S;
  return S < 1 ? 1 : S;
}

// tile (MI, NI) for a shape, encoded (10*MI + NI)*100 + BK
int gemm_variant(int M, int N, int K) {
  // Measured per shape of the trunk (B = 4 clips, us: 128x128 / 128x64 / 64x64): fc2 stage 2 (K = 1536) 36.7 / 41.4 /
  // 44.3 although 128x128 makes only 294 workgroups; qkv stage 3 25.6 / 31.0 / 35.1 (450 workgroups); proj stage 2
  // (K = 384, epilogue-dominated) 25.0 / 21.3 / 20.2; merge stage 0 (N = 192 = 1.5 tiles of 128) 28.0 / 24.7 / 28.1.
  // => the big tile whenever the K loop is long or it still fills the chip; 64-wide columns when the last
  // 128-column tile would be at most half used.
  // Long K with few tiles: ONE big tile per CU, 64-deep slices (what the vendor library picks for these shapes:
  // 128x160x64 ... 128x256x64 tiles, 225-294 of them).  First candidate that fits the chip in a single round.
  // Few rows (KSVQE's CLIP tower: 200-800 token rows; one clip's stage 3: 784): 128x128 tiles leave most of the chip idle
  // (12-48 workgroups walking a K of 3072) — 64x64 tiles quadruple the workgroups; operand re-reads are irrelevant at this size.
  if ((long)ceil_div(M, 128) * ceil_div(N, 128) < 96) return 11 * 100 + 32;
  if (K % 64 == 0 && K >= 1024) {      // measured: fc2 stage 3 45 -> 40 us, merges -2 us; K = 768 shapes lose
    static const int cand[3][2] = {{2, 2}, {3, 2}, {2, 4}};
    for (auto& c : cand) {
      const long tiles = (long)ceil_div(M, 64 * c[0]) * ceil_div(N, 64 * c[1]);
      if (tiles <= 256) return (c[0] * 10 + c[1]) * 100 + 64;
    }
  }
  const long blocks128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
  const bool big = blocks128 >= 400 || K >= 1024 || (blocks128 >= 256 && K >= 768);
  if (!big) return 11 * 100 + 32;
  const int rem = N % 128;
  return ((rem != 0 && rem <= 64) ? 21 : 22) * 100 + 32;
}

// the tile grid and slice count of the variant launch_gemm / launch_conv would pick
static void variant_tiles(int M, int N, int K, bool conv, long* tiles, int* nk, int* bk_out) {
  int var = gemm_variant(M, N, K);
  int bk = var % 100, mi = var / 1000, ni = (var / 100) % 10;
  if (conv) {
    const int v = var / 100;
    if (v != 22 && v != 21 && v != 12 && v != 11) { mi = 2; ni = 2; }
    bk = 32;
  }
  *tiles = (long)ceil_div(M, 64 * mi) * ceil_div(N, 64 * ni);
  *nk = K / bk;
  *bk_out = bk;
}

template <typename E, int EPI>
static int finish_splitk(const GemmParams& p, const float* partial, int S, hipStream_t st) {
  const long chunks = (long)p.M * (p.N / 8);
  hipLaunchKernelGGL(splitk_reduce_kernel<E>, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, p, partial, S, (int)EPI);
  KVQ_CHECK_LAUNCH("splitk_reduce_kernel");
  return KVQ_OK;
}

template <typename E, int EPI>
static int launch_gemm_variant(const GemmParams& p, hipStream_t st) {
  if (gemm8p_wanted(p.M, p.N, p.K)) return launch_gemm8p<E, EPI>(p, st);
  const int var = gemm_variant(p.M, p.N, p.K);
  // 128x128 with 64-deep slices and a 2-slice ring (64 KB, 2 workgroups per CU; whole 128-B lines per DMA row):
  // measured per epilogue on the trunk's shapes — qkv stage 2: 28 -> 26 us, stage 3: 25 -> 22; fc1 equal; fc2 +2 us
  if (EPI == KVQ_EPI_QKV_BF16 && var == 2232 && p.K % 64 == 0) return launch_one<E, 2, 2, 64, EPI, 2>(p, st);
  if (var % 100 == 64) {
    switch (var / 100) {
      case 22: return launch_one<E, 2, 2, 64, EPI>(p, st);
      case 32: return launch_one<E, 3, 2, 64, EPI>(p, st);
      default: return launch_one<E, 2, 4, 64, EPI>(p, st);
    }
  }
  switch (var / 100) {
    case 22: return launch_one<E, 2, 2, 32, EPI>(p, st);
    case 21: return launch_one<E, 2, 1, 32, EPI>(p, st);
    case 12: return launch_one<E, 1, 2, 32, EPI>(p, st);
    default: return launch_one<E, 1, 1, 32, EPI>(p, st);
  }
}

// split-K around a variant launcher: partial tiles (STORE_F32 instantiation of the same variant) + the reduce / epilogue launch
template <typename E, int EPI, bool CONV>
static int launch_maybe_split(const GemmParams& p, hipStream_t st);

template <typename E, int EPI>
static int launch_gemm(const GemmParams& p, hipStream_t st) { return launch_maybe_split<E, EPI, false>(p, st); }

// implicit-GEMM convolution: the 32-deep variants only (a slice = 4 chunks = at most 4 taps)
template <typename E, int EPI>
static int launch_conv_variant(const GemmParams& p, hipStream_t st) {
  int var = gemm_variant(p.M, p.N, p.K) / 100;
  if (var != 22 && var != 21 && var != 12 && var != 11) var = 22;
  if (!p.taps) {                // C % 32 == 0, full tap set: the kernel walks the taps with wave-uniform counters
    switch (var) {
      case 22: return launch_one<E, 2, 2, 32, EPI, KVQ_GEMM_NST, true, true>(p, st);
      case 21: return launch_one<E, 2, 1, 32, EPI, KVQ_GEMM_NST, true, true>(p, st);
      case 12: return launch_one<E, 1, 2, 32, EPI, KVQ_GEMM_NST, true, true>(p, st);
      default: return launch_one<E, 1, 1, 32, EPI, KVQ_GEMM_NST, true, true>(p, st);
    }
  }
  switch (var) {
    case 22: return launch_one<E, 2, 2, 32, EPI, KVQ_GEMM_NST, true>(p, st);
    case 21: return launch_one<E, 2, 1, 32, EPI, KVQ_GEMM_NST, true>(p, st);
    case 12: return launch_one<E, 1, 2, 32, EPI, KVQ_GEMM_NST, true>(p, st);
    default: return launch_one<E, 1, 1, 32, EPI, KVQ_GEMM_NST, true>(p, st);
  }
}

template <typename E, int EPI>
static int launch_conv(const GemmParams& p, hipStream_t st) { return launch_maybe_split<E, EPI, true>(p, st); }

template <typename E, int EPI, bool CONV>
static int launch_maybe_split(const GemmParams& p, hipStream_t st) {
  long tiles;
  int nk, bk;
  variant_tiles(p.M, p.N, p.K, CONV, &tiles, &nk, &bk);
  static const bool splitk_on = !(getenv("KVQ_SPLITK") && atoi(getenv("KVQ_SPLITK")) == 0);      // A/B knob (round 5): 0 = never split K
  const int S = (splitk_on && EPI != KVQ_EPI_QKV_BF16 && p.sk_ws) ? splitk_factor(tiles, nk, bk) : 1;
  if (S > 1 && (size_t)S * p.M * p.N * sizeof(float) <= p.sk_bytes) {
    GemmParams q = p;
    q.bias = nullptr; q.out_f32 = p.sk_ws; q.out_h = nullptr; q.scatter_map = nullptr; q.resid_h = nullptr; q.resid_f32 = nullptr;
    q.ksplit = S;
    const int rc = CONV ? launch_conv_variant<E, KVQ_EPI_STORE_F32>(q, st) : launch_gemm_variant<E, KVQ_EPI_STORE_F32>(q, st);
    if (rc) return rc;
    return finish_splitk<E, EPI>(p, p.sk_ws, S, st);
  }
  return CONV ? launch_conv_variant<E, EPI>(p, st) : launch_gemm_variant<E, EPI>(p, st);
}

template <int EPI>
static int launch_dt(int dtype, const GemmParams& p, hipStream_t st) {
  return dtype == KVQ_DT_FP16 ? launch_gemm<Fp16, EPI>(p, st) : launch_gemm<Bf16, EPI>(p, st);
}

}  // namespace kvq

namespace kvq {
unsigned long long* g_trace = nullptr;     // shared with attn.hip (declared in common.hpp)
int g_trace_blocks = 0;
}
using kvq::g_trace;
using kvq::g_trace_blocks;

extern "C" int kvq_debug_gemm_trace(void* dev_buf, int max_blocks) {
  g_trace = (unsigned long long*)dev_buf;
  g_trace_blocks = dev_buf ? max_blocks : 0;
  return KVQ_OK;
}

extern "C" int kvq_gemm_splitk_factor(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 1;
  long tiles;
  int nk, bk;
  kvq::variant_tiles(M, N, K, false, &tiles, &nk, &bk);
  return kvq::splitk_factor(tiles, nk, bk);
}

extern "C" size_t kvq_gemm_splitk_bytes(int M, int N, int K) {
  // the conv front end may pick a different tile for the same (M, N, K): size for the larger factor of the two
  if (M <= 0 || N <= 0 || K <= 0 || K % 32) return 0;
  long t1, t2;
  int n1, n2, b1, b2;
  kvq::variant_tiles(M, N, K, false, &t1, &n1, &b1);
  kvq::variant_tiles(M, N, K, true, &t2, &n2, &b2);
  const int S = std::max(kvq::splitk_factor(t1, n1, b1), kvq::splitk_factor(t2, n2, b2));
  return S > 1 ? (size_t)S * M * N * sizeof(float) : 0;
}

extern "C" int kvq_gemm_bf16(const KvqGemmArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->A && a->W, KVQ_ERR_NULL, "kvq_gemm_bf16: NULL A/W");
  KVQ_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0 && a->N % 8 == 0 && a->K % 32 == 0, KVQ_ERR_SHAPE,
              "kvq_gemm_bf16: need M>0, N%%8==0, K%%32==0 (got M=%d N=%d K=%d)", a->M, a->N, a->K);
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED,
              "kvq_gemm_bf16: unknown dtype %d", a->dtype);
  GemmParams p{a->A, a->W, a->bias, a->M, a->N, a->K, a->out_bf16, a->out_f32, a->num_heads, a->q_scale,
               a->scatter_map, a->map_rows, a->out_rows, a->resid_bf16, a->resid_f32, g_trace, g_trace_blocks};
  p.ksplit = 1;
  p.sk_ws = (float*)a->splitk_ws; p.sk_bytes = a->splitk_ws_bytes;
  KVQ_REQUIRE(!p.sk_ws || ((size_t)p.sk_ws & 15) == 0, KVQ_ERR_SHAPE, "kvq_gemm_bf16: splitk_ws must be 16-byte aligned");
  p.ldc = a->ldc; p.col_off = a->col_off;
  p.a_gather = a->a_gather; p.a_rows = a->a_rows; p.a_phys_rows = a->a_phys_rows;
  KVQ_REQUIRE(!a->a_gather || (a->a_rows > 0 && a->a_phys_rows >= a->a_rows && a->M % a->a_rows == 0 && !a->splitk_ws), KVQ_ERR_SHAPE,
              "kvq_gemm_bf16: a_gather needs M = n_batch * a_rows, a_phys_rows >= a_rows, no split-K");
  KVQ_REQUIRE(a->ldc != 0 || a->col_off == 0, KVQ_ERR_SHAPE, "kvq_gemm_bf16: col_off = %d without ldc (the epilogue would write outside the row)",
              a->col_off);
  KVQ_REQUIRE(a->ldc == 0 || (a->epilogue != KVQ_EPI_QKV_BF16 && a->epilogue != KVQ_EPI_RESID_F32 && a->epilogue != KVQ_EPI_STORE_F32 &&
                              a->ldc % 8 == 0 && a->col_off % 8 == 0 && a->col_off >= 0 && a->col_off + a->N <= a->ldc),
              KVQ_ERR_SHAPE, "kvq_gemm_bf16: ldc / col_off need a 16-bit row-major epilogue, multiples of 8, col_off + N <= ldc");
  hipStream_t st = (hipStream_t)stream;
  switch (a->epilogue) {
    case KVQ_EPI_BIAS_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      return launch_dt<KVQ_EPI_BIAS_BF16>(a->dtype, p, st);
    case KVQ_EPI_GELU_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      return launch_dt<KVQ_EPI_GELU_BF16>(a->dtype, p, st);
    case KVQ_EPI_QGELU_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      return launch_dt<KVQ_EPI_QGELU_BF16>(a->dtype, p, st);
    case KVQ_EPI_RELU_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      return launch_dt<KVQ_EPI_RELU_BF16>(a->dtype, p, st);
    case KVQ_EPI_QKV_BF16:
      KVQ_REQUIRE(a->out_bf16, KVQ_ERR_NULL, "kvq_gemm_bf16: out_bf16 NULL");
      KVQ_REQUIRE(a->num_heads > 0 && a->N == 96 * a->num_heads, KVQ_ERR_SHAPE,
                  "kvq_gemm_bf16: QKV epilogue needs N == 3*32*num_heads (N=%d nH=%d)", a->N, a->num_heads);
      return launch_dt<KVQ_EPI_QKV_BF16>(a->dtype, p, st);
    case KVQ_EPI_RESID_F32:
      KVQ_REQUIRE(a->out_f32, KVQ_ERR_NULL, "kvq_gemm_bf16: out_f32 NULL");
      KVQ_REQUIRE(!a->scatter_map || (a->map_rows > 0 && a->out_rows > 0), KVQ_ERR_SHAPE,
                  "kvq_gemm_bf16: scatter map needs map_rows/out_rows");
      return launch_dt<KVQ_EPI_RESID_F32>(a->dtype, p, st);
    case KVQ_EPI_STORE_F32:
      KVQ_REQUIRE(a->out_f32, KVQ_ERR_NULL, "kvq_gemm_bf16: out_f32 NULL");
      return launch_dt<KVQ_EPI_STORE_F32>(a->dtype, p, st);
    default:
      set_error("kvq_gemm_bf16: unknown epilogue %d", a->epilogue);
      return KVQ_ERR_UNSUPPORTED;
  }
}

namespace kvq {
// q | k | v of the PADDING rows of a window partition: the reference computes qkv(0) = bias for them (F.pad after norm1,
// swin_backbone.py:416-449) and they take part in every window's softmax as keys.  One thread = 8 values of one (row, head).
template <typename E>
__global__ void qkv_fill_pad_kernel(uint16_t* out, const float* bias, const int32_t* pad_rows, int n_pad, int n_batch, int rows_per_batch,
                                    int num_heads, float q_scale) {
  const long gi = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)3 * num_heads * n_batch * n_pad * 4;
  if (gi >= total) return;
  const int e8 = (int)(gi & 3);
  long r = gi >> 2;
  const int pr = (int)(r % n_pad); r /= n_pad;
  const int b = (int)(r % n_batch); r /= n_batch;
  const int head = (int)(r % num_heads), which = (int)(r / num_heads);
  const float sc = which == 0 ? q_scale : 1.f;
  const float* bs = bias + (size_t)which * num_heads * 32 + head * 32 + 8 * e8;
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs), b1 = *reinterpret_cast<const f32x4*>(bs + 4);
  const size_t mtot = (size_t)n_batch * rows_per_batch, row = (size_t)b * rows_per_batch + pad_rows[pr];
  *reinterpret_cast<u32x4*>(out + ((size_t)(which * num_heads + head) * mtot + row) * 32 + 8 * e8) =
      (u32x4){E::pack2(b0[0] * sc, b0[1] * sc), E::pack2(b0[2] * sc, b0[3] * sc), E::pack2(b1[0] * sc, b1[1] * sc), E::pack2(b1[2] * sc, b1[3] * sc)};
}
}  // namespace kvq

extern "C" int kvq_qkv_fill_pad(void* qkv, const float* qkv_bias, const int32_t* pad_rows, int n_pad, int n_batch, int rows_per_batch,
                                int num_heads, float q_scale, int dtype, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(qkv && qkv_bias && pad_rows, KVQ_ERR_NULL, "kvq_qkv_fill_pad: NULL pointer");
  KVQ_REQUIRE(n_pad > 0 && n_batch > 0 && rows_per_batch >= n_pad && num_heads > 0, KVQ_ERR_SHAPE, "kvq_qkv_fill_pad: bad shape");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_qkv_fill_pad: dtype %d", dtype);
  const long total = (long)3 * num_heads * n_batch * n_pad * 4;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == KVQ_DT_FP16)
    hipLaunchKernelGGL(qkv_fill_pad_kernel<Fp16>, grid, block, 0, (hipStream_t)stream, (uint16_t*)qkv, qkv_bias, pad_rows, n_pad, n_batch,
                       rows_per_batch, num_heads, q_scale);
  else
    hipLaunchKernelGGL(qkv_fill_pad_kernel<Bf16>, grid, block, 0, (hipStream_t)stream, (uint16_t*)qkv, qkv_bias, pad_rows, n_pad, n_batch,
                       rows_per_batch, num_heads, q_scale);
  KVQ_CHECK_LAUNCH("qkv_fill_pad_kernel");
  return KVQ_OK;
}

extern "C" int kvq_conv_implicit(const KvqConvArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->x && a->W && (a->epilogue == KVQ_EPI_STORE_F32 ? (const void*)a->out_f32 : (const void*)a->out_bf16),
              KVQ_ERR_NULL, "kvq_conv_implicit: NULL pointer");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_conv_implicit: dtype %d", a->dtype);
  const int B = a->dims5[0], Cin = a->dims5[1], D = a->dims5[2], H = a->dims5[3], W = a->dims5[4];
  KVQ_REQUIRE(B > 0 && Cin > 0 && Cin % 8 == 0 && D > 0 && H > 0 && W > 0, KVQ_ERR_SHAPE,
              "kvq_conv_implicit: channels-last input needs C %% 8 == 0 (got B=%d C=%d D=%d H=%d W=%d)", B, Cin, D, H, W);
  for (int i = 0; i < 3; ++i)
    KVQ_REQUIRE(a->kernel3[i] > 0 && a->stride3[i] > 0 && a->pad3[i] >= 0, KVQ_ERR_SHAPE, "kvq_conv_implicit: bad kernel/stride/pad");
  const int Do = (D + 2 * a->pad3[0] - a->kernel3[0]) / a->stride3[0] + 1;
  const int Ho = (H + 2 * a->pad3[1] - a->kernel3[1]) / a->stride3[1] + 1;
  const int Wo = (W + 2 * a->pad3[2] - a->kernel3[2]) / a->stride3[2] + 1;
  // Kpad = 8 x the rows of the tap table.  Normally >= kd*kh*kw*C; a caller may leave out taps that fall into the padding for
  // EVERY output position (3x3 / pad 1 on a 1x1 map: the centre tap only) together with the matching columns of W.
  KVQ_REQUIRE(Do > 0 && Ho > 0 && Wo > 0 && a->Kpad >= 32 && a->Kpad % 32 == 0 && a->N > 0 && a->N % 8 == 0, KVQ_ERR_SHAPE,
              "kvq_conv_implicit: Kpad=%d must be a positive multiple of 32, N=%d a multiple of 8", a->Kpad, a->N);
  KVQ_REQUIRE((long)B * D * H * W * Cin < (1L << 31) && (long)B * Do * Ho * Wo < (1L << 31), KVQ_ERR_SHAPE,
              "kvq_conv_implicit: tensor too large for 32-bit tap offsets");
  KVQ_REQUIRE(a->epilogue == KVQ_EPI_RELU_BF16 || a->epilogue == KVQ_EPI_BIAS_BF16 || a->epilogue == KVQ_EPI_STORE_F32,
              KVQ_ERR_UNSUPPORTED, "kvq_conv_implicit: epilogue %d", a->epilogue);
  GemmParams p{};
  p.A = a->x; p.W = a->W; p.bias = a->bias; p.M = B * Do * Ho * Wo; p.N = a->N; p.K = a->Kpad;
  p.out_h = a->out_bf16; p.out_f32 = a->out_f32; p.resid_h = a->resid_bf16; p.resid_f32 = a->resid_f32;
  p.trace = g_trace; p.trace_blocks = g_trace_blocks;
  p.taps = reinterpret_cast<const int4*>(a->taps);
  p.ckd = a->kernel3[0]; p.ckh = a->kernel3[1]; p.ckw = a->kernel3[2];
  KVQ_REQUIRE(a->taps || (Cin % 32 == 0 && a->Kpad >= p.ckd * p.ckh * p.ckw * Cin), KVQ_ERR_SHAPE,
              "kvq_conv_implicit: without a tap table C=%d must be a multiple of 32 and Kpad=%d cover kd*kh*kw*C", Cin, a->Kpad);
  p.cD = D; p.cH = H; p.cW = W; p.cC = Cin; p.cDo = Do; p.cHo = Ho; p.cWo = Wo;
  p.csd = a->stride3[0]; p.csh = a->stride3[1]; p.csw = a->stride3[2];
  p.cpd = a->pad3[0]; p.cph = a->pad3[1]; p.cpw = a->pad3[2];
  p.ksplit = 1;
  p.sk_ws = (float*)a->splitk_ws; p.sk_bytes = a->splitk_ws_bytes;
  KVQ_REQUIRE(!p.sk_ws || ((size_t)p.sk_ws & 15) == 0, KVQ_ERR_SHAPE, "kvq_conv_implicit: splitk_ws must be 16-byte aligned");
  p.ldc = a->ldc; p.col_off = a->col_off;
  KVQ_REQUIRE(a->ldc != 0 || a->col_off == 0, KVQ_ERR_SHAPE, "kvq_conv_implicit: col_off = %d without ldc", a->col_off);
  KVQ_REQUIRE(a->ldc == 0 || (a->epilogue != KVQ_EPI_STORE_F32 && a->ldc % 8 == 0 && a->col_off % 8 == 0 && a->col_off >= 0 &&
                              a->col_off + a->N <= a->ldc),
              KVQ_ERR_SHAPE, "kvq_conv_implicit: ldc / col_off need a 16-bit epilogue, multiples of 8, col_off + N <= ldc");
  hipStream_t st = (hipStream_t)stream;
  if (a->epilogue == KVQ_EPI_STORE_F32)      // projection shortcuts: conv + BN, no ReLU, kept in fp32
    return a->dtype == KVQ_DT_FP16 ? launch_conv<Fp16, KVQ_EPI_STORE_F32>(p, st) : launch_conv<Bf16, KVQ_EPI_STORE_F32>(p, st);
  if (a->epilogue == KVQ_EPI_RELU_BF16)
    return a->dtype == KVQ_DT_FP16 ? launch_conv<Fp16, KVQ_EPI_RELU_BF16>(p, st) : launch_conv<Bf16, KVQ_EPI_RELU_BF16>(p, st);
  return a->dtype == KVQ_DT_FP16 ? launch_conv<Fp16, KVQ_EPI_BIAS_BF16>(p, st) : launch_conv<Bf16, KVQ_EPI_BIAS_BF16>(p, st);
}
