// 256 x 256 x 64 "eight-phase" MFMA main loop for gfx950 (MI355X) — the wide-tile GEMM of the trunk's C >= 256 stages
// (swin_backbone.py:64-89 Mlp, :252-326 qkv / proj, :533-556 reduction) and of the conv nets' implicit GEMMs.
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]        A [M][K], W [N][K] 16-bit, fp32 accumulation
//
// Why a second main loop beside gemm_kernel (gemm.hip): that one ingests a 128 x 128 x 32 slice per barrier (64-byte row
// segments, <= 48 KB in flight per CU) and tops out at ~700 TFLOP/s; the bytes loaded per flop have to halve and the
// operand stream has to stay in flight ACROSS the barriers.  Structure (MI355X guide, section 5 "8-phase"):
//
// * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns a contiguous 128 x 64 block of the tile = 2 x 2 "quadrants" of
//   64 x 32, each 2 x 1 v_mfma_f32_32x32x16 tiles: 128 accumulator registers.
// * A K-tile (64 deep) is FOUR half-tiles of 16 KB: B0, A0, B1, A1 (A_i = the i-th 64-row half of every wave row,
//   B_j = the j-th 32-column half of every wave column: 128 rows x 128 B each).  LDS holds two K-tiles = 8 slots = 128 KB.
// * One half-tile is requested per phase by LDS-DMA through a buffer resource (buffer_load_dwordx4 ... lds: per-lane
//   voffset, the K position is the scalar soffset: no VALU on the issue path, rows past M / N read zeros), SEVEN half-tiles
//   ahead of the phase that reads it; the only vmcnt wait is a counted one (3 half-tiles stay in flight) once per K-tile.
// * A phase = [fragment reads + 1 half-tile request] s_barrier [8 MFMAs = one quadrant x K 64] s_barrier.  The two wave
//   rows run staggered by one barrier: while the four waves of one row (one per SIMD) multiply, the other four read —
//   every SIMD always has exactly one wave in its MFMA section.
// * quadrant order (0,0) (0,1) (1,1) (1,0): 12 / 4 / 8 / 0 fragment reads per phase, A fragments share one register set.
// * LDS image rows are 128 B (lane-linear per DMA: 8 rows x 128 B per wave instruction); the 16-B chunk c of row r sits at
//   chunk c ^ ((r >> 1) & 7) — applied on the DMA's SOURCE address and on the fragment read: conflict-free ds_read_b128.
//
// Hazards (all by barrier count, see the phase table in DESIGN.md section 4):
//   RAW  half-tiles of tile t are waited for (vmcnt) before the first barrier of phase P3 of tile t-1 and first read in
//        P0 of tile t — at least one barrier later for either wave row.
//   WAR  a slot is re-requested two phases after the phase that read it (one full barrier of margin for the staggered row);
//        B0 one phase after, which is why P0 issues its 4 B reads first and waits lgkmcnt(8) before its barrier.
#pragma once
#include "common.hpp"

namespace kvq {
namespace g8 {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KB
constexpr int LDS_BYTES = 8 * HALF_BYTES;         // two K-tiles
// slot of a half-tile inside a K-tile buffer
constexpr int SLOT_B0 = 0, SLOT_A0 = 1, SLOT_B1 = 2, SLOT_A1 = 3;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

template <typename V>
__device__ __forceinline__ void ds_read16(V& dst, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory");
}

// buffer resource over [base, base + bytes): raw (stride 0), out-of-range reads return 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
  const unsigned nr = bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)nr, 0x00020000);
}

// Operand source of the main loop.  Plain row-major [rows][K] operands: voffset = row * K * 2 + chunk * 16, the K position
// is the scalar offset.  (An implicit-convolution source would plug in here: same issue<H>() / advance() interface; none exists yet.)
struct PlainSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned vo[2][2];                          // [half][q] per-lane byte offsets
  int kt0;                                    // first K-tile of this workgroup's K range (split-K)
  // row[half][q]: row of this lane's chunk RELATIVE to `base` (rows >= rows_avail read zeros)
  __device__ __forceinline__ void init(const uint16_t* base, size_t rows_avail, int K, const int (&row)[2][2], int lc, int kt0_) {
    rsrc = make_rsrc(base, rows_avail * K * 2);
    kt0 = kt0_;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) vo[h][q] = (unsigned)row[h][q] * (unsigned)(K * 2) + (unsigned)lc * 16u;
  }
  template <int H>
  __device__ __forceinline__ void issue(unsigned char* dst, int kt) const {       // dst: this wave's 1 KB of piece q = 0
    const int so = (kt0 + kt) * (BK * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, vo[H][0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + HALF_BYTES / 2), 16, vo[H][1], so, 0, 0);
  }
  __device__ __forceinline__ void advance(int) {}
};

// The main loop.  acc[i][mi][j]: quadrant (i, j), m-tile mi.  Wave (wr, wc) ends up with rows wr*128 + i*64 + mi*32 + [0,32),
// columns wc*64 + j*32 + [0,32) of the tile.  SrcA / SrcB: issue<H>(dst, kt) requests this thread's two 16-B chunks of half H of
// K-tile kt.  Ends with every DMA landed and a barrier: LDS is free for the epilogue.
template <typename E, typename SrcA, typename SrcB>
__device__ __forceinline__ void mainloop(unsigned char* lds, SrcA& sa, SrcB& sb, int nk, f32x16 (&acc)[2][2][2]) {
  using V8 = typename E::v8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int frow = lane & 31, fkg = lane >> 5;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)lds;
  // fragment addresses (K-tile buffer 0, slot 0), one per k-step: the swizzle depends on (row >> 1) & 7 = (frow >> 1) & 7 only
  unsigned a_addr[4], b_addr[4];
  {
    const int sw = (frow >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const unsigned ch = (unsigned)(((kk * 2 + fkg) ^ sw) << 4);
      a_addr[kk] = lds_base + (unsigned)((wr * 64 + frow) * 128) + ch;
      b_addr[kk] = lds_base + (unsigned)((wc * 32 + frow) * 128) + ch;
    }
  }
  unsigned char* const my = lds + wave * 1024;          // this wave's 1 KB inside piece q = 0 of a slot
  auto stage = [&](int tt, auto slot_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_tag)::value;
    const int kt = tt < nk ? tt : nk - 1;               // past the end: re-request the last tile (keeps the vmcnt arithmetic uniform)
    unsigned char* dst = my + ((tt & 1) * 4 + S) * HALF_BYTES;
    if (S == SLOT_A0) sa.template issue<0>(dst, kt);
    else if (S == SLOT_A1) sa.template issue<1>(dst, kt);
    else if (S == SLOT_B0) sb.template issue<0>(dst, kt);
    else sb.template issue<1>(dst, kt);
  };
  using S_B0 = std::integral_constant<int, SLOT_B0>;
  using S_A0 = std::integral_constant<int, SLOT_A0>;
  using S_B1 = std::integral_constant<int, SLOT_B1>;
  using S_A1 = std::integral_constant<int, SLOT_A1>;

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][mi][j][r] = 0.f;

  // prologue: half-tiles 0 .. 6 (tile 0 whole, tile 1 without A1)
  stage(0, S_B0{}); stage(0, S_A0{}); stage(0, S_B1{}); stage(0, S_A1{});
  stage(1, S_B0{}); stage(1, S_A0{}); stage(1, S_B1{});
  wait_vmcnt<6>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();            // stagger the second wave row by one barrier

  V8 a[2][4], b0[4], b1[4];
  auto mfma_quadrant = [&](auto i_tag, auto j_tag) __attribute__((always_inline)) {
    constexpr int I = decltype(i_tag)::value, J = decltype(j_tag)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[I][mi][J] = E::mfma32(a[mi][kk], J == 0 ? b0[kk] : b1[kk], acc[I][mi][J]);
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  unsigned bufstep = 4u * HALF_BYTES;
  for (int t = 0; t < nk; ++t) {
    // ---- P0: quadrant (0,0).  reads: B0 (4, first), A0 (8); requests A1 of tile t+1
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ds_read16(b0[kk], b_addr[kk], SLOT_B0 * HALF_BYTES);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) ds_read16(a[mi][kk], a_addr[kk], SLOT_A0 * HALF_BYTES + mi * 4096);
    stage(t + 1, S_A1{});
    wait_lgkmcnt<8>();                                  // the B0 reads have returned: its slot may be re-requested next phase
    __builtin_amdgcn_s_barrier();
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P1: quadrant (0,1).  reads: B1 (4); requests B0 of tile t+2
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ds_read16(b1[kk], b_addr[kk], SLOT_B1 * HALF_BYTES);
    stage(t + 2, S_B0{});
    __builtin_amdgcn_s_barrier();
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(I0{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P2: quadrant (1,1).  reads: A1 (8, into the A registers); requests A0 of tile t+2
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) ds_read16(a[mi][kk], a_addr[kk], SLOT_A1 * HALF_BYTES + mi * 4096);
    stage(t + 2, S_A0{});
    __builtin_amdgcn_s_barrier();
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P3: quadrant (1,0).  no reads; requests B1 of tile t+2; tile t+1 must have landed: 3 half-tiles stay in flight
    stage(t + 2, S_B1{});
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(I1{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // next K-tile buffer
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { a_addr[kk] += bufstep; b_addr[kk] += bufstep; }
    bufstep = 0u - bufstep;
    sa.advance(t);
    sb.advance(t);
  }
  wait_vmcnt<0>();
  if (wr == 0) __builtin_amdgcn_s_barrier();            // un-stagger
  __builtin_amdgcn_s_barrier();
}

// XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed; a speed assumption only): every XCD gets a CONTIGUOUS range
// of the logical tile index, and inside it tiles are ordered in groups of GM tile-rows, N fastest inside a group's column —
// the 32 tiles an XCD works on at once form a GM x (32 / GM) block: GM + 32 / GM operand panels instead of 33.
__device__ __forceinline__ int logical_block() {
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, qd = nwg >> 3, rm = nwg & 7;
  return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
}
__device__ __forceinline__ void tile_of(int lid, int nbm, int nbn, int& bm, int& bn) {
  constexpr int GM = 4;
  const int per = GM * nbn, grp = lid / per, first = grp * GM;
  const int gsz = nbm - first < GM ? nbm - first : GM;
  const int in = lid - grp * per;
  bm = first + in % gsz;
  bn = in / gsz;
}

// per-lane staging geometry shared by both operands: chunk c_lin = q * 512 + tid of a half-tile image
struct StageGeom {
  int a_row[2][2], b_row[2][2], lc;
  __device__ __forceinline__ void init() {
    const int tid = threadIdx.x;
    const int r64 = tid >> 3, pc = tid & 7;
    lc = pc ^ ((r64 >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        a_row[h][q] = q * 128 + h * 64 + r64;                                  // image row q*64 + r64 of A_h: wave row q
        b_row[h][q] = (q * 2 + (tid >> 8)) * 64 + h * 32 + (r64 & 31);        // image row q*64 + r64 of B_h: wave column 2q + (r64 >> 5)
      }
  }
};

}  // namespace g8
}  // namespace kvq
