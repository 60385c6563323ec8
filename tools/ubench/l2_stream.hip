// Micro-benchmark: how fast can one CU pull L2-resident bytes (a) by LDS-DMA, (b) global -> VGPR (-> LDS)?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_stream tools/ubench/l2_stream.hip ; run: /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// each workgroup streams `iters` x (waves x 1 KB x UNROLL) from a window of `win` bytes
// ROWB > 0: the GEMM's operand pattern — a wave-load covers 1024 / ROWB rows of ROWB contiguous bytes, rows `pitch` bytes apart
// (ROWB = 64: a 32-deep K slice of 16-bit operands; 128: a 64-deep one) instead of 1 KB contiguous
template <int MODE, int UNROLL, int ROWB = 0>
__global__ void stream_kernel(const unsigned char* src, size_t win, int iters, unsigned* sink, int pitch = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  size_t off = ((size_t)blockIdx.x * 7919 * 1024) % win;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t o = (off + (size_t)(u * nw + wave) * 1024) % win;
      if (MODE == 0 && ROWB > 0) {
        constexpr int LPR = ROWB / 16;                       // lanes per row
        const size_t ro = (off + ((size_t)(u * nw + wave) * (64 / LPR) + lane / LPR) * pitch + (size_t)(it % (pitch / ROWB)) * ROWB) % win;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + ro + (lane % LPR) * 16), (lds_ptr_t)(lds + ((u * nw + wave) % 32) * 1024), 16, 0, 0);
      } else if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + o + lane * 16), (lds_ptr_t)(lds + ((u * nw + wave) % 32) * 1024), 16, 0, 0);
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + o + lane * 16);
        if (MODE == 2) *reinterpret_cast<u32x4*>(lds + ((u * nw + wave) % 32) * 1024 + lane * 16) = v;
        else acc ^= v;
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ROWB == 0) off = (off + (size_t)UNROLL * nw * 1024) % win;
  }
  __syncthreads();
  if (MODE != 1) acc[0] ^= reinterpret_cast<unsigned*>(lds)[threadIdx.x];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int UNROLL, int ROWB = 0>
static void run(const char* name, const unsigned char* d, size_t win, int waves, int blocks_per_cu, unsigned* sink, int pitch = 0) {
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((stream_kernel<MODE, UNROLL, ROWB>), dim3(grid), dim3(64 * waves), 32768, 0, d, win, iters, sink, pitch);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * iters * UNROLL * waves * 1024;
  printf("%-28s waves/WG %2d  WG/CU %d  unroll %d : %7.2f TB/s  = %6.1f GB/s per CU\n", name, waves, blocks_per_cu, UNROLL, bytes / ms / 1e9,
         bytes / ms / 1e6 / 256);
}

int main() {
  const size_t win = 2u << 20;     // 2 MB window: L2-resident per XCD
  unsigned char* d; unsigned* sink;
  hipMalloc(&d, win + (1 << 20)); hipMemset(d, 1, win + (1 << 20)); hipMalloc(&sink, 4);
  for (int waves : {4, 8}) for (int bpc : {1, 2, 3}) {
    run<0, 4>("LDS-DMA (global_load_lds)", d, win, waves, bpc, sink);
    run<1, 4>("global -> VGPR", d, win, waves, bpc, sink);
    run<2, 4>("global -> VGPR -> ds_write", d, win, waves, bpc, sink);
  }
  run<0, 8>("LDS-DMA (global_load_lds)", d, win, 4, 3, sink);
  run<1, 8>("global -> VGPR", d, win, 4, 3, sink);
  // the GEMM operand pattern (rows of 64 / 128 / 256 B, 6 KB apart = K of 3072 16-bit values), L2-resident window
  for (int bpc : {1, 3}) {
    run<0, 4, 64>("LDS-DMA, 64-B rows", d, win, 4, bpc, sink, 6144);
    run<0, 4, 128>("LDS-DMA, 128-B rows", d, win, 4, bpc, sink, 6144);
    run<0, 4, 256>("LDS-DMA, 256-B rows", d, win, 4, bpc, sink, 6144);
  }
  return 0;
}
