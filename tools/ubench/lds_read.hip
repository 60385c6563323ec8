// Micro-benchmark: ds_read_b128 throughput per CU (1 KB per wave-instruction, conflict-free lane*16 addressing), with
// NB reads in flight per wave, for 4 / 8 / 12 / 16 waves per CU; optionally with one 16x16x32 MFMA per read.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_read.bin tools/ubench/lds_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NB, int MFMA>
__global__ void k(unsigned* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(lds) + lane * 16;
  u32x4 v[NB];
  f32x4 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) { v[i] = (u32x4){0, 0, 0, 0}; acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  u32x4 sum = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NB; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(i * 1024));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, v[i]), __builtin_bit_cast(f16x8, v[i]), acc[i], 0, 0, 0);
      else sum ^= v[i];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  for (int i = 0; i < NB; ++i) sum[0] ^= __builtin_bit_cast(unsigned, acc[i][0]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum[0] ^ sum[1] ^ sum[2] ^ sum[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NB, int MFMA>
static void run(int threads, unsigned* out, unsigned long long* cyc) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<NB, MFMA>), dim3(256), dim3(threads), 0, 0, out, 10, cyc);
  hipLaunchKernelGGL((k<NB, MFMA>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_read = (double)c / (iters * (double)NB);
  printf("waves/CU %2d, %2d reads in flight%s: %6.1f ticks per read per wave, %6.1f B/tick/CU\n", threads / 64, NB,
         MFMA ? " + MFMA each" : "            ", per_read, 1024.0 * (threads / 64) / per_read);
}

int main() {
  unsigned* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  for (int threads : {256, 512, 768, 1024}) {
    run<4, 0>(threads, out, cyc);
    run<8, 0>(threads, out, cyc);
    run<16, 0>(threads, out, cyc);
    run<8, 1>(threads, out, cyc);
    run<16, 1>(threads, out, cyc);
  }
  return 0;
}
