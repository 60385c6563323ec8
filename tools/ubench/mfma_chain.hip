// Micro-benchmark: cycles per v_mfma_f32_16x16x32_f16 (and 32x32x16) for different accumulator-dependency patterns, with one
// or two waves per SIMD, optionally with VALU / ds_read fillers between the MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/ubench/mfma_chain.hip ; run: /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// PAT: 0 = 24 independent accumulators; 1 = 2 chains of 12, one after the other (back-to-back dependent);
//      2 = 2 chains alternating (distance 2); 3 = 4 chains alternating (distance 4); 4 = 1 chain of 24
// FILL: number of independent v_fma_f32 between consecutive MFMAs; LDSR: a ds_read_b128 per MFMA
template <int PAT, int FILL, int LDSR>
__global__ void k16(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63;
  f32x4 acc[24];
  for (int i = 0; i < 24; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
  float f[4] = {1.f, 2.f, 3.f, 4.f};
  const unsigned char* lp = lds + lane * 16;
  f16x8 av[4] = {a, a, a, a};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const int idx = PAT == 0 ? k : PAT == 1 ? k / 12 : PAT == 2 ? k % 2 : PAT == 3 ? k % 4 : 0;
      f16x8 aa = a;
      if (LDSR) {
        aa = av[k % 4];
        av[k % 4] = *reinterpret_cast<const f16x8*>(lp + ((k + it) % 48) * 1024);
      }
      acc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa, b, acc[idx], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < FILL; ++q) {
        f[q % 4] = __builtin_fmaf(f[q % 4], 1.0001f, 0.5f);
        asm volatile("" : "+v"(f[q % 4]));
      }
      if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f[0] + f[1] + f[2] + f[3];
  for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// 32x32x16: one ds_read_b128 per MFMA through a rotating buffer of D fragments, FILL v_fma_f32 per MFMA
template <int D, int FILL>
__global__ void k32(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63;
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
  float f[4] = {1.f, 2.f, 3.f, 4.f};
  const unsigned char* lp = lds + lane * 16;
  f16x8 av[D];
#pragma unroll
  for (int i = 0; i < D; ++i) av[i] = a;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const f16x8 aa = av[k % D];
      acc[k % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b, acc[k % 6], 0, 0, 0);
      av[k % D] = *reinterpret_cast<const volatile f16x8*>(lp + k * 1024);
#pragma unroll
      for (int q = 0; q < FILL; ++q) {
        f[q % 4] = __builtin_fmaf(f[q % 4], 1.0001f, 0.5f);
        asm volatile("" : "+v"(f[q % 4]));
      }
      if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f[0] + f[1] + f[2] + f[3];
  for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int D, int FILL>
static void run32(const char* name, int threads, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k32<D, FILL>), dim3(256), dim3(threads), 0, 0, out, 10, cyc);
  hipLaunchKernelGGL((k32<D, FILL>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD %d: %6.1f cyc per MFMA per wave\n", name, threads / 256, (double)c / (iters * 24.0));
}

template <int PAT, int FILL, int LDSR>
static void run(const char* name, int threads, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k16<PAT, FILL, LDSR>), dim3(256), dim3(threads), 0, 0, out, 10, cyc);
  hipLaunchKernelGGL((k16<PAT, FILL, LDSR>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  printf("%-44s waves/SIMD %d: %6.1f cyc per MFMA per wave, %6.1f per SIMD\n", name, wps, (double)c / (iters * 24.0),
         (double)c / (iters * 24.0 * wps));
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run32<4, 0>("32x32x16 + ds_read_b128 (4 deep)", 256, out, cyc);
  run32<8, 0>("32x32x16 + ds_read_b128 (8 deep)", 256, out, cyc);
  run32<12, 0>("32x32x16 + ds_read_b128 (12 deep)", 256, out, cyc);
  run32<8, 4>("32x32x16 + ds_read_b128 (8 deep) + 4 fma", 256, out, cyc);
  run32<8, 6>("32x32x16 + ds_read_b128 (8 deep) + 6 fma", 256, out, cyc);
  run32<8, 8>("32x32x16 + ds_read_b128 (8 deep) + 8 fma", 256, out, cyc);
  run32<8, 0>("32x32x16 + ds_read_b128 (8 deep)", 512, out, cyc);
  for (int threads : {256}) {
    run<0, 0, 0>("24 independent", threads, out, cyc);
    run<1, 0, 0>("2 chains of 12, sequential", threads, out, cyc);
    run<2, 0, 0>("2 chains alternating", threads, out, cyc);
    run<3, 0, 0>("4 chains alternating", threads, out, cyc);
    run<4, 0, 0>("1 chain of 24", threads, out, cyc);
    run<0, 2, 0>("24 independent + 2 fma", threads, out, cyc);
    run<0, 4, 0>("24 independent + 4 fma", threads, out, cyc);
    run<0, 8, 0>("24 independent + 8 fma", threads, out, cyc);
    run<2, 2, 0>("2 chains alternating + 2 fma", threads, out, cyc);
    run<2, 4, 0>("2 chains alternating + 4 fma", threads, out, cyc);
    run<1, 4, 0>("2 chains sequential + 4 fma", threads, out, cyc);
    run<3, 4, 0>("4 chains alternating + 4 fma", threads, out, cyc);
    run<0, 0, 1>("24 independent + ds_read_b128", threads, out, cyc);
    run<2, 0, 1>("2 chains alternating + ds_read_b128", threads, out, cyc);
    run<0, 4, 1>("24 independent + ds_read_b128 + 4 fma", threads, out, cyc);
    run<2, 4, 1>("2 chains alt + ds_read_b128 + 4 fma", threads, out, cyc);
  }
  return 0;
}
