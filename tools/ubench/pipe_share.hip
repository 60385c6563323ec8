// Micro-benchmark: which instruction classes of DIFFERENT waves on one SIMD overlap?  One workgroup of 4*W waves on one CU
// (W per SIMD); every wave runs the same loop of NM v_mfma_f32_16x16x32_f16 + NE v_exp_f32 + NF v_fma_f32 per iteration
// (the compiler interleaves them); reported: the SLOWEST wave's cycles per iteration divided by W = SIMD cycles per
// wave-iteration.  Perfect overlap -> max(pipe times); none -> their sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NM, int NE, int NF, int BIG>
__global__ void kern(float* out, int iters, unsigned long long* cyc) {
  f32x4 acc[8];
  f32x16 big[2];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * e); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  float c1 = 1.0001f, c2 = 0.5f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NM; ++k) {
      if (BIG) big[k % 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[k % 2], 0, 0, 0);
      else acc[k % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k % 8], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k % 8]));
#pragma unroll
    for (int k = 0; k < NF; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[8 + k % 8]) : "v"(c1), "v"(c2));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  s += big[0][0] + big[1][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NM, int NE, int NF, int BIG>
void run(float* out, unsigned long long* cyc) {
  const int iters = 200;
  printf("%2d mfma%s + %3d exp + %3d fma:", NM, BIG ? "32" : "16", NE, NF);
  for (int wps = 1; wps <= 4; ++wps) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((kern<NM, NE, NF, BIG>), dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
      (void)hipDeviceSynchronize();
    }
    unsigned long long c[16];
    (void)hipMemcpy(c, cyc, 8 * 4 * wps, hipMemcpyDeviceToHost);
    double mn = 1e30, mx = 0;
    for (int i = 0; i < 4 * wps; ++i) { mn = c[i] < mn ? c[i] : mn; mx = c[i] > mx ? c[i] : mx; }
    printf("  %dw: %6.1f (min %6.1f)", wps, mx / iters / wps, mn / iters);
  }
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 4);
  (void)hipMalloc(&cyc, 256);
  run<16, 0, 0, 0>(out, cyc);
  run<0, 16, 0, 0>(out, cyc);
  run<0, 0, 48, 0>(out, cyc);
  run<0, 16, 48, 0>(out, cyc);
  run<16, 0, 48, 0>(out, cyc);
  run<16, 0, 96, 0>(out, cyc);
  run<16, 16, 0, 0>(out, cyc);
  run<16, 32, 0, 0>(out, cyc);
  run<16, 16, 48, 0>(out, cyc);
  run<16, 32, 96, 0>(out, cyc);
  run<8, 0, 0, 1>(out, cyc);
  run<8, 16, 0, 1>(out, cyc);
  run<8, 0, 48, 1>(out, cyc);
  run<8, 16, 48, 1>(out, cyc);
  run<64, 104, 326, 0>(out, cyc);
  return 0;
}
