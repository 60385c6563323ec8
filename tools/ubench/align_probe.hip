// Micro-benchmark: does the byte placement of a loop that mixes v_mfma and v_exp change how several waves of one SIMD
// share it?  (valu_rate.hip's `mixed<16,16,0>` and exp_mfma.hip's pattern 0 compile to the SAME loop body and run 2.6x
// apart with 3 waves per SIMD.)  PAD s_nop (4 bytes each) are placed before the loop; every wave reports its cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int PAD>
__global__ void kern(float* out, int iters, unsigned long long* cyc) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * e); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
  __syncthreads();
  asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(PAD));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k % 8], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k % 8]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int PAD>
void run(float* out, unsigned long long* cyc) {
  const int iters = 200;
  printf("pad %2d:", PAD);
  for (int wps = 1; wps <= 4; ++wps) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kern<PAD>, dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
      (void)hipDeviceSynchronize();
    }
    unsigned long long c[16];
    (void)hipMemcpy(c, cyc, 8 * 4 * wps, hipMemcpyDeviceToHost);
    double mn = 1e30, mx = 0;
    for (int i = 0; i < 4 * wps; ++i) { mn = c[i] < mn ? c[i] : mn; mx = c[i] > mx ? c[i] : mx; }
    printf("  %dw: %6.1f..%6.1f", wps, mn / iters, mx / iters);
  }
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 4);
  (void)hipMalloc(&cyc, 256);
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<5>(out, cyc); run<6>(out, cyc);
  run<7>(out, cyc); run<8>(out, cyc); run<9>(out, cyc); run<10>(out, cyc); run<11>(out, cyc); run<12>(out, cyc); run<13>(out, cyc);
  run<14>(out, cyc); run<15>(out, cyc); run<16>(out, cyc);
  return 0;
}
