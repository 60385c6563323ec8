// Micro-benchmark: the attention q-tile's instruction mix (64 v_mfma_16x16x32 + 104 v_exp + 326 plain VALU per iteration,
// no memory) on W identical waves per SIMD, with the kinds (a) interleaved finely, (b) separated into long homogeneous
// phases by sched_barrier.  Reported: slowest wave's cycles per iteration / W (= SIMD cycles per wave-iteration).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(k) acc[(k) % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(k) % 8], 0, 0, 0)
#define EXP(k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(k) % 8]))
#define FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[8 + (k) % 8]) : "v"(c1), "v"(c2))
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int PAT>
__global__ void kern(float* out, int iters, unsigned long long* cyc) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * e); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  float c1 = 1.0001f, c2 = 0.5f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (PAT == 0) {            // fine interleave: {mfma, 2 exp (first 52 only), 5 fma} x 64
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        MFMA(k); SB();
        if (k < 52) { EXP(2 * k); EXP(2 * k + 1); }
#pragma unroll
        for (int q = 0; q < 5; ++q) FMA(5 * k + q);
        SB();
      }
    } else if (PAT == 1) {     // two phases: 64 mfma | 104 exp + 326 fma
#pragma unroll
      for (int k = 0; k < 64; ++k) MFMA(k);
      SB();
#pragma unroll
      for (int k = 0; k < 104; ++k) { EXP(k); FMA(3 * k); FMA(3 * k + 1); FMA(3 * k + 2); }
#pragma unroll
      for (int k = 0; k < 14; ++k) FMA(k);
      SB();
    } else if (PAT == 2) {     // attention order: 104 fma (cvt) | 25 mfma | 120 fma (max, sub) + 104 exp + 52 fma (pack) | 39 mfma | 50 fma
#pragma unroll
      for (int k = 0; k < 104; ++k) FMA(k);
      SB();
#pragma unroll
      for (int k = 0; k < 25; ++k) MFMA(k);
      SB();
#pragma unroll
      for (int k = 0; k < 120; ++k) FMA(k);
#pragma unroll
      for (int k = 0; k < 52; ++k) { EXP(2 * k); EXP(2 * k + 1); FMA(k); }
      SB();
#pragma unroll
      for (int k = 0; k < 39; ++k) MFMA(k);
      SB();
#pragma unroll
      for (int k = 0; k < 50; ++k) FMA(k);
      SB();
    } else if (PAT == 3) {     // as the compiler schedules today: {4 fma, mfma} x 25 | 67 fma | {8 exp, 8 fma, 4 fma, 3 mfma} x 13 | 50 fma
#pragma unroll
      for (int k = 0; k < 25; ++k) { FMA(4 * k); FMA(4 * k + 1); FMA(4 * k + 2); FMA(4 * k + 3); SB(); MFMA(k); SB(); }
#pragma unroll
      for (int k = 0; k < 67; ++k) FMA(k);
      SB();
#pragma unroll
      for (int g = 0; g < 13; ++g) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { FMA(k); EXP(k); }
#pragma unroll
        for (int k = 0; k < 4; ++k) FMA(k);
        SB();
        MFMA(3 * g); MFMA(3 * g + 1); MFMA(3 * g + 2);
        SB();
      }
#pragma unroll
      for (int k = 0; k < 50; ++k) FMA(k);
      SB();
    } else if (PAT == 4) {     // VALU diet: 52 cvt-ish | 25 mfma | 52 max3 + 52 pk_fma + 104 exp + 52 pack | 39 mfma | 30
#pragma unroll
      for (int k = 0; k < 104; ++k) FMA(k);
      SB();
#pragma unroll
      for (int k = 0; k < 25; ++k) MFMA(k);
      SB();
#pragma unroll
      for (int k = 0; k < 52; ++k) FMA(k);
#pragma unroll
      for (int k = 0; k < 52; ++k) { EXP(2 * k); EXP(2 * k + 1); FMA(k); FMA(k + 1); }
      SB();
#pragma unroll
      for (int k = 0; k < 39; ++k) MFMA(k);
      SB();
#pragma unroll
      for (int k = 0; k < 30; ++k) FMA(k);
      SB();
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

static const char* NAMES[] = {"fine interleave {mfma, 2 exp, 5 fma}", "2 phases: 64 mfma | all VALU", "attention order, separated phases",
                              "attention order, as scheduled today", "separated phases, VALU diet (268 + 104 exp)"};
template <int PAT>
void run(float* out, unsigned long long* cyc) {
  const int iters = 100;
  printf("%-46s", NAMES[PAT]);
  for (int wps = 1; wps <= 4; ++wps) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kern<PAT>, dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
      (void)hipDeviceSynchronize();
    }
    unsigned long long c[16];
    (void)hipMemcpy(c, cyc, 8 * 4 * wps, hipMemcpyDeviceToHost);
    double mn = 1e30, mx = 0;
    for (int i = 0; i < 4 * wps; ++i) { mn = c[i] < mn ? c[i] : mn; mx = c[i] > mx ? c[i] : mx; }
    printf("  %dw: %6.0f (min %6.0f)", wps, mx / iters / wps, mn / iters);
  }
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 4);
  (void)hipMalloc(&cyc, 256);
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc);
  return 0;
}
