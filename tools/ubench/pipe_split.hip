// Micro-benchmark: one SIMD, two KINDS of waves: "matrix" waves run only v_mfma, "vector" waves only v_fma / v_exp.
// A workgroup has 4*(WM+WV) waves; wave w is a matrix wave when (w / 4) < WM (waves go to SIMDs cyclically, so every SIMD
// gets WM matrix + WV vector waves).  Every wave reports cycles per iteration: if the pipes are concurrent on one SIMD the
// matrix waves keep their solo time; if not, they stretch by the vector waves' issue time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BIG, int TRANS>
__global__ void kern(float* out, int iters, int wm, unsigned long long* cyc) {
  f32x4 acc[8];
  f32x16 big[2];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * e); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
  float c1 = 1.0001f, c2 = 0.5f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool matrix = (wave >> 2) < wm;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (matrix) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (BIG) { if (k < 8) big[k % 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[k % 2], 0, 0, 0); }
        else acc[k % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k % 8], 0, 0, 0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 48; ++k) {
        if (TRANS) { if (k < 16) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k % 8])); }
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k % 8]) : "v"(c1), "v"(c2));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + acc[i][0] + acc[i][3];
  s += big[0][0] + big[1][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int BIG, int TRANS>
void run(float* out, unsigned long long* cyc, int wm, int wv) {
  const int iters = 400;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((kern<BIG, TRANS>), dim3(1), dim3(256 * (wm + wv)), 0, 0, out, iters, wm, cyc);
    (void)hipDeviceSynchronize();
  }
  unsigned long long c[16];
  (void)hipMemcpy(c, cyc, 8 * 4 * (wm + wv), hipMemcpyDeviceToHost);
  printf("%s x%d + %s x%d per SIMD: matrix waves", BIG ? "8 mfma32" : "16 mfma16", wm, TRANS ? "16 exp" : "48 fma", wv);
  for (int i = 0; i < 4 * wm; ++i) printf(" %6.1f", (double)c[i] / iters);
  printf(" | vector waves");
  for (int i = 4 * wm; i < 4 * (wm + wv); ++i) printf(" %6.1f", (double)c[i] / iters);
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 4);
  (void)hipMalloc(&cyc, 256);
  run<0, 0>(out, cyc, 1, 0); run<0, 0>(out, cyc, 0, 1); run<0, 0>(out, cyc, 1, 1); run<0, 0>(out, cyc, 1, 2); run<0, 0>(out, cyc, 1, 3);
  run<0, 0>(out, cyc, 2, 2);
  run<0, 1>(out, cyc, 0, 1); run<0, 1>(out, cyc, 1, 1); run<0, 1>(out, cyc, 1, 2); run<0, 1>(out, cyc, 1, 3);
  run<1, 0>(out, cyc, 1, 0); run<1, 0>(out, cyc, 1, 1); run<1, 0>(out, cyc, 1, 2); run<1, 0>(out, cyc, 1, 3);
  run<1, 1>(out, cyc, 1, 1); run<1, 1>(out, cyc, 1, 2);
  return 0;
}
