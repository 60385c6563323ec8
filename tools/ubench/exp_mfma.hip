// Micro-benchmark: how do v_exp_f32 (transcendental) and v_mfma_f32_16x16x32_f16 streams of SEVERAL waves on one SIMD share
// it, depending on how the two kinds are ordered inside each wave's stream?  (valu_rate.hip: "16 mfma then 16 exp" per
// iteration does not speed up with more waves at all, while "16 mfma, 16 exp, 48 fma" scales perfectly.)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_mfma tools/ubench/exp_mfma.hip ; run: /tmp/exp_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(k) acc[(k) % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(k) % 8], 0, 0, 0)
#define EXP(k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(k) % 8]))
#define FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[8 + (k) % 8]) : "v"(c1), "v"(c2))
#define NOP(n) asm volatile("s_nop %0" ::"n"(n))
#define CVT(k) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[8 + (k) % 8]) : "v"(c1))

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
    if (PAT == 0) {          // 16 mfma, 16 exp
#pragma unroll
      for (int k = 0; k < 16; ++k) MFMA(k);
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
    } else if (PAT == 1) {   // interleaved 1:1
#pragma unroll
      for (int k = 0; k < 16; ++k) { MFMA(k); EXP(k); }
    } else if (PAT == 2) {   // 16 mfma, 1 fma, 16 exp, 1 fma
#pragma unroll
      for (int k = 0; k < 16; ++k) MFMA(k);
      FMA(0);
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
      FMA(1);
    } else if (PAT == 3) {   // 16 mfma, 4 fma, 16 exp, 4 fma
#pragma unroll
      for (int k = 0; k < 16; ++k) MFMA(k);
#pragma unroll
      for (int k = 0; k < 4; ++k) FMA(k);
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
#pragma unroll
      for (int k = 0; k < 4; ++k) FMA(k + 4);
    } else if (PAT == 4) {   // 16 mfma, 16 exp, 16 fma  (fma only between exp and the next mfma)
#pragma unroll
      for (int k = 0; k < 16; ++k) MFMA(k);
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
#pragma unroll
      for (int k = 0; k < 16; ++k) FMA(k);
    } else if (PAT == 5) {   // 16 mfma, 16 fma, 16 exp  (fma only between mfma and exp)
#pragma unroll
      for (int k = 0; k < 16; ++k) MFMA(k);
#pragma unroll
      for (int k = 0; k < 16; ++k) FMA(k);
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
    } else if (PAT == 6) {   // {mfma, exp, fma, fma} x 16
#pragma unroll
      for (int k = 0; k < 16; ++k) { MFMA(k); EXP(k); FMA(2 * k); FMA(2 * k + 1); }
    } else if (PAT == 7) {   // {mfma, fma, exp, fma} x 16
#pragma unroll
      for (int k = 0; k < 16; ++k) { MFMA(k); FMA(2 * k); EXP(k); FMA(2 * k + 1); }
    } else if (PAT == 8) {   // 16 mfma, s_nop 7, 16 exp, s_nop 7
#pragma unroll
      for (int k = 0; k < 16; ++k) MFMA(k);
      NOP(7);
#pragma unroll
      for (int k = 0; k < 16; ++k) EXP(k);
      NOP(7);
    } else if (PAT == 9) {   // {mfma x4, exp x4} x 4
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) MFMA(4 * g + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) EXP(4 * g + k);
      }
    } else if (PAT == 10) {  // {mfma x4, fma, exp x4, fma} x 4
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) MFMA(4 * g + k);
        FMA(2 * g);
#pragma unroll
        for (int k = 0; k < 4; ++k) EXP(4 * g + k);
        FMA(2 * g + 1);
      }
    } else if (PAT == 11) {  // the attention PV phase as hipcc emits it: {exp x8, mfma x3} x 5 + 1 mfma
#pragma unroll
      for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int k = 0; k < 8; ++k) EXP(8 * g + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) MFMA(3 * g + k);
      }
      MFMA(15);
    } else if (PAT == 12) {  // {exp x8, cvt x4, mfma x3} x 5 + 1 mfma
#pragma unroll
      for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int k = 0; k < 8; ++k) EXP(8 * g + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) CVT(k);
#pragma unroll
        for (int k = 0; k < 3; ++k) MFMA(3 * g + k);
      }
      MFMA(15);
    } else if (PAT == 13) {  // {exp x8, cvt x2, mfma x3, cvt x2} x 5 + 1 mfma
#pragma unroll
      for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int k = 0; k < 8; ++k) EXP(8 * g + k);
        CVT(0); CVT(1);
#pragma unroll
        for (int k = 0; k < 3; ++k) MFMA(3 * g + k);
        CVT(2); CVT(3);
      }
      MFMA(15);
    } else if (PAT == 14) {  // 16 mfma only
#pragma unroll
      for (int k = 0; k < 16; ++k) MFMA(k);
    } else if (PAT == 15) {  // 40 exp only
#pragma unroll
      for (int k = 0; k < 40; ++k) EXP(k);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static const char* NAMES[] = {"16 mfma | 16 exp", "{mfma, exp} x16", "16 mfma | fma | 16 exp | fma", "16 mfma | 4 fma | 16 exp | 4 fma",
                              "16 mfma | 16 exp | 16 fma", "16 mfma | 16 fma | 16 exp", "{mfma, exp, fma, fma} x16",
                              "{mfma, fma, exp, fma} x16", "16 mfma | s_nop 7 | 16 exp | s_nop 7", "{4 mfma, 4 exp} x4",
                              "{4 mfma, fma, 4 exp, fma} x4", "{8 exp, 3 mfma} x5 + mfma", "{8 exp, 4 cvt, 3 mfma} x5 + mfma",
                              "{8 exp, 2 cvt, 3 mfma, 2 cvt} x5 + mfma", "16 mfma", "40 exp"};

template <int PAT>
void run(float* out, unsigned long long* cyc) {
  const int iters = 200;
  printf("%-42s", NAMES[PAT]);
  for (int wps = 1; wps <= 4; ++wps) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kern<PAT>, dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
      (void)hipDeviceSynchronize();
    }
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %dw: %6.1f/wave %6.1f/SIMD", wps, (double)c / iters, (double)c / iters / wps);
  }
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 4);
  (void)hipMalloc(&cyc, 64);
  run<14>(out, cyc); run<15>(out, cyc);
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<5>(out, cyc); run<6>(out, cyc);
  run<7>(out, cyc); run<8>(out, cyc); run<9>(out, cyc); run<10>(out, cyc); run<11>(out, cyc); run<12>(out, cyc); run<13>(out, cyc);
  return 0;
}
