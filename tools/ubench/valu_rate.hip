// Micro-benchmark: issue cost (shader cycles per wave-instruction, per SIMD) of the VALU instructions the window
// attention's softmax is made of, at 1..4 waves per SIMD — which of them are quarter-rate, which pack two values.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip ; run: /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP 16
#define OPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// each OP: 8 independent chains (v0..v7 are float regs; pairs for packed ops)
template <int OP>
__global__ void kern(float* out, int iters, unsigned long long* cyc) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  float c1 = 1.0001f, c2 = 0.5f;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = (f2){v[2 * i], v[2 * i + 1]};
  f2 pc1 = {1.0001f, 1.0002f}, pc2 = {0.5f, 0.25f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 2) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
        if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc1), "v"(pc2));
        if (OP == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
        if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        if (OP == 6) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
        if (OP == 7) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
        if (OP == 8) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
        if (OP == 9) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]" : "+v"(v[i]) : "v"(c1), "v"(c2));
        if (OP == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));
        if (OP == 11) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 12) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
        if (OP == 13) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
        if (OP == 14) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
        if (OP == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc2));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// mixed: NM 16x16x32 MFMAs (independent accumulators) + NV v_exp + NF v_fma per iteration, several waves per SIMD:
// does VALU work of one wave hide under the MFMAs of another?
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NM, int NE, int NF>
__global__ void mixed(float* out, int iters, unsigned long long* cyc) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * e); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
  float c1 = 1.0001f, c2 = 0.5f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NM; ++k) acc[k % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k % 8], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NE; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k % 8]));
#pragma unroll
    for (int k = 0; k < NF; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k % 8]) : "v"(c1), "v"(c2));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static const char* NAMES[] = {"v_fma_f32", "v_exp_f32", "v_cvt_f32_f16", "v_pk_fma_f32", "v_max_f32", "v_max3_f32",
                              "v_cvt_pkrtz_f16_f32", "v_pk_add_f16", "v_exp_f16", "v_fma_mix_f32", "v_pk_mul_f32", "v_rcp_f32",
                              "v_pk_max_f16", "v_cvt_f16_f32", "v_sub_f32", "v_pk_add_f32"};

template <int OP>
void run(float* out, unsigned long long* cyc) {
  const int iters = 200;
  printf("%-22s", NAMES[OP]);
  for (int wps = 1; wps <= 4; ++wps) {
    // one workgroup of wps*4 waves on one CU -> wps waves per SIMD
    hipLaunchKernelGGL(kern<OP>, dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern<OP>, dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)c / ((double)iters * REP * 8);   // cycles per instruction of ONE wave
    printf("  %dw/SIMD: %6.2f cyc/inst/wave = %6.2f per SIMD", wps, per, per / wps);
  }
  printf("\n");
}

template <int NM, int NE, int NF>
void runm(float* out, unsigned long long* cyc) {
  const int iters = 200;
  printf("mixed %2d mfma16 + %2d exp + %2d fma:", NM, NE, NF);
  for (int wps = 1; wps <= 4; ++wps) {
    hipLaunchKernelGGL((mixed<NM, NE, NF>), dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((mixed<NM, NE, NF>), dim3(1), dim3(256 * wps), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %dw: %7.1f cyc/iter/wave = %7.1f per SIMD", wps, (double)c / iters, (double)c / iters / wps);
  }
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 4096 * 4);
  hipMalloc(&cyc, 64);
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<5>(out, cyc);
  run<6>(out, cyc); run<7>(out, cyc); run<8>(out, cyc); run<9>(out, cyc); run<10>(out, cyc); run<11>(out, cyc);
  run<12>(out, cyc); run<13>(out, cyc); run<14>(out, cyc); run<15>(out, cyc);
  runm<16, 0, 0>(out, cyc);
  runm<0, 16, 0>(out, cyc);
  runm<0, 0, 48>(out, cyc);
  runm<16, 16, 0>(out, cyc);
  runm<16, 0, 48>(out, cyc);
  runm<16, 16, 48>(out, cyc);
  runm<64, 104, 326>(out, cyc);   // the attention q-tile's mix (no LDS / memory)
  return 0;
}
