// Standalone driver for the 256x256x64 eight-phase main loop (csrc/gemm8p.hpp): correctness on sampled outputs against a
// host fp64 sum, then the rate over repeated launches.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../kvq-.../csrc
//   ./gemm8p_bench M N K [iters]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include "gemm8p.hpp"

namespace kvq { void set_error(const char*, ...) {} int hip_fail(hipError_t, const char*) { return -1; } }
using namespace kvq;

struct P { const uint16_t* A; const uint16_t* W; const float* bias; uint16_t* out; int M, N, K; };

template <typename E>
__global__ __launch_bounds__(512, 2) void k8p(P p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int nbm = (p.M + 255) / 256, nbn = (p.N + 255) / 256;
  int bm, bn;
  g8::tile_of(g8::logical_block(), nbm, nbn, bm, bn);
  const int m0 = bm * 256, n0 = bn * 256;
  g8::StageGeom sg; sg.init();
  g8::PlainSrc sa, sb;
  sa.init(p.A + (size_t)m0 * p.K, p.M - m0, p.K, sg.a_row, sg.lc, 0);
  sb.init(p.W + (size_t)n0 * p.K, p.N - n0, p.K, sg.b_row, sg.lc, 0);
  f32x16 acc[2][2][2];
  g8::mainloop<E>(lds, sa, sb, p.K / 64, acc);
  // epilogue: per wave a 32 x 64 fp32 slab, lane = 8 consecutive columns of a row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
  float* slab = reinterpret_cast<float*>(lds) + wave * 32 * 64;
  const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
  const int ch = lane & 7, rsub = lane >> 3;
  const int n = n0 + wc * 64 + ch * 8;
  float bias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias[k] = (p.bias && n + k < p.N) ? p.bias[n + k] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + row_hi) * 64 + j * 32 + col_in] = acc[i][mi][j][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rl = it * 8 + rsub;
        const int m = m0 + wr * 128 + i * 64 + mi * 32 + rl;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(slab + rl * 64 + ch * 8), s1 = *reinterpret_cast<const f32x4*>(slab + rl * 64 + ch * 8 + 4);
        if (m < p.M && n < p.N) {
          const u32x4 o = {E::pack2(s0[0] + bias[0], s0[1] + bias[1]), E::pack2(s0[2] + bias[2], s0[3] + bias[3]),
                           E::pack2(s1[0] + bias[4], s1[1] + bias[5]), E::pack2(s1[2] + bias[6], s1[3] + bias[7])};
          *reinterpret_cast<u32x4*>(p.out + (size_t)m * p.N + n) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; __builtin_memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; __builtin_memcpy(&h, &u, 2); return (float)h; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
  int iters = argc > 4 ? atoi(argv[4]) : 20;
  if (K % 64 || N % 8) { printf("need K %% 64 == 0, N %% 8 == 0\n"); return 1; }
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
  std::vector<float> hb(N);
  for (auto& v : hA) v = f2h(U(rng));
  for (auto& v : hW) v = f2h(U(rng) * 0.25f);
  for (auto& v : hb) v = U(rng);
  uint16_t *dA, *dW, *dO; float* db;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dO, (size_t)M * N * 2)); CK(hipMalloc(&db, N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dO, 0xff, (size_t)M * N * 2));
  auto kern = k8p<Fp16>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, g8::LDS_BYTES));
  P p{dA, dW, db, dO, M, N, K};
  dim3 grid(((M + 255) / 256) * ((N + 255) / 256)), block(512);
  hipLaunchKernelGGL(kern, grid, block, g8::LDS_BYTES, 0, p);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> hO((size_t)M * N);
  CK(hipMemcpy(hO.data(), dO, hO.size() * 2, hipMemcpyDeviceToHost));
  // sampled check: every tile corner region + random positions
  double maxerr = 0; long bad = 0, checked = 0;
  std::uniform_int_distribution<int> Um(0, M - 1), Un(0, N - 1);
  for (int s = 0; s < 6000; ++s) {
    int m = Um(rng), n = Un(rng);
    if (s < 64) { m = (s & 7) * (M - 1) / 7; n = (s >> 3) * (N - 1) / 7; }
    double ref = hb[n];
    for (int k = 0; k < K; ++k) ref += (double)h2f(hA[(size_t)m * K + k]) * h2f(hW[(size_t)n * K + k]);
    const double got = h2f(hO[(size_t)m * N + n]);
    const double err = fabs(got - ref), tol = 2e-3 * fabs(ref) + 2e-2 + 1e-3 * sqrt((double)K) * 0.02;
    if (!(err <= tol)) { if (bad < 5) printf("  mismatch (%d,%d): got %f ref %f\n", m, n, got, ref); ++bad; }
    if (err > maxerr) maxerr = err;
    ++checked;
  }
  printf("check: %ld/%ld bad, max |err| %.4g\n", bad, checked, maxerr);
  // race screen: repeat and compare bitwise with the first result
  long diff = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(dO, 0xff, (size_t)M * N * 2));
    hipLaunchKernelGGL(kern, grid, block, g8::LDS_BYTES, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> h2((size_t)M * N);
    CK(hipMemcpy(h2.data(), dO, h2.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h2.size(); ++i) diff += h2[i] != hO[i];
  }
  printf("race screen: %ld differing elements over 3 repeats\n", diff);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, block, g8::LDS_BYTES, 0, p);
  float best = 1e30f, tot = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, block, g8::LDS_BYTES, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    best = ms < best ? ms : best; tot += ms;
  }
  const double fl = 2.0 * M * N * K;
  printf("M=%d N=%d K=%d: %.1f us mean, %.1f us best -> %.1f TF/s mean, %.1f best (%d tiles)\n", M, N, K, tot / 5 * 1e3, best * 1e3,
         fl / (tot / 5 * 1e-3) / 1e12, fl / (best * 1e-3) / 1e12, (int)grid.x);
  return bad || diff;
}
