// Issue-rate bench of the streaming attention q-block body (csrc/attn32.hip::a32_qblock): ONE workgroup on one CU, W waves per SIMD,
// every wave runs `iters` q-blocks against an LDS-resident K | V slot and an L1/L2-resident bias tile row (no HBM, no flags, no
// tickets).  Reports SIMD cycles per 32 x 32 score block = slowest wave's cycles / (iters * 13) / W.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef A32_LOOP_SHORT
#define A32_LOOP_SHORT 0          // 1: the N = 385..392 form of the 13th key block (round 6)
#endif
#include "attn32.hip"

namespace kvq {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
int hip_fail(hipError_t e, const char* w) { fprintf(stderr, "HIP %s: %s\n", w, hipGetErrorString(e)); return -1; }
int LdsOptIn::ensure(const void* kernel, int want) { return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess ? 0 : -1; }
bool stem_pool_shape_ok(int, int, int, int, int) { return false; }
bool latency_mode() { return false; }
unsigned long long* g_trace = nullptr;
int g_trace_blocks = 0;

template <int W>
__global__ __launch_bounds__(256 * W) void loop_kernel(Attn32Params p, int iters, int do_store, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using V8 = Fp16::v8;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < (A32_SLOT + 1024) / 4; i += 256 * W) reinterpret_cast<uint32_t*>(smem)[i] = 0x2c002e00u + (uint32_t)(i & 0xff) * 0x00010001u;     // small fp16 values
  __syncthreads();
  const V8 qf0 = *reinterpret_cast<const V8*>(p.qkv + lane * 8), qf1 = *reinterpret_cast<const V8*>(p.qkv + 512 + lane * 8);
  const u32x4* bd = p.image + lane;
  const u32x4 pre[2] = {bd[0], bd[64]};
  uint16_t* orow = p.out + (size_t)tid * 64;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) a32_qblock<Fp16, 0, A32_KB, A32_LOOP_SHORT != 0>(smem, A32_SLOT >> 4, bd, qf0, qf1, pre, orow, do_store != 0);
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[tid >> 6] = t1 - t0;
}
}  // namespace kvq

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int W>
static void run(const kvq::Attn32Params& p, unsigned long long* dcyc, int iters) {
  auto kern = kvq::loop_kernel<W>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kvq::A32_SLOT + 1024));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(256 * W), kvq::A32_SLOT + 1024, 0, p, iters, 0, dcyc);
    CK(hipDeviceSynchronize());
  }
  unsigned long long c[16];
  CK(hipMemcpy(c, dcyc, 8 * 4 * W, hipMemcpyDeviceToHost));
  double mx = 0, mn = 1e30;
  for (int i = 0; i < 4 * W; ++i) { mx = c[i] > mx ? c[i] : mx; mn = c[i] < mn ? c[i] : mn; }
  printf("  %dw: %6.1f (min %6.1f)", W, mx / iters / kvq::A32_KB / W, mn / iters / kvq::A32_KB / W);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  uint16_t *dq, *dimg, *dout; unsigned long long* dcyc;
  CK(hipMalloc(&dq, 4096)); CK(hipMalloc(&dimg, kvq::A32_KB * 2048 + 4096)); CK(hipMalloc(&dout, 1 << 20)); CK(hipMalloc(&dcyc, 256));
  std::vector<uint16_t> hq(2048), hi(kvq::A32_KB * 1024);
  for (size_t i = 0; i < hq.size(); ++i) { _Float16 h = (_Float16)(0.01f * (float)(i % 37) - 0.2f); __builtin_memcpy(&hq[i], &h, 2); }
  for (size_t i = 0; i < hi.size(); ++i) { _Float16 h = (_Float16)(-0.05f * (float)(i % 61)); __builtin_memcpy(&hi[i], &h, 2); }
  CK(hipMemcpy(dq, hq.data(), 4096, hipMemcpyHostToDevice));
  CK(hipMemcpy(dimg, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
  kvq::Attn32Params p{};
  p.qkv = dq; p.image = (const u32x4*)dimg; p.out = dout; p.N = 392; p.nH = 3;
  printf("cycles per 32x32 block per SIMD (abl %d):", A32_ABL);
  run<1>(p, dcyc, iters); run<2>(p, dcyc, iters); run<3>(p, dcyc, iters); run<4>(p, dcyc, iters);
  printf("\n");
  return 0;
}
