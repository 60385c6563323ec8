// Shared device/host helpers for libkvq_hip.so (gfx950 only — no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/kvq_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) int32_t i32x4;

namespace kvq {

// thread-local error message behind kvq_last_error()
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define KVQ_CHECK_HIP(expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return kvq::hip_fail(_e, #expr);    \
  } while (0)

#define KVQ_CHECK_LAUNCH(name)                                \
  do {                                                        \
    hipError_t _e = hipGetLastError();                        \
    if (_e != hipSuccess) return kvq::hip_fail(_e, name);     \
  } while (0)

#define KVQ_REQUIRE(cond, code, ...)                          \
  do {                                                        \
    if (!(cond)) {                                            \
      kvq::set_error(__VA_ARGS__);                            \
      return code;                                            \
    }                                                         \
  } while (0)

__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: round-to-nearest-even
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t u) {
  return __uint_as_float(((uint32_t)u) << 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// ---- 16-bit MFMA operand types.  KVQ_DT_BF16 / KVQ_DT_FP16 select one of these at run time; both
// feed the same-rate MFMA (v_mfma_f32_*_bf16 / _f16) and move the same bytes.  fp16's 11-bit
// mantissa is what holds the 1e-3 MOS parity gate on O(1)-logit weights;
// conversions saturate at +-65504 instead of producing inf.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

struct Bf16 {
  using v8 = bf16x8;
  static __device__ __forceinline__ uint16_t cvt(float f) { return f2bf(f); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf2(lo, hi); }
  static __device__ __forceinline__ uint32_t pack2_raw(float lo, float hi) { return pack_bf2(lo, hi); }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return bf2f(u); }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // acc + a.lo * b.lo + a.hi * b.hi on packed pairs (v_dot2c_f32_bf16)
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
};

// fp32 -> fp16 conversions must SATURATE (a value past 65504 would become inf and poison the row): instead of a
// v_med3_f32 per element, every kernel that narrows sets MODE.FP16_OVFL (bit 23) once at its start — the converter
// itself then clamps overflowed results to +/-65504 (true infinities are preserved).  ~1 VALU slot per element saved
// in every 16-bit epilogue; bf16 shares fp32's exponent range and needs nothing.
__device__ __forceinline__ void fp16_saturate_mode() {
  __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1 /* hwreg(HW_REG_MODE, 23, 1) */, 1);
}

struct Fp16 {
  using v8 = f16x8;
  static __device__ __forceinline__ float sat(float f) { return f; }      // see fp16_saturate_mode()
  static __device__ __forceinline__ uint16_t cvt(float f) {
    _Float16 h = (_Float16)sat(f);
    return __builtin_bit_cast(uint16_t, h);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2 v = {(_Float16)sat(lo), (_Float16)sat(hi)};
    return __builtin_bit_cast(uint32_t, v);
  }
  // no saturation: for values known to be in range (softmax probabilities) -> one v_cvt_pk_f16_f32
  static __device__ __forceinline__ uint32_t pack2_raw(float lo, float hi) {
    f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {      // v_dot2c_f32_f16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), acc, false);
  }
};

// exact GELU (erf), as nn.GELU() default (swin_backbone.py:72, head.py:56)
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for the fused epilogues, two values at a time.  erfc by Abramowitz-Stegun 7.1.28,
//   erfc(z) = (1 + a1 z + ... + a6 z^6)^-16,  |err| <= 3e-7  (fp32-roundoff class, ~3 orders below the 16-bit
// rounding applied to the result), and gelu(x) = max(x,0) - |x|/2 * erfc(|x|/sqrt2).  Written on float2 so that
// the polynomial, the four squarings and the tail compile to packed v_pk_fma_f32 / v_pk_mul_f32 (2 lanes-worth
// per issue slot); the only quarter-rate instruction left is ONE v_rcp_f32 per value (A&S 7.1.26, used before,
// needs rcp + exp and ~2x the issue cycles — the fused MLP launch is bound by exactly this).  Overflow of the
// 16th power for huge |x| gives rcp(inf) = 0, i.e. the exact limit.  The fp32 score head keeps libm erff.
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  f32x2 p = z * 0.0000430638f + 0.0002765672f;
  p = p * z + 0.0001520143f;
  p = p * z + 0.0092705272f;
  p = p * z + 0.0422820123f;
  p = p * z + 0.0705230784f;
  p = p * z + 1.0f;
  p = p * p;
  p = p * p;
  p = p * p;
  p = p * p;
  const f32x2 rc = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
  const f32x2 hz = z * 0.70710678118654752440f;        // |x| / 2
  const f32x2 r = x * 0.5f + hz;                       // max(x, 0)
  return r - hz * rc;
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  float p = fmaf(z, 0.0000430638f, 0.0002765672f);
  p = fmaf(p, z, 0.0001520143f);
  p = fmaf(p, z, 0.0092705272f);
  p = fmaf(p, z, 0.0422820123f);
  p = fmaf(p, z, 0.0705230784f);
  p = fmaf(p, z, 1.0f);
  p = p * p;
  p = p * p;
  p = p * p;
  p = p * p;
  const float hz = z * 0.70710678118654752440f;
  return fmaf(-hz, __builtin_amdgcn_rcpf(p), fmaf(x, 0.5f, hz));
}

// diagnostic stamp buffer (kvq_debug_gemm_trace): 8 uint64 per workgroup, NULL = off
extern unsigned long long* g_trace;
extern int g_trace_blocks;

// tile variant the GEMM dispatcher picks for a shape: (MI==NI) * 100 + BK  (gemm.hip)
int gemm_variant(int M, int N, int K);
// true: kvq_gemm_bf16 takes the 256 x 256 x 64 eight-phase kernel for this shape (gemm256.hip); profile records carry tile code 4464
bool gemm8p_wanted(int M, int N, int K);

// The opt-in to more than 64 KiB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE property of a KERNEL:
// the largest request granted so far is remembered per (kernel address, device ordinal) in one table (common.cpp) — the object itself
// carries no state, so it does not matter how many kernels share one call site (a generic lambda over kernels of one pointer type
// is a single instantiation).
struct LdsOptIn {
  int ensure(const void* kernel, int want);      // KVQ_OK, or the HIP failure
};

// shape limits of the fused fast-pathway stem (conv.hip::kvq_conv_stem_pool), pointer alignment aside
bool stem_pool_shape_ok(int B, int T, int H, int W, int kd);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace kvq
