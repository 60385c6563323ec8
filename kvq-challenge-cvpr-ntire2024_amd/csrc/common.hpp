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
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

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
// exact GELU (erf), as nn.GELU() default (swin_backbone.py:72, head.py:56)
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace kvq
