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

// GELU for the fused epilogues (round 5 form).  gelu(x) = max(x, 0) - |x| * Q(-|x|), Q the normal tail probability 0.5 erfc(a / sqrt 2).
// log2 Q(-a) is smooth and nearly quadratic (-1 - 1.15 a - 0.46 a^2 ...), so  |x| * Q  =  a * exp2(P5(a))  with a degree-5 polynomial
// fitted (minimax in the ABSOLUTE error of a * exp2(P), a in [0, 9]) to |err| <= 8.5e-7 over all x — below the 1.4e-6 of the form
// used before (A&S 7.1.28: erfc = (1 + a1 z + ... + a6 z^6)^-16, 14 full-rate VALU + one v_rcp_f32 per value) and three orders below the
// 16-bit rounding applied to the result.  Cost per value: |x|, 5 v_fma, one v_exp_f32, v_max, v_fma = 8 full-rate instructions + one
// quarter-rate (about 11 issue slots against 17).  The leading coefficient is negative: P5 -> -inf, the tail term underflows to exactly 0
// for large |x|, where gelu = max(x, 0).  The fused launches (tails, MLP epilogues) are bound by exactly this VALU work.
// The fp32 score head keeps libm erff.
__device__ __forceinline__ float gelu_fast(float x) {
  const float a = fabsf(x);
  float p = fmaf(-4.7330835272e-04f, a, 7.0845445981e-03f);
  p = fmaf(p, a, -5.1827334402e-02f);
  p = fmaf(p, a, -4.5999251338e-01f);
  p = fmaf(p, a, -1.1507878060e+00f);
  p = fmaf(p, a, -1.0000376313e+00f);
  return fmaf(-a, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}
// two values: plain scalar code (a packed v_pk_fma_f32 occupies the VALU for two plain instructions on gfx950: nothing to gain)
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) { return f32x2{gelu_fast(x[0]), gelu_fast(x[1])}; }

// GELU of two values straight into the packed 16-bit MFMA operand of the fused tails (fc1 accumulator -> fc2 operand).
// bf16: the fp32 form, then one pack.  fp16 (round 5, KVQ_GELU_PK16=1): the SAME polynomial evaluated on the packed pair —
// v_cvt_pk_f16_f32, v_and (|x|), 5 v_pk_fma_f16, 2 v_exp_f16 (the high half by SDWA), v_pack_b32_f16, v_pk_max_f16,
// v_pk_fma_f16 = 10 full-rate + 2 quarter-rate instructions per PAIR against 17 + 2: the result is a 16-bit operand either way,
// the evaluation adds the rounding of x and of the Horner steps (rms error of the operand x 1.6-1.8 at |x| <= 1, level past
// |x| ~ 4: tools/diag/gelu_pk16_error.py); MODE.FP16_OVFL (fp16_saturate_mode) clamps every overflow on the way — P5 -> -65504,
// exp2 -> 0 — so no NaN can form.  -DKVQ_GELU_PK16=0 builds the fp32 evaluation.
#ifndef KVQ_GELU_PK16
#define KVQ_GELU_PK16 0
#endif
template <class E>
__device__ __forceinline__ uint32_t gelu_pack2(float lo, float hi) {
  return E::pack2(gelu_fast(lo), gelu_fast(hi));
}
#if KVQ_GELU_PK16
template <>
__device__ __forceinline__ uint32_t gelu_pack2<Fp16>(float lo, float hi) {
  const f16x2 x = {(_Float16)lo, (_Float16)hi};                                  // one v_cvt_pk_f16_f32 (saturating mode)
  const f16x2 a = __builtin_bit_cast(f16x2, __builtin_bit_cast(uint32_t, x) & 0x7fff7fffu);
  auto c = [](float v) { return f16x2{(_Float16)v, (_Float16)v}; };
  f16x2 p = __builtin_elementwise_fma(c(-4.7330835272e-04f), a, c(7.0845445981e-03f));
  p = __builtin_elementwise_fma(p, a, c(-5.1827334402e-02f));
  p = __builtin_elementwise_fma(p, a, c(-4.5999251338e-01f));
  p = __builtin_elementwise_fma(p, a, c(-1.1507878060e+00f));
  p = __builtin_elementwise_fma(p, a, c(-1.0f));
  const f16x2 e2 = __builtin_elementwise_exp2(p);      // two v_exp_f16 (the high half by SDWA) + v_pack_b32_f16; the compiler places
                                                       // gfx950's trans-result wait states (hand-written SDWA into one register read stale data)
  const f16x2 m = __builtin_elementwise_max(x, f16x2{(_Float16)0.f, (_Float16)0.f});
  const f16x2 r = __builtin_elementwise_fma(-a, e2, m);
  return __builtin_bit_cast(uint32_t, r);
}
#endif

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

// KVQ_LATENCY=1: the launch geometries that are fastest for ONE step alone on the chip (q-split attention at the late stages, the
// stage-3 GEMM chain, the C = 384 tail at one 512-VGPR workgroup per CU) instead of the defaults, which minimise CU x time for several
// steps in flight on HIP streams (one stream 279 -> 245 videos/s, four lanes 356 -> 407).  Results are identical up to the
// rounding points that differ between a fused tail and the GEMM chain; each choice also has its own knob.
bool latency_mode();

// csrc/ln.hip: kvq_layernorm_rows on an fp32 (x_f16 = 0) or fp16 residual stream
int layernorm_rows_stream(const float* x, int x_f16, const int32_t* map, int nparts, int n_batch, int rows_in, int rows_out, int Cin,
                          const float* gamma, const float* beta, float eps, uint16_t* out_h, int dtype, float* out_f32, void* stream);

// shape limits of the fused fast-pathway stem (conv.hip::kvq_conv_stem_pool), pointer alignment aside
bool stem_pool_shape_ok(int B, int T, int H, int W, int kd);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace kvq
