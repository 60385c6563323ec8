"""Error of the fused tails' GELU operand (csrc/common.hpp gelu_pack2): the fp32 evaluation rounded once to fp16 against the
packed-fp16 evaluation (x rounded first, five fp16 Horner steps, v_exp_f16, one fp16 fma), both against erf-GELU in float64.
CPU only:  python tools/diag/gelu_pk16_error.py"""
import numpy as np
from scipy.special import erf

C = [-4.7330835272e-04, 7.0845445981e-03, -5.1827334402e-02, -4.5999251338e-01, -1.1507878060e+00, -1.0000376313e+00]


def r16(x):
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


def gelu_ref(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def gelu_fp32_then_round(x):
    x = np.float32(x)
    a = np.abs(x)
    p = np.float32(C[0]) * a + np.float32(C[1])
    for c in C[2:]:
        p = p * a + np.float32(c)
    return r16(np.maximum(x, 0) - a * np.exp2(p.astype(np.float64)))


def gelu_packed_fp16(x):
    xh = np.clip(r16(np.float32(x)), -65504, 65504)
    a = np.abs(xh)
    c = [r16(v) for v in C[:5]] + [-1.0]
    p = r16(c[0] * a + c[1])
    for k in c[2:]:
        p = np.clip(r16(p * a + k), -65504, 65504)
    e = r16(np.exp2(p))
    return r16(-a * e + np.maximum(xh, 0))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for sig in (0.25, 0.5, 1, 2, 4):
        x = rng.normal(0, sig, 2_000_000)
        ref = gelu_ref(x)
        row = []
        for f in (gelu_fp32_then_round, gelu_packed_fp16):
            d = f(x) - ref
            row.append((np.sqrt((d ** 2).mean()), np.abs(d).max()))
        print(f"x ~ N(0, {sig}^2): fp32-then-round rms {row[0][0]:.3e} max {row[0][1]:.3e} | packed fp16 rms {row[1][0]:.3e} max {row[1][1]:.3e}"
              f" | rms ratio {row[1][0] / row[0][0]:.2f}")
