#!/usr/bin/env python
"""The trunk's stage-3 GEMM shapes with L2-warm operands (the same launch repeated) against HBM-cold ones (weights and
activations rotated over enough copies to cycle 400 MB through, as a C2 step does between two uses of a weight) — measurement only.
    python tools/gemm_cold.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import kvq_amd  # noqa: E402,F401
from kvq_amd import _abi, kernels  # noqa: E402

SHAPES = [(1568, 768, 3072, "fc2 stage 3"), (1568, 3072, 768, "fc1 stage 3"), (1568, 2304, 768, "qkv stage 3"), (1568, 768, 768, "proj stage 3"),
          (12544, 1152, 384, "qkv stage 2"), (1568, 768, 1536, "merge 2->3")]


def timed(fns, n):
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    dev = "cuda:0"
    for M, N, K, name in SHAPES:
        per = (M * K + N * K) * 2 + M * N * 2
        ncopy = max(2, int(400e6 // per) + 1)
        As = [torch.randn(M, K, device=dev).half() for _ in range(ncopy)]
        Ws = [(torch.randn(N, K, device=dev) * 0.1).half() for _ in range(ncopy)]
        b = torch.randn(N, device=dev)
        outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(ncopy)]
        warm = timed([lambda: kernels.gemm(As[0], Ws[0], b, _abi.EPI_BIAS_BF16, out=outs[0])], 64)
        cold_w = timed([(lambda i=i: kernels.gemm(As[0], Ws[i], b, _abi.EPI_BIAS_BF16, out=outs[0])) for i in range(ncopy)], 64)
        cold = timed([(lambda i=i: kernels.gemm(As[i], Ws[i], b, _abi.EPI_BIAS_BF16, out=outs[i])) for i in range(ncopy)], 64)
        fl = 2.0 * M * N * K
        print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: warm {warm:6.1f} us ({fl / warm / 1e6:6.1f} TF/s) | cold weights {cold_w:6.1f} us | "
              f"cold weights + activations {cold:6.1f} us ({fl / cold / 1e6:6.1f} TF/s)   [{ncopy} copies]", flush=True)
