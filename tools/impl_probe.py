#!/usr/bin/env python
"""implicit-GEMM conv against the plain GEMM of the same (M, N, K) and tile — where does the conv path lose?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd
from kvq_amd import _abi, kernels


def t_of(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = "cuda:0"
N = 256
bias = torch.randn(N, device=dev)
for name, shape, kern, pad in [("1x1x1 C=3072", (8, 8, 14, 14, 3072), (1, 1, 1), (0, 0, 0)),
                               ("3x1x1 C=1024", (8, 8, 14, 14, 1024), (3, 1, 1), (1, 0, 0)),
                               ("1x3x3 C=256", (8, 8, 14, 14, 256), (1, 3, 3), (0, 1, 1))]:
    x = torch.randn(*shape, device=dev).half()
    K = kern[0] * kern[1] * kern[2] * shape[-1]
    W = (torch.randn(N, K, device=dev) * 0.05).half()
    M = shape[0] * shape[1] * shape[2] * shape[3]
    A = torch.randn(M, K, device=dev).half()
    tc = t_of(lambda: kernels.conv_implicit(x, W, bias, kern, (1, 1, 1), pad, True))
    tg = t_of(lambda: kernels.gemm(A, W, bias, _abi.EPI_BIAS_BF16))
    tr = t_of(lambda: kernels.conv_gemm(A, W, bias, True))           # the plain GEMM with the conv path's ReLU epilogue
    print(f"{name}: M={M} K={K}: implicit conv {tc:6.1f} us | plain GEMM {tg:6.1f} us | plain GEMM + ReLU epilogue {tr:6.1f} us", flush=True)
