#!/usr/bin/env python
"""qkv GEMM of the trunk's stages at 4 clips (head-major epilogue) next to the plain row-major epilogue — measurement only."""
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

for M, C in ((200704, 96), (50176, 192), (12544, 384), (3136, 768)):
    nH = C // 32
    A = torch.randn(M, C, device="cuda").half(); W = (torch.randn(3 * C, C, device="cuda") * 0.1).half(); b = torch.randn(3 * C, device="cuda")
    out = torch.empty(3, nH, M, 32, device="cuda", dtype=torch.float16)
    uq = t_of(lambda: kernels.gemm(A, W, b, _abi.EPI_QKV_BF16, num_heads=nH, q_scale=0.1767, out=out))
    ub = t_of(lambda: kernels.gemm(A, W, b, _abi.EPI_BIAS_BF16))
    byts = 2.0 * (M * C + 3 * C * C + M * 3 * C)
    print(f"M={M:6d} C={C:4d}: qkv epilogue {uq:7.1f} us ({byts/uq/1e6:5.2f} TB/s, {2.0*M*3*C*C/uq/1e6:6.1f} TF/s) | row-major {ub:7.1f} us", flush=True)
