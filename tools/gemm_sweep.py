#!/usr/bin/env python
"""GEMM rate of kvq_gemm on the trunk's shapes next to torch.matmul (hipBLASLt) as a yardstick — measurement only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd
from kvq_amd import _abi, kernels

SHAPES = [(8192, 8192, 8192), (56448, 1536, 512), (56448, 512, 512), (32768, 2048, 512), (32768, 512, 2048),
          (12544, 256, 3072), (12544, 1152, 384), (12544, 1536, 384), (12544, 384, 1536), (200704, 288, 96), (1568, 3072, 768), (1568, 768, 3072), (3136, 3072, 768), (3136, 768, 3072), (3136, 2304, 768),
          (131072, 768, 256), (32768, 1536, 512), (8192, 3072, 1024), (8192, 4096, 1024), (8192, 1024, 4096)]


def t_of(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    dev = "cuda:0"
    shapes = SHAPES
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.1).half(); b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        kernels.gemm_tile_mode(0)
        us = t_of(lambda: kernels.gemm(A, W, b, _abi.EPI_BIAS_BF16))
        kernels.gemm_tile_mode(1)
        u8 = t_of(lambda: kernels.gemm(A, W, b, _abi.EPI_BIAS_BF16)) if K % 64 == 0 else float("nan")
        kernels.gemm_tile_mode(-1)
        ua = t_of(lambda: kernels.gemm(A, W, b, _abi.EPI_BIAS_BF16))
        ut = t_of(lambda: torch.matmul(A, W.t(), out=out))
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N:5d} K={K:5d}: ring128 {us:8.1f} us {fl/us/1e6:7.1f} TF/s | 8phase256 {u8:8.1f} us {fl/u8/1e6:7.1f} TF/s | "
              f"by-shape {ua:8.1f} us {fl/ua/1e6:7.1f} | hipBLASLt {ut:8.1f} us {fl/ut/1e6:7.1f} TF/s", flush=True)
