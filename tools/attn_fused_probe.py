#!/usr/bin/env python
"""Attention launch with the fused qkv projection against qkv GEMM + plain launch, trunk geometries at 4 clips — measurement only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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

N = 392
for nW, C, ntyp in ((128, 96, 64),):
    nH, B = C // 32, 4
    BW = B * nW
    x = torch.randn(BW * N, C, device="cuda").half(); W = (torch.randn(3 * C, C, device="cuda") * 0.1).half(); b = torch.randn(3 * C, device="cuda")
    tok = torch.zeros(ntyp * N, 2, dtype=torch.int32, device="cuda")
    n = torch.arange(N, device="cuda")
    tok[:, 0] = ((n // 49) * 169 + ((n // 7) % 7) * 13 + n % 7).repeat(ntyp).int()
    rpb = torch.randn(2535, nH, device="cuda") * 0.5
    dense = kernels.attn_bias_dense(tok, rpb, None, 1267, ntyp, N, False)
    qkv = torch.empty(3, nH, BW * N, 32, device="cuda", dtype=torch.float16)
    out = torch.empty(BW * N, C, device="cuda", dtype=torch.float16)
    ug = t_of(lambda: kernels.gemm(x, W, b, _abi.EPI_QKV_BF16, num_heads=nH, q_scale=0.1767, out=qkv))
    ua = t_of(lambda: kernels.window_attention_dense(qkv, dense, nW, N, ntyp, out=out))
    uf = t_of(lambda: kernels.window_attention_dense(qkv, dense, nW, N, ntyp, out=out, x_ln=x, w_qkv=W, b_qkv=b, q_scale=0.1767))
    print(f"nW={nW} C={C}: qkv GEMM {ug:6.1f} us + attention {ua:6.1f} us = {ug+ua:6.1f} | fused {uf:6.1f} us (prologue +{uf-ua:5.1f})", flush=True)
