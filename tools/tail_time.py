#!/usr/bin/env python
"""Time the fused post-attention launch (kvq_block_tail) at a stage's size: tools/tail_time.py [M] [C] (default 12544 384 = stage 2
of Swin-T at 4 clips).  KVQ_TAILMM=0 selects csrc/tail16.hip for C = 384."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kvq_amd
from kvq_amd import kernels

M = int(sys.argv[1]) if len(sys.argv) > 1 else 12544
C = int(sys.argv[2]) if len(sys.argv) > 2 else 384
hid, dev = 4 * C, "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
A, x = r(M, C).half(), r(M, C, sc=2.0)
Wp, W1, W2 = r(C, C, sc=0.1).half(), r(hid, C, sc=0.1).half(), r(C, hid, sc=0.05).half()
bp, b1, b2, g2, b2n = r(C), r(hid), r(C), 1 + 0.1 * r(C), 0.1 * r(C)
pack = kernels.block_tail_pack(Wp, bp, g2, b2n, W1, b1, W2, b2)
perm = torch.randperm(M, device=dev, generator=g).int()
kw = dict(next_norm=(1 + 0.1 * r(C), 0.1 * r(C)), next_dst=perm, next_rows=M)
for emit in (True, False):
    k = kw if emit else {}
    for _ in range(3): kernels.block_tail(A, x.clone(), pack, hid, **k)
    xs = [x.clone() for _ in range(20)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in xs: kernels.block_tail(A, t, pack, hid, **k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * M * C * C + 4.0 * M * C * hid
    print(f"block_tail M={M} C={C} emit={emit} TAILMM={os.environ.get('KVQ_TAILMM', '1')}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s")
