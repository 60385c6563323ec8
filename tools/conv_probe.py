#!/usr/bin/env python
"""Where do the conv nets spend their time?  SimpleVQA ResNet-50 (8 frames of 448x448) and SlowFast-R50 (8 clips):
total forward time and the share of the im2col gathers / GEMMs (hipEvents around the wrapped calls, serial)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd
from kvq_amd import kernels
from kvq_amd.models.backbones.simpleVQA_model import resnet50
from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast

dev = "cuda:0"
acc = {}
def wrap(name):
    f = getattr(kernels, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record()
        acc.setdefault(name, []).append((e0, e1))
        return r
    setattr(kernels, name, g)
for n in ("im2col_nd", "conv_gemm", "conv_implicit", "gemm", "pool_nd", "mean_std_pool", "conv_stem_direct"):
    wrap(n)

def run(label, fn, flops):
    for _ in range(2): fn()
    torch.cuda.synchronize(); acc.clear()
    t = time.time(); fn(); torch.cuda.synchronize(); dt = (time.time() - t) * 1e3
    parts = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in acc.items()}
    print(f"{label}: {dt:.2f} ms ({flops / dt / 1e9:.0f} TFLOP/s)  " + "  ".join(f"{k} {v:.2f} ms x{len(acc[k])}" for k, v in parts.items()))

with torch.no_grad():
    net = resnet50(pretrained=False).to(dev).eval()
    x = torch.randn(1, 3, 8, 448, 448, device=dev)
    feat = torch.randn(1, 8, 2304, device=dev)
    run("SimpleVQA ResNet-50, 8 x 448x448", lambda: net({"simpleVQA": x, "feat": feat}), 261.6e9)
    sf = slowfast().to(dev).eval()
    clips = torch.randn(8, 3, 32, 224, 224, device=dev)
    run("SlowFast-R50, 8 clips", lambda: sf(pack_pathway_output(clips)), 8 * 2 * 50e9)
