"""slow-pathway fused stem: time against the number of clips (fixed cost vs per-item cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd  # noqa
from kvq_amd import kernels
w64 = torch.randn(64, 147, device="cuda") / 12
wimg = kernels.stem64_pack_weight(w64, torch.float16)
b64 = torch.randn(64, device="cuda")
ti = torch.linspace(0, 31, 8).long().int().cuda()
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, 3, 32, 224, 224, device="cuda")
    out = torch.empty(B, 8, 56, 56, 80, dtype=torch.float16, device="cuda")
    fn = lambda: kernels.conv_stem64_pool(x, ti, wimg, b64, True, out=out)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B:2d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  ({B * 8 * 28} items)")
