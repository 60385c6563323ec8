#!/usr/bin/env python
"""SlowFast-R50 per-layer times (kvq_convnet_profile: HIP events around every op of the plan), one video = 8 clips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd
from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast
from kvq_amd.utils import synth
dev = "cuda:0"
sf = slowfast().to(dev).eval()
x = torch.from_numpy(synth.synth_clip(8, 32, 224, 224, batch=8)).to(dev)
with torch.no_grad():
    for _ in range(3): sf.forward_clips(x)
    torch.cuda.synchronize()
    rows = sf.profile_layers(x)
tot = sum(r["ms"] for r in rows)
print(f"{'op':58} {'M':>7} {'N':>5} {'K':>5} {'us':>8} {'TF/s':>7}")
for r in rows:
    print(f"{(r['kind'] + ' ' + r['name'])[:58]:58} {r['M']:7d} {r['N']:5d} {r['K']:5d} {r['ms']*1e3:8.1f} {r['tflops']:7.1f}")
print(f"total {tot:.3f} ms")
