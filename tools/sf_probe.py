#!/usr/bin/env python
"""SlowFast-R50 motion branch alone on one video (8 clips of 3x32x224x224): ms per video + algorithmic TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd
from kvq_amd.models.backbones.slowfast_model import conv_flops, pack_pathway_output, slowfast
from kvq_amd.utils import synth
dev = "cuda:0"
sf = slowfast().to(dev).eval()
x = torch.from_numpy(synth.synth_clip(8, 32, 224, 224, batch=8)).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with torch.no_grad():
    for _ in range(3): sf.forward_clips(x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): sf.forward_clips(x)
    torch.cuda.synchronize()
dt = (time.time() - t) / n
fl = 8 * conv_flops()[0]
print(f"SlowFast-R50, 8 clips: {dt*1e3:.2f} ms per video, {fl/dt/1e12:.0f} TFLOP/s algorithmic ({fl/8e9:.1f} GFLOP per clip)")
