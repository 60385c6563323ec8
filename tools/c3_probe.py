#!/usr/bin/env python
"""BASELINE config 3 probe: Swin3D-T(GRPB) trunk + head and the SlowFast-R50 motion branch on the same 8 clips."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd
from kvq_amd.models import VQA_Network
from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast
from kvq_amd.utils import synth

dev = "cuda:0"
net = VQA_Network({"model": {"args": {"swin_tiny_grpb": {"backbone": {}, "head": {"in_channels": 768, "hidden_channels": 64}}}}}).to(dev).eval()
sf = slowfast().to(dev).eval()
x = torch.from_numpy(synth.synth_clip(8, 32, 224, 224, batch=8)).to(dev)

def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3

with torch.no_grad():
    t_swin = timed(lambda: net(inputs={"technical": x}, reduce_scores=True))
    t_sf = timed(lambda: sf.forward_clips(x))
print(f"C3, one video = 8 clips of 3x32x224x224: Swin3D-T+head {t_swin:.2f} ms, SlowFast-R50 {t_sf:.2f} ms -> "
      f"{1e3 / (t_swin + t_sf):.1f} videos/s with both branches on one GPU")

# both branches of consecutive videos on their own HIP streams (independent work: the launches fill each other's gaps)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def both(n):
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    for _ in range(n):
        with torch.cuda.stream(s1): net(inputs={"technical": x}, reduce_scores=True)
        with torch.cuda.stream(s2): sf.forward_clips(x)
    main.wait_stream(s1); main.wait_stream(s2)
with torch.no_grad():
    both(2); torch.cuda.synchronize(); t = time.time(); both(10); torch.cuda.synchronize()
    dt = (time.time() - t) / 10 * 1e3
print(f"    the two branches on two streams: {dt:.2f} ms per video -> {1e3 / dt:.1f} videos/s")
