#!/usr/bin/env python
"""Per-launch table of one forward step (hipEvent brackets inside libkvq_hip.so): kernel, time,
achieved TFLOP/s and GB/s on the ALGORITHMIC flops/bytes of each launch.
    python tools/profile_step.py [--batch 4] [--dtype bf16] [--weights init] [--steps 3]   (defaults = the bench headline)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dtype", default=os.environ.get("KVQ_BENCH_DTYPE", "bf16"))
    ap.add_argument("--weights", default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--clip-tensor", action="store_true", help="feed a pre-sampled fp32 clip instead of (uint8 frames, sampler draws)")
    a = ap.parse_args()
    from kvq_amd.utils import synth
    dev = torch.device("cuda", 0)
    net, cfg, wts, hw = bench.build_net(a.dtype, dev, a.weights or ("init" if a.dtype == "bf16" else "stress"))
    if a.clip_tensor:
        x = torch.from_numpy(synth.synth_clip(1234, 32, 224, 224, batch=a.batch)).to(dev)
    else:       # the bench's step: the embedding launch reads through the fragment sampler
        x = bench.Source(a.batch, dev, 1234).fragments(0, a.batch)
    bb = net.swin_tiny_grpb_backbone
    with torch.no_grad():
        for _ in range(2):
            net(inputs={"technical": x}, reduce_scores=True)
        bb.profile(a.batch, 32, 224, 224, dev, True)
        for _ in range(a.steps):
            net(inputs={"technical": x}, reduce_scores=True)
        recs = bb.profile_read(a.batch, 32, 224, 224, dev)
    n = len(recs) // a.steps
    print(f"{'#':>3} {'kind':10} {'kernel':52} {'us':>8} {'GFLOP':>8} {'TF/s':>7} {'MB':>8} {'GB/s':>7}")
    tot = 0.0
    for i in range(n):
        ms = sum(recs[i + s * n]["ms"] for s in range(a.steps)) / a.steps
        r = recs[i]
        tot += ms
        print(f"{i:3d} {r['kind']:10} {r['kernel'][:52]:52} {ms * 1e3:8.1f} {r['flops'] / 1e9:8.2f} "
              f"{r['flops'] / ms / 1e9:7.1f} {r['bytes'] / 1e6:8.1f} {r['bytes'] / ms / 1e6:7.0f}")
    print(f"total GPU time per step: {tot:.3f} ms  ({a.batch} clips)")


if __name__ == "__main__":
    main()
