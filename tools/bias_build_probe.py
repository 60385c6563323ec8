#!/usr/bin/env python
"""First-forward cost of a new clip geometry: the dense attention-bias images of every block (kvq_swin3d_bias_dense_build) — measurement only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd
from kvq_amd.models.backbones.swin_backbone import SwinTransformer3D
from kvq_amd.utils import synth

for name, kw, shape in (("Swin-T 32x224x224 (C2)", {}, (4, 3, 32, 224, 224)),
                        ("Swin-B 64x256x256 (C5)", dict(embed_dim=128, depths=list(synth.SWIN_B_GRPB.depths), num_heads=list(synth.SWIN_B_GRPB.num_heads)), (4, 3, 64, 256, 256))):
    bb = SwinTransformer3D(**kw).to("cuda:0").eval()
    x = torch.randn(*shape, device="cuda:0")
    with torch.no_grad():
        bb({"technical": x[:, :, :8, :64, :64].contiguous()})          # weights converted / packed on another geometry first
        torch.cuda.synchronize()
        bb._dense.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        B, _, T, H, W = shape
        handle, *_ = bb._plan(B, T, H, W, x.device)
        torch.cuda.synchronize()
        bb._dense.clear()
        e0.record(); t0 = time.perf_counter()
        bb._set_dense_bias(handle, (T, H, W), x.device, B)
        e1.record(); torch.cuda.synchronize()
        gib = sum(b.numel() for b in bb._dense[list(bb._dense)[-1]] if b is not None) / 2 ** 30
        print(f"{name}: {gib:.2f} GiB of bias images built in {e0.elapsed_time(e1):.2f} ms GPU ({(time.perf_counter()-t0)*1e3:.1f} ms wall incl. allocation)", flush=True)
