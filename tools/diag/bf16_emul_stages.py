"""Teacher-forced per-stage comparison: each stage of the HIP trunk against the fp32 emulations of its operand rounding, both started from
the HIP path's own stage input (test-side diagnostic, GPU box)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import kvq_amd  # noqa
from kvq_amd.utils import synth
from oracle import swin3d_oracle as O
sys.path.insert(0, "tests")
from test_gpu_e2e import build_network  # noqa

cfg = synth.SWIN_T_GRPB
shift = tuple(w // 2 for w in cfg.window)


def stage(y, p, i, q, ko):
    for b in range(cfg.depths[i]):
        y = O.swin_block(y, p, f"layers.{i}.blocks.{b}.", cfg.num_heads[i], cfg.window, (0, 0, 0) if b % 2 == 0 else shift, q, ko)
    if i < len(cfg.depths) - 1:
        y = (O.patch_merge_kernel_order if ko and y.shape[-1] <= 192 else O.patch_merge)(y, p, f"layers.{i}.downsample.", q)
    return y


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


for dtype, odt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    for wseed, cseed, B, T, H, W in [(1, 2, 2, 8, 80, 80), (3, 4, 1, 16, 64, 64), (9, 10, 1, 32, 96, 96)]:
        net, key = build_network("SWIN_T_GRPB", wseed, "stress", dtype)
        bb = getattr(net, key + "_backbone")
        x = torch.from_numpy(synth.synth_clip(cseed, T, H, W, batch=B))
        p = {k: torch.from_numpy(v).float() for k, v in synth.synth_swin_weights(cfg, wseed, "stress").items()}
        q = O.operand_rounding(odt)
        with torch.no_grad():
            taps = [bb({"technical": x.to("cuda:0")}, layer=i).cpu().permute(0, 2, 3, 4, 1).contiguous() for i in range(5)]
            e = O.patch_embed(x, p, cfg.patch, q)
            print(dtype, (wseed, cseed, B, T, H, W), "embed rel %.2e (exact %.2e)" % (rel(taps[0], e), rel(taps[0], O.patch_embed(x, p, cfg.patch))), flush=True)
            for i in range(4):
                r0 = stage(taps[i], p, i, O._ident, False)
                e0 = stage(taps[i], p, i, q, False)
                e1 = stage(taps[i], p, i, q, True)
                print("   stage %d: HIP vs exact %.2e | vs emu %.2e | vs emu(kernel order) %.2e | emu(ko) vs exact %.2e" % (
                    i, rel(taps[i + 1], r0), rel(taps[i + 1], e0), rel(taps[i + 1], e1), rel(e1, r0)), flush=True)
