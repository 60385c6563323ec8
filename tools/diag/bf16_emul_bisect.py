"""Which part of a stage the emulation does not follow: attention / MLP switched off by zeroed weights (test-side diagnostic, GPU box)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import kvq_amd  # noqa
from kvq_amd import _abi
from kvq_amd.models import VQA_Network
from kvq_amd.utils import synth
from oracle import swin3d_oracle as O

cfg = synth.SWIN_T_GRPB
shift = tuple(w // 2 for w in cfg.window)
dtype, odt = sys.argv[1] if len(sys.argv) > 1 else "bf16", None
odt = torch.bfloat16 if dtype == "bf16" else torch.float16


def stage(y, p, i, q, ko, merge=True):
    for b in range(cfg.depths[i]):
        y = O.swin_block(y, p, f"layers.{i}.blocks.{b}.", cfg.num_heads[i], cfg.window, (0, 0, 0) if b % 2 == 0 else shift, q, ko)
    if merge and i < len(cfg.depths) - 1:
        y = (O.patch_merge_kernel_order if ko and y.shape[-1] <= 192 else O.patch_merge)(y, p, f"layers.{i}.downsample.", q)
    return y


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


wseed, cseed, B, T, H, W = 3, 4, 1, 16, 64, 64
x = torch.from_numpy(synth.synth_clip(cseed, T, H, W, batch=B))
for variant in ("full", "no_attn", "no_mlp", "merge_only"):
    w = synth.synth_swin_weights(cfg, wseed, "stress")
    for k in list(w):
        if variant in ("no_attn", "merge_only") and (k.endswith("attn.proj.weight") or k.endswith("attn.proj.bias")):
            w[k] = np.zeros_like(w[k])
        if variant in ("no_mlp", "merge_only") and (k.endswith("mlp.fc2.weight") or k.endswith("mlp.fc2.bias")):
            w[k] = np.zeros_like(w[k])
    hw = synth.synth_vqa_head_weights(768, 64, wseed, "stress")
    net = VQA_Network({"model": {"args": {"swin_tiny_grpb": {"backbone": {}, "head": {"in_channels": 768, "hidden_channels": 64}}}}})
    sd = {f"swin_tiny_grpb_backbone.{k}": torch.from_numpy(v) for k, v in w.items()}
    sd.update({f"swin_tiny_grpb_head.{k}": torch.from_numpy(v) for k, v in hw.items()})
    net.load_state_dict(sd, strict=False)
    bb = net.swin_tiny_grpb_backbone
    bb.operand_dtype = _abi.dtype_code(dtype)
    net = net.to("cuda:0").eval()
    p = {k: torch.from_numpy(v).float() for k, v in w.items()}
    q = O.operand_rounding(odt)
    with torch.no_grad():
        taps = [bb({"technical": x.to("cuda:0")}, layer=i).cpu().permute(0, 2, 3, 4, 1).contiguous() for i in range(5)]
        for i in range(4):
            r0 = stage(taps[i], p, i, O._ident, False)
            e1 = stage(taps[i], p, i, q, True)
            print("%s %-10s stage %d: HIP vs exact %.2e | HIP vs emu(kernel order) %.2e | emu vs exact %.2e" % (
                dtype, variant, i, rel(taps[i + 1], r0), rel(taps[i + 1], e1), rel(e1, r0)), flush=True)
