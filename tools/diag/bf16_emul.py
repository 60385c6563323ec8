"""bf16 kernels against the two fp32 emulations of bf16 operand rounding (oracle/swin3d_oracle.py): test-side diagnostic, GPU box."""
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
for dtype, odt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    for wseed, cseed, B, T, H, W in [(1, 2, 2, 8, 80, 80), (3, 4, 1, 16, 64, 64), (5, 6, 1, 8, 96, 96), (7, 8, 2, 16, 64, 64), (9, 10, 1, 32, 64, 64)]:
        net, _ = build_network("SWIN_T_GRPB", wseed, "stress", dtype)
        x = torch.from_numpy(synth.synth_clip(cseed, T, H, W, batch=B))
        hw = synth.synth_vqa_head_weights(768, 64, wseed, "stress")
        w = synth.synth_swin_weights(cfg, wseed, "stress")
        with torch.no_grad():
            score = net(inputs={"technical": x.to("cuda:0")}, reduce_scores=True).cpu().ravel()
            ref = O.vqa_head(O.swin3d_trunk(x, w, cfg), hw).ravel()
            e0 = O.vqa_head(O.swin3d_trunk(x, w, cfg, operand_dtype=odt), hw).ravel()
            e1 = O.vqa_head(O.swin3d_trunk(x, w, cfg, operand_dtype=odt, kernel_order=True), hw).ravel()
        print(dtype, (wseed, cseed, B, T, H, W), "vs ref %.2e  vs emu %.2e  vs emu(kernel order) %.2e   emu-ref %.2e" % (
            (score - ref).abs().max(), (score - e0).abs().max(), (score - e1).abs().max(), (e1 - ref).abs().max()), flush=True)
