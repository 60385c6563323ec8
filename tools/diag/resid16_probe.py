"""Score sensitivity to a 16-bit residual stream (CPU, fp32 oracle with the stream rounded behind the embedding, every block and every merge),
alone and together with 16-bit MFMA operands:   python tools/diag/resid16_probe.py T H W     (32 224 224: ~80 s per weight set)
Round 6, stress weights, 32x224x224: resid fp16 2.95e-06, ops fp16 3.60e-04, ops fp16 + resid fp16 3.39e-04, ops bf16 1.45e-03, ops bf16 + resid fp16 1.24e-03."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import kvq_amd
from kvq_amd.utils import synth
from oracle import swin3d_oracle as O
torch.set_num_threads(8)
cfg = synth.SWIN_T_GRPB
def trunk_r16(x, params, cfg, rdt, opdt=None):
    p = {k: O._t(v).float() for k, v in params.items()}
    q = O.operand_rounding(opdt)
    r = O.operand_rounding(rdt)
    shift = tuple(w // 2 for w in cfg.window)
    y = r(O.patch_embed(x.float(), p, cfg.patch, q))
    for i in range(len(cfg.depths)):
        for b in range(cfg.depths[i]):
            y = r(O.swin_block(y, p, f"layers.{i}.blocks.{b}.", cfg.num_heads[i], cfg.window, (0,0,0) if b % 2 == 0 else shift, q, False, None))
        if i < len(cfg.depths) - 1:
            y = r(O.patch_merge(y, p, f"layers.{i}.downsample.", q))
    y = torch.nn.functional.layer_norm(y, (y.shape[-1],), p["norm.weight"], p["norm.bias"])
    return y.permute(0, 4, 1, 2, 3).contiguous()
T,H,W = (int(v) for v in sys.argv[1:4])
for kind, seed in (("stress", 0), ("init", 3)):
    wts = synth.synth_swin_weights(cfg, seed, kind); hw = synth.synth_vqa_head_weights(768, 64, seed, kind)
    x = torch.from_numpy(synth.synth_clip(15, T, H, W, batch=1))
    t0=time.time()
    with torch.no_grad():
        s32 = O.vqa_head(O.swin3d_trunk(x, wts, cfg), hw)
        res = {}
        for name, rdt, opdt in (("resid fp16", torch.float16, None), ("resid bf16", torch.bfloat16, None), ("ops fp16", None, torch.float16), ("ops fp16 + resid fp16", torch.float16, torch.float16),
                                ("ops bf16", None, torch.bfloat16), ("ops bf16 + resid fp16", torch.float16, torch.bfloat16)):
            s = O.vqa_head(trunk_r16(x, wts, cfg, rdt, opdt), hw)
            res[name] = float((s - s32).abs().max())
    print(kind, "score", s32.ravel().tolist(), {k: f"{v:.2e}" for k, v in res.items()}, f"{time.time()-t0:.0f}s", flush=True)
