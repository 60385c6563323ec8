"""C5 probe: Swin-B (E=128, depths 2/2/18/2, heads 4/8/16/32) on 64x256x256 clips, fp16 operands; B = argv[1] (default 2)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kvq_amd
from kvq_amd.models.backbones.swin_backbone import SwinTransformer3D
from kvq_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = synth.SWIN_B_GRPB
bb = SwinTransformer3D(embed_dim=128, depths=list(cfg.depths), num_heads=list(cfg.num_heads)).to("cuda:0").eval()
x = torch.randn(B, 3, 64, 256, 256, device="cuda:0")
with torch.no_grad():
    for _ in range(2): f = bb({"technical": x})
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): f = bb({"technical": x})
    torch.cuda.synchronize()
dt = (time.time() - t) / 3
print(f"Swin-B 64x256x256 B={B}:", tuple(f.shape), f"{dt*1e3:.1f} ms/step = {dt*1e3/B:.2f} ms/clip  {B*1892.3/dt/1e3:.0f} TFLOP/s model  finite={torch.isfinite(f).all().item()}")
print("dense:", sum(b is not None for v in bb._dense.values() for b in v), "of", sum(cfg.depths), "blocks;", f"{torch.cuda.max_memory_allocated()/2**30:.1f} GiB peak")
if len(sys.argv) > 2:      # per-launch table
    bb.profile(B, 64, 256, 256, x.device, True)
    with torch.no_grad(): bb({"technical": x})
    torch.cuda.synchronize()
    import collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in bb.profile_read(B, 64, 256, 256, x.device):
        agg[(r["kind"], r["kernel"])][0] += 1; agg[(r["kind"], r["kernel"])][1] += r["ms"]
    for (k, n), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:12} {n:60} x{c:3d} {ms*1e3:9.1f} us")
