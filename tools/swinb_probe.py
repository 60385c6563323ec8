import sys, time, torch
sys.path.insert(0, "/root/repo")
import kvq_amd
from kvq_amd.models.backbones.swin_backbone import SwinTransformer3D
from kvq_amd.utils import synth
cfg = synth.SWIN_B_GRPB
bb = SwinTransformer3D(embed_dim=128, depths=list(cfg.depths), num_heads=list(cfg.num_heads)).to("cuda:0").eval()
x = torch.randn(2, 3, 64, 256, 256, device="cuda:0")
with torch.no_grad():
    for _ in range(2): f = bb({"technical": x})
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): f = bb({"technical": x})
    torch.cuda.synchronize()
dt = (time.time() - t) / 3
print("Swin-B 64x256x256 B=2:", f.shape, f"{dt*1e3:.1f} ms/step  {2*1892.3/dt/1e3:.0f} TFLOP/s model  finite={torch.isfinite(f).all().item()}")
print("dense:", sum(b is not None for v in bb._dense.values() for b in v), "of", sum(cfg.depths), "blocks;", torch.cuda.max_memory_allocated()/2**30, "GiB peak")
