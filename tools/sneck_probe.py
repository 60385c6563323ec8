"""slow-pathway fused res2 block on 8 clips (8 x 8 frames of 56 x 56 x 256): time, or a few launches for the counter passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kvq_amd  # noqa
from kvq_amd import kernels
from kvq_amd.models.backbones.slowfast_model import pack_slow_bottleneck
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 8, 56, 56, 256, generator=g).half().cuda()
w = lambda r, k: (torch.randn(r, k, generator=g) * (2.0 / k) ** 0.5).half().cuda()
pack = pack_slow_bottleneck(w(64, 256), torch.zeros(64).cuda(), w(64, 576), torch.zeros(64).cuda(), w(256, 64), torch.zeros(256).cuda())
out = torch.empty_like(x)
fn = lambda: kernels.slow_bottleneck(x, pack, 64, 256, out=out)
for _ in range(3): fn()
torch.cuda.synchronize()
if "pmc" in sys.argv[1:]:
    sys.exit(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print(f"slow res2 block, 8 clips: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
