"""SlowFast fast-pathway stem (Conv3d 3 -> 8, 5x7x7, stride 1x2x2) on 8 clips of 32 x 224 x 224: fp32 direct kernel vs MFMA kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kvq_amd  # noqa
from kvq_amd import kernels
x = torch.randn(8, 3, 32, 224, 224, device="cuda")
k, s, p = (5, 7, 7), (1, 2, 2), (2, 3, 3)
w = torch.randn(735, 8, device="cuda") / 27
b = torch.randn(8, device="cuda")
wp = kernels.stem_mfma_pack_weight(w, k, 3, torch.float16)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 100
if "pool" in sys.argv[1:]:
    w64 = torch.randn(64, 147, device="cuda") / 12
    wimg = kernels.stem64_pack_weight(w64, torch.float16)
    b64 = torch.randn(64, device="cuda")
    out = torch.empty(8, 8, 56, 56, 80, dtype=torch.float16, device="cuda")
    for _ in range(3):
        kernels.conv_stem_pool(x, wp, b, 5, True)
        kernels.conv_stem64_pool(x, torch.linspace(0, 31, 8).long().tolist(), wimg, b64, True, out=out)
    torch.cuda.synchronize(); sys.exit(0)
print(f"direct fp32: {t(lambda: kernels.conv_stem_direct(x, w, b, k, s, p, True, torch.float16)):.3f} ms   "
      f"mfma (pack + conv): {t(lambda: kernels.conv_stem_mfma(x, wp, b, k, s, p, True)):.3f} ms")
print(f"stem + pool in one launch: {t(lambda: kernels.conv_stem_pool(x, wp, b, 5, True)):.3f} ms")
w64 = torch.randn(64, 147, device="cuda") / 12
wimg = kernels.stem64_pack_weight(w64, torch.float16)
b64 = torch.randn(64, device="cuda")
ti = torch.linspace(0, 31, 8).long().tolist()
out = torch.empty(8, 8, 56, 56, 80, dtype=torch.float16, device="cuda")
print(f"slow stem + pool in one launch: {t(lambda: kernels.conv_stem64_pool(x, ti, wimg, b64, True, out=out)):.3f} ms")
