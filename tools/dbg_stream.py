import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import kvq_amd
from kvq_amd import kernels
import importlib
O = importlib.import_module("swin3d_oracle")
def tok_table(lay, window):
    N, nW = lay["N"], lay["nW"]
    Wd, Wh, Ww = window
    n = np.arange(N)
    code = (n // (Wh * Ww)) * (2 * Wh - 1) * (2 * Ww - 1) + ((n // Ww) % Wh) * (2 * Ww - 1) + n % Ww
    desc = lay["frag"][:, 0] | (lay["frag"][:, 1] << 8) | (lay["region"] << 16)
    tok = np.stack([np.tile(code, nW), desc], -1).astype(np.int32)
    center = (Wd - 1) * (2 * Wh - 1) * (2 * Ww - 1) + (Wh - 1) * (2 * Ww - 1) + (Ww - 1)
    return tok, center
half = torch.float16
g = np.random.default_rng(31)
window, shift = (8, 7, 7), (4, 3, 3)
lay = O.window_layout(16, 14, 14, window, shift)
N, nW, nH = lay["N"], lay["nW"], 2
q = torch.from_numpy(g.standard_normal((nW, nH, N, 32)).astype(np.float32)) * 0.6
k = torch.from_numpy(g.standard_normal((nW, nH, N, 32)).astype(np.float32))
q[:, 1, :, 0] = 16.0
k[:, 1, :, 0] = 0.0
k[:, 1, 300, 0] = 6.0
k[:, 0, 40::57] *= 9.0
rnd = lambda t: t.to(half).float()
q2, k, v = rnd(q * kernels.LOG2E), rnd(k), rnd(torch.from_numpy(g.standard_normal((nW, nH, N, 32)).astype(np.float32)))
rpb = torch.from_numpy((0.5 * g.standard_normal((2535, nH))).astype(np.float32))
ref = O.attention_core(q2 / kernels.LOG2E, k, v, rpb, None, window, lay).reshape(nW * N, nH * 32)
tok, center = tok_table(lay, window)
qkv = torch.stack([q2, k, v]).permute(0, 2, 1, 3, 4).reshape(3, nH, nW * N, 32).contiguous().to(half).cuda()
image = kernels.attn_bias_stream(torch.from_numpy(tok).cuda(), rpb.cuda(), None, center, nW, N, True)
out = kernels.window_attention_stream(qkv, image, nW, N).float().cpu()
err = (out - ref).abs()
print("max err", err.max().item())
rows = err.max(dim=1).values
bad = torch.nonzero(rows > 2e-2).flatten()
print("bad rows", bad.numel(), "of", rows.numel())
for r in bad[:12].tolist():
    e = err[r]
    print("row", r, "window", r // N, "token", r % N, "qblock", (r % N) // 32, "head0 err", e[:32].max().item(), "head1 err", e[32:].max().item())
# compare with dense kernel
qkv_d = qkv.clone(); qkv_d[0] = (qkv[0].float() / kernels.LOG2E).to(half)
dense = kernels.attn_bias_dense(torch.from_numpy(tok).cuda(), rpb.cuda(), None, center, nW, N, True)
outd = kernels.window_attention_dense(qkv_d, dense, nW, N).float().cpu()
print("dense kernel max err", (outd - ref).abs().max().item())
