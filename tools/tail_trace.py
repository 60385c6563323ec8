#!/usr/bin/env python
"""Where does a block_tail workgroup's lifetime go?  Per-block shader-clock stamps (kvq_debug_gemm_trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kvq_amd
from kvq_amd import _abi, kernels

def run(M, C, emit=False):
    dev = "cuda:0"
    hidden = 4 * C
    h = torch.float16
    A = torch.randn(M, C, device=dev).to(h); x = torch.randn(M, C, device=dev)
    Wp = (torch.randn(C, C, device=dev) * 0.1).to(h); W1 = (torch.randn(hidden, C, device=dev) * 0.1).to(h)
    W2 = (torch.randn(C, hidden, device=dev) * 0.1).to(h)
    v = lambda n: torch.randn(n, device=dev)
    pack = kernels.block_tail_pack(Wp, v(C), v(C), v(C), W1, v(hidden), W2, v(C))
    kw = {}
    if emit:
        kw = dict(next_norm=(v(C), v(C)), next_dst=torch.randperm(M, device=dev).int(), next_rows=M)
    nblk = 8192
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    for _ in range(3): kernels.block_tail(A, x, pack, hidden, **kw)
    torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(buf.data_ptr(), nblk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); kernels.block_tail(A, x, pack, hidden, **kw); e1.record(); torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(None, 0)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 4] != 0]
    print(f"M={M} C={C} emit={emit}: {len(t)} blocks, kernel {e0.elapsed_time(e1)*1e3:.1f} us")
    names = ["start -> item 0 landed", "proj + norm2", "MLP loop", "stores (+ next norm1)"]
    for i, name in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"   {name:26} mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} ticks")
    wide = C >= 256            # csrc/tailmm.hip: slot 5 = cycles in fc1 (all chunks), slot 7 = in fc2
    for i, name in ((5, "  of which: fc1" if wide else "  of which: DMA waits"), (6, "  of which: barrier waits")) + (((7, "  of which: fc2"),) if wide else ()):
        print(f"   {name:26} mean {t[:, i].mean():9.0f}  p10 {np.percentile(t[:, i],10):9.0f}  p90 {np.percentile(t[:, i],90):9.0f} ticks (wave 0, all items)")
    d = t[:, 4] - t[:, 0]
    print(f"   {'lifetime':26} mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} ticks")

if __name__ == "__main__":
    if len(sys.argv) > 3:
        run(int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])))
    elif len(sys.argv) > 2:
        run(int(sys.argv[1]), int(sys.argv[2]), True)
    else:
        run(200704, 96); run(200704, 96, True); run(50176, 192); run(50176, 192, True)
