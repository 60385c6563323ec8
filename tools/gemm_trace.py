#!/usr/bin/env python
"""Where does a GEMM block's lifetime go?  Per-block shader-clock stamps (kvq_debug_gemm_trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kvq_amd
from kvq_amd import _abi, kernels

def run(M, N, K, epi, nH=0):
    dev = "cuda:0"
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.1).half(); b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev) if epi == _abi.EPI_RESID_F32 else None
    nblk = 8192
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    for _ in range(3): kernels.gemm(A, W, b, epi, out=out, num_heads=nH)
    torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(buf.data_ptr(), nblk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); kernels.gemm(A, W, b, epi, out=out, num_heads=nH); e1.record(); torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(None, 0)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 3] != 0]
    t0 = t[:, 0].min()
    start, land, loop, end = [(t[:, i] - t0) for i in range(4)]
    clk = (end.max()) / (e0.elapsed_time(e1) * 1e3)    # ticks per us (incl. launch)
    print(f"M={M} N={N} K={K} epi={epi}: {len(t)} blocks, kernel {e0.elapsed_time(e1)*1e3:.1f} us, span {end.max()} ticks (~{clk:.0f} ticks/us)")
    for name, d in [("prologue (start->1st slice)", land - start), ("K loop", loop - land), ("epilogue", end - loop), ("lifetime", end - start)]:
        print(f"   {name:28} mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} ticks")
    order = np.argsort(start)
    print("   start times of blocks (ticks, every 128th):", start[order][::128][:12])

if __name__ == "__main__":
    if len(sys.argv) > 1:
        for e in (_abi.EPI_BIAS_BF16, _abi.EPI_GELU_BF16, _abi.EPI_STORE_F32):
            run(200704, 384, 96, e)
            run(12544, 1536, 384, e)
        sys.exit(0)
    run(12544, 1536, 384, _abi.EPI_GELU_BF16)
    run(12544, 384, 1536, _abi.EPI_RESID_F32)
    run(200704, 384, 96, _abi.EPI_GELU_BF16)
    run(3136, 768, 3072, _abi.EPI_RESID_F32)
