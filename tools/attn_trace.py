#!/usr/bin/env python
"""Where does an attention workgroup's lifetime go (staging vs q-tile loop)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kvq_amd
from kvq_amd import _abi, kernels
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

def run(nW, nH, B, mask):
    dev = "cuda:0"; N = 392; BW = B * nW
    qkv = (torch.randn(3, nH, BW * N, 32, device=dev) * 0.5).half()
    tok = torch.zeros(nW * N, 2, dtype=torch.int32, device=dev)
    n = torch.arange(N, device=dev)
    tok[:, 0] = ((n // 49) * 169 + ((n // 7) % 7) * 13 + n % 7).repeat(nW).int()
    rpb = torch.randn(2535, nH, device=dev); fpb = torch.randn(2535, nH, device=dev)
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    for _ in range(2): kernels.window_attention(qkv, tok, rpb, fpb, 1267, nW, N, mask)
    torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(buf.data_ptr(), 8192)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); kernels.window_attention(qkv, tok, rpb, fpb, 1267, nW, N, mask); e1.record(); torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(None, 0)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 2] != 0]
    st, lp = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
    print(f"nW={nW} nH={nH} B={B} mask={mask}: {len(t)} units, kernel {e0.elapsed_time(e1)*1e3:.1f} us; staging mean {st.mean():.0f} (p90 {np.percentile(st,90):.0f}) ticks, q-tile loop mean {lp.mean():.0f} (p90 {np.percentile(lp,90):.0f}) ticks")

def run_dense(nW, nH, B, mask):
    """needs a -DKVQ_ATT_TRACE build (KVQ_EXTRA_HIPCC_FLAGS)"""
    dev = "cuda:0"; N = 392; BW = B * nW
    qkv = (torch.randn(3, nH, BW * N, 32, device=dev) * 0.5).half()
    tok = torch.zeros(nW * N, 2, dtype=torch.int32, device=dev)
    n = torch.arange(N, device=dev)
    tok[:, 0] = ((n // 49) * 169 + ((n // 7) % 7) * 13 + n % 7).repeat(nW).int()
    rpb = torch.randn(2535, nH, device=dev); fpb = torch.randn(2535, nH, device=dev)
    dense = kernels.attn_bias_dense(tok, rpb, fpb, 1267, nW, N, mask)
    buf = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
    for _ in range(2): kernels.window_attention_dense(qkv, dense, nW, N)
    torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(buf.data_ptr(), 16384)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); kernels.window_attention_dense(qkv, dense, nW, N); e1.record(); torch.cuda.synchronize()
    _abi.lib().kvq_debug_gemm_trace(None, 0)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 2] != 0]
    st, lp = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
    print(f"dense nW={nW} nH={nH} B={B}: {len(t)} units, kernel {e0.elapsed_time(e1)*1e3:.1f} us; staging mean {st.mean():.0f}, "
          f"q-tile loop mean {lp.mean():.0f} ticks; wave 0: bias+QK {t[:,3].mean():.0f}, max/exp {t[:,4].mean():.0f}, PV+store {t[:,5].mean():.0f}")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dense":
        run_dense(128, 3, 4, True); run_dense(32, 6, 4, True); run_dense(8, 12, 4, False); run_dense(2, 24, 4, False)
        sys.exit(0)
    run(128, 3, 4, False); run(128, 3, 4, True); run(8, 12, 4, False); run(2, 24, 4, False)
