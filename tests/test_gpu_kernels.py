"""GPU parity, kernel by kernel: every C-ABI entry point against the CPU oracle on the same
seeded inputs.  Tolerances are stated per test: integer/byte work is bit-exact; bf16-operand
MFMA work is compared against fp32 math on the *same bf16-rounded operands*."""
import os

import numpy as np
import pytest
import torch

import kvq_amd  # noqa: F401
from kvq_amd import _abi, kernels
from kvq_amd.utils import synth
from oracle import sampler_oracle as SO
from oracle import swin3d_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


HALVES = [torch.float16, torch.bfloat16]
EPS = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}     # half-ulp relative rounding error


@pytest.fixture(params=HALVES, ids=["fp16", "bf16"])
def half(request):
    return request.param


def rnd(t: torch.Tensor, half=torch.bfloat16) -> torch.Tensor:
    """fp32 -> 16-bit -> fp32 (the rounding the kernels apply to MFMA operands)."""
    return t.to(half).to(torch.float32)


bf = rnd


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


def test_library_loads_on_gpu():
    assert torch.cuda.is_available()
    name = kernels.device_name()
    assert "gfx950" in name, name


# ------------------------------------------------------------------------------------------- GEMM
@pytest.fixture(params=[-1, 1], ids=["by-shape", "wide8p"])
def tile_mode(request):
    """Every GEMM test runs through the by-shape dispatch and with the 256 x 256 x 64 eight-phase kernel forced on every eligible
    shape (K % 64 == 0): ragged M / N, all epilogues, scatter / gather maps."""
    prev = kernels.gemm_tile_mode(request.param)
    yield request.param
    kernels.gemm_tile_mode(prev)


@pytest.mark.parametrize("M,N,K", [(200, 96, 96), (1000, 288, 96), (130, 768, 3072), (8960, 1152, 384),
                                    (3136, 64, 768), (64 * 9 + 5, 2304, 768), (517, 520, 128), (1100, 264, 1024)])
def test_gemm_store_and_bias(M, N, K, half, tile_mode):
    g = rng(M + N + K)
    A = rnd(torch.from_numpy(g.standard_normal((M, K)).astype(np.float32)), half)
    W = rnd(torch.from_numpy(g.standard_normal((N, K)).astype(np.float32)), half)
    b = torch.from_numpy(g.standard_normal(N).astype(np.float32))
    ref = A.double() @ W.double().t() + b.double()
    out = kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_STORE_F32)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-4 * np.sqrt(K), err       # fp32 accumulation order only
    out_bf = kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_BIAS_BF16)
    assert (out_bf.float().cpu().double() - ref).abs().max().item() <= EPS[half] * ref.abs().max().item() + 1e-4
    out_g = kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_GELU_BF16)
    ref_g = torch.nn.functional.gelu(ref.float())
    assert (out_g.float().cpu() - ref_g).abs().max().item() <= EPS[half] * ref_g.abs().max().item() + 1e-4


def test_gemm_no_bias_and_asymmetric_layout(half, tile_mode):
    # A = I (padded) with an asymmetric W catches a transposed C-write or swapped operands
    M = N = K = 128
    A = torch.eye(M)
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125
    out = kernels.gemm(dev(A, half), dev(W, half), None, _abi.EPI_STORE_F32)
    assert torch.equal(out.cpu(), W.t().contiguous())


@pytest.mark.parametrize("nH", [3, 12])
def test_gemm_qkv_epilogue(half, tile_mode, nH):
    g = rng(5)
    M, C = 392 * 2, 32 * nH
    A = rnd(torch.from_numpy(g.standard_normal((M, C)).astype(np.float32)), half)
    W = rnd(torch.from_numpy(g.standard_normal((3 * C, C)).astype(np.float32) * 0.2), half)
    b = torch.from_numpy(g.standard_normal(3 * C).astype(np.float32))
    scale = 32 ** -0.5
    out = kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_QKV_BF16, num_heads=nH,
                       q_scale=scale)
    ref = (A @ W.t() + b).reshape(M, 3, nH, 32).permute(1, 2, 0, 3).clone()
    ref[0] *= scale
    assert out.shape == (3, nH, M, 32)
    assert (out.float().cpu() - ref).abs().max().item() <= EPS[half] * ref.abs().max().item() + 1e-4


@pytest.mark.parametrize("C", [96, 384])
def test_padded_partition_qkv_over_tokens_and_proj_row_gather(half, tile_mode, C):
    """Padded + shifted partition: (i) the qkv GEMM over the TOKENS with its rows scattered to their window rows + ``qkv_fill_pad`` for the
    padding rows == the qkv GEMM over all window rows of the zero-padded input, bit for bit; (ii) the proj GEMM with ``a_gather``
    (token -> window row) == the proj GEMM over the window rows with the scatter map, bit for bit."""
    g = rng(61)
    lay = O.window_layout(4, 10, 9, (8, 7, 7), (4, 3, 3))
    Lp, L, B, nH = lay["nW"] * lay["N"], 4 * 10 * 9, 2, C // 32
    src = lay["src"].astype(np.int64)
    inv = np.zeros(L, np.int32)
    inv[src[src >= 0]] = np.nonzero(src >= 0)[0].astype(np.int32)
    pad = np.nonzero(src < 0)[0].astype(np.int32)
    tok = rnd(torch.from_numpy(g.standard_normal((B * L, C)).astype(np.float32)), half)          # norm1 output, token order
    win = torch.zeros(B, Lp, C)
    win[:, src >= 0] = tok.reshape(B, L, C)[:, src[src >= 0]]                                     # window order, zero padding rows
    W = rnd(torch.from_numpy(g.standard_normal((3 * C, C)).astype(np.float32) * 0.2), half)
    b = torch.from_numpy(g.standard_normal(3 * C).astype(np.float32))
    scale = 32 ** -0.5
    full = kernels.gemm(dev(win.reshape(B * Lp, C), half), dev(W, half), dev(b), _abi.EPI_QKV_BF16, num_heads=nH, q_scale=scale)
    part = torch.full_like(full, float("nan"))
    kernels.gemm(dev(tok, half), dev(W, half), dev(b), _abi.EPI_QKV_BF16, num_heads=nH, q_scale=scale, out=part,
                 scatter_map=dev(torch.from_numpy(inv)), map_rows=L, out_rows=Lp)
    kernels.qkv_fill_pad(part, dev(b), dev(torch.from_numpy(pad)), B, scale)
    assert torch.equal(part, full)
    # proj
    A = rnd(torch.from_numpy(g.standard_normal((B * Lp, C)).astype(np.float32)), half)
    Wp = rnd(torch.from_numpy(g.standard_normal((C, C)).astype(np.float32) * 0.2), half)
    bp = torch.from_numpy(g.standard_normal(C).astype(np.float32))
    x = torch.from_numpy(g.standard_normal((B * L, C)).astype(np.float32))
    x_rows, x_tok = dev(x.clone()), dev(x.clone())
    kernels.gemm(dev(A, half), dev(Wp, half), dev(bp), _abi.EPI_RESID_F32, out=x_rows,
                 scatter_map=dev(torch.from_numpy(lay["src"].astype(np.int32))), map_rows=Lp, out_rows=L)
    kernels.gemm(dev(A, half), dev(Wp, half), dev(bp), _abi.EPI_RESID_F32, out=x_tok, a_gather=dev(torch.from_numpy(inv)), a_rows=L,
                 rows=B * L)
    assert torch.equal(x_rows, x_tok)


@pytest.mark.parametrize("C", [96, 256])
def test_gemm_residual_scatter(half, tile_mode, C):
    g = rng(6)
    lay = O.window_layout(4, 10, 9, (8, 7, 7), (4, 3, 3))       # padded + shifted: rows dropped and permuted
    Lp, L, B = lay["nW"] * lay["N"], 4 * 10 * 9, 2
    A = rnd(torch.from_numpy(g.standard_normal((B * Lp, C)).astype(np.float32)), half)
    W = rnd(torch.from_numpy(g.standard_normal((C, C)).astype(np.float32) * 0.2), half)
    b = torch.from_numpy(g.standard_normal(C).astype(np.float32))
    x = torch.from_numpy(g.standard_normal((B * L, C)).astype(np.float32))
    src = torch.from_numpy(lay["src"].astype(np.int32))
    xd = dev(x.clone())
    kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_RESID_F32, out=xd,
                 scatter_map=dev(src), map_rows=Lp, out_rows=L)
    y = (A @ W.t() + b).reshape(B, Lp, C)
    ref = x.reshape(B, L, C) + O.scatter_windows(y, lay, B, 4, 10, 9).reshape(B, L, C)
    assert (xd.cpu().reshape(B, L, C) - ref).abs().max().item() <= 1e-4
    # identity map
    x2 = dev(x[:300].clone())
    A2 = A[:300]
    kernels.gemm(dev(A2, half), dev(W, half), dev(b), _abi.EPI_RESID_F32, out=x2)
    assert (x2.cpu() - (x[:300] + A2 @ W.t() + b)).abs().max().item() <= 1e-4


@pytest.mark.parametrize("M,N,K", [(300, 256, 4096), (196, 256, 1536), (98, 768, 3072)])
def test_gemm_split_k_every_epilogue(M, N, K, half, tile_mode):
    """Long K, few output tiles (the late convolutions of the conv nets, stage 3 of the trunk): the launch cuts K into S ranges,
    partial tiles go to the caller's scratch and a second launch adds them IN ORDER and applies the epilogue.  Every epilogue that
    may split against the fp64 product; and twice the same call -> bit-identical (the order of the partial sums is fixed)."""
    S = _abi.lib().kvq_gemm_splitk_factor(M, N, K)
    assert S > 1 and _abi.lib().kvq_gemm_splitk_bytes(M, N, K) >= S * M * N * 4
    g = rng(M + N + K)
    A = rnd(torch.from_numpy(g.standard_normal((M, K)).astype(np.float32)), half)
    W = rnd(torch.from_numpy((g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)), half)
    b = torch.from_numpy(g.standard_normal(N).astype(np.float32))
    ref = A.double() @ W.double().t() + b.double()
    out = kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_STORE_F32)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5 * np.sqrt(K)
    assert torch.equal(out, kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_STORE_F32))
    out_h = kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_BIAS_BF16)
    assert (out_h.float().cpu().double() - ref).abs().max().item() <= EPS[half] * ref.abs().max().item() + 1e-4
    out_g = kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_GELU_BF16)
    ref_g = torch.nn.functional.gelu(ref.float())
    assert (out_g.float().cpu() - ref_g).abs().max().item() <= EPS[half] * ref_g.abs().max().item() + 1e-4
    # residual accumulate (identity map) and conv-style ReLU with an fp32 identity branch + fp32 copy
    x = torch.from_numpy(g.standard_normal((M, N)).astype(np.float32))
    xd = dev(x.clone())
    kernels.gemm(dev(A, half), dev(W, half), dev(b), _abi.EPI_RESID_F32, out=xd)
    assert (xd.cpu().double() - (x.double() + ref)).abs().max().item() <= 2e-5 * np.sqrt(K)
    y16, y32 = kernels.conv_gemm(dev(A, half), dev(W, half), dev(b), True, resid_f32=dev(x), want_f32=True)
    ref_r = torch.relu(ref + x.double())
    assert (y32.cpu().double() - ref_r).abs().max().item() <= 2e-5 * np.sqrt(K)
    assert (y16.float().cpu().double() - ref_r).abs().max().item() <= EPS[half] * ref_r.abs().max().item() + 1e-4


def test_conv_implicit_split_k_matches_conv3d():
    """A temporal 3x1x1 convolution over 512 channels on a small map (the res5 geometry of SlowFast's slow pathway): K = 1536 over
    8 output tiles -> split-K; against F.conv3d of the rounded operands."""
    g = rng(99)
    B, D, H, W, Cc, N = 1, 4, 7, 7, 512, 256
    x = torch.from_numpy(g.standard_normal((B, D, H, W, Cc)).astype(np.float32)).half()
    w5 = torch.from_numpy((g.standard_normal((N, Cc, 3, 1, 1)) / np.sqrt(3 * Cc)).astype(np.float32)).half()
    wk = w5.permute(0, 2, 3, 4, 1).reshape(N, 3 * Cc).contiguous()
    bias = torch.from_numpy(g.standard_normal(N).astype(np.float32))
    assert _abi.lib().kvq_gemm_splitk_bytes(B * D * H * W, N, 3 * Cc) > 0
    got = kernels.conv_implicit(x.cuda(), wk.cuda(), bias.cuda(), (3, 1, 1), (1, 1, 1), (1, 0, 0), True)
    ref = torch.relu(torch.nn.functional.conv3d(x.float().permute(0, 4, 1, 2, 3), w5.float(), bias, 1, (1, 0, 0))).permute(0, 2, 3, 4, 1)
    assert got.shape == ref.shape
    assert (got.float().cpu() - ref).abs().max().item() <= 2.0 ** -10 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape,Cout,k,stride,pad", [
    ((2, 5, 9, 11, 64), 48, (3, 1, 1), (1, 1, 1), (1, 0, 0)),       # temporal taps, borders in time
    ((2, 3, 13, 10, 32), 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)),      # spatial taps, stride 2, one chunk group per tap
    ((1, 4, 8, 8, 96), 40, (3, 3, 3), (1, 1, 1), (1, 1, 1)),        # 27 taps x 3 slices each, K padding behind the last tap
    ((3, 1, 7, 7, 160), 256, (1, 1, 1), (1, 2, 2), (0, 0, 0)),      # strided 1x1 (projection shortcut)
])
def test_conv_implicit_tap_walk_equals_tap_table(shape, Cout, k, stride, pad, half):
    """C % 32 == 0: the kernel walks (kd, kh, kw, c) with wave-uniform counters instead of reading the tap table — same slices in the
    same order, so the results are bit-identical to the table-driven instantiation; a table is still required when C % 32 != 0."""
    g = rng(sum(shape) + Cout)
    Cc = shape[-1]
    K = k[0] * k[1] * k[2] * Cc
    kpad = -(-K // 32) * 32
    x = dev(torch.from_numpy(g.standard_normal(shape).astype(np.float32)).to(half))
    w = torch.zeros(Cout, kpad)
    w[:, :K] = torch.from_numpy((g.standard_normal((Cout, K)) / np.sqrt(K)).astype(np.float32))
    w, bias = dev(w.to(half)), dev(torch.from_numpy(g.standard_normal(Cout).astype(np.float32)))
    assert not kernels.TAP_TABLE
    walked = kernels.conv_implicit(x, w, bias, k, stride, pad, True)
    kernels.TAP_TABLE = True
    try:
        tabled = kernels.conv_implicit(x, w, bias, k, stride, pad, True)
    finally:
        kernels.TAP_TABLE = False
    assert torch.equal(walked, tabled)
    a = _abi.KvqConvArgs()
    x8 = dev(torch.zeros(1, 1, 4, 4, 8, dtype=half))
    w8 = dev(torch.zeros(8, 32, dtype=half))
    o8 = dev(torch.zeros(16, 8, dtype=half))
    a.x, a.W, a.out_bf16 = _abi.ptr(x8), _abi.ptr(w8), _abi.ptr(o8)
    a.dims5[:] = (1, 8, 1, 4, 4)
    a.kernel3[:], a.stride3[:], a.pad3[:] = (1, 1, 1), (1, 1, 1), (0, 0, 0)
    a.Kpad, a.N, a.epilogue, a.dtype = 32, 8, _abi.EPI_BIAS_BF16, _abi.dtype_code(half)
    import ctypes
    assert _abi.lib().kvq_conv_implicit(ctypes.byref(a), _abi.current_stream()) != 0
    assert b"tap table" in _abi.lib().kvq_last_error()


def test_gemm_rejects_bad_shapes():
    A = torch.zeros(8, 40, dtype=torch.float16, device=DEV)
    W = torch.zeros(32, 40, dtype=torch.float16, device=DEV)
    with pytest.raises(_abi.KvqError, match="K%32"):
        kernels.gemm(A, W, None, _abi.EPI_STORE_F32)


# -------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("C", [96, 128, 192, 384, 768, 1024])
def test_layernorm_identity(C, half):
    g = rng(C)
    x = torch.from_numpy((g.standard_normal((777, C)) * 3 + 1.5).astype(np.float32))
    ga = torch.from_numpy((1 + 0.1 * g.standard_normal(C)).astype(np.float32))
    be = torch.from_numpy((0.1 * g.standard_normal(C)).astype(np.float32))
    ref = torch.nn.functional.layer_norm(x, (C,), ga, be)
    out = kernels.layernorm_rows(dev(x), dev(ga), dev(be), out_dtype=torch.float32)
    assert (out.cpu() - ref).abs().max().item() <= 2e-5
    out_h = kernels.layernorm_rows(dev(x), dev(ga), dev(be), out_dtype=half)
    assert torch.equal(out_h.cpu(), out.cpu().to(half))      # same values, RNE rounding


@pytest.mark.parametrize("dims,shift", [((8, 14, 14), (0, 0, 0)), ((8, 14, 14), (4, 3, 3)), ((4, 10, 9), (4, 3, 3)),
                                        ((10, 9, 23), (4, 3, 3))])
def test_layernorm_window_gather(dims, shift):
    D, H, W = dims
    g = rng(sum(dims))
    C, B = 96, 2
    x = torch.from_numpy(g.standard_normal((B, D, H, W, C)).astype(np.float32))
    ga = torch.from_numpy((1 + 0.1 * g.standard_normal(C)).astype(np.float32))
    be = torch.from_numpy((0.1 * g.standard_normal(C)).astype(np.float32))
    lay = O.window_layout(D, H, W, (8, 7, 7), shift)
    ref = O.gather_windows(torch.nn.functional.layer_norm(x, (C,), ga, be), lay).reshape(-1, C)
    src = dev(torch.from_numpy(lay["src"].astype(np.int32)))
    out = kernels.layernorm_rows(dev(x.reshape(-1, C)), dev(ga), dev(be), index_map=src, n_batch=B,
                                 rows_out=lay["nW"] * lay["N"], out_dtype=torch.float32)
    assert (out.cpu() - ref).abs().max().item() <= 2e-5
    pad = torch.from_numpy(np.tile(lay["src"] < 0, B))
    assert torch.all(out.cpu()[pad] == 0)            # pad AFTER the norm: exact zeros, not beta


def _merge_map(D, H, W):
    Hn, Wn = (H + 1) // 2, (W + 1) // 2
    mp = np.full((D, Hn, Wn, 4), -1, np.int32)
    for d in range(D):
        for h2 in range(Hn):
            for w2 in range(Wn):
                for part, (dh, dw) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):      # concat order x0 x1 x2 x3 (swin_backbone.py:546-550)
                    hh, ww = 2 * h2 + dh, 2 * w2 + dw
                    if hh < H and ww < W:
                        mp[d, h2, w2, part] = (d * H + hh) * W + ww
    return mp.reshape(-1, 4), Hn, Wn


@pytest.mark.parametrize("dims,emit,C", [((2, 4, 8, 8), True, 96), ((2, 3, 5, 7), False, 96), ((3, 8, 14, 14), True, 96),
                                         ((2, 8, 14, 14), True, 192), ((2, 2, 9, 6), False, 192), ((2, 8, 14, 14), True, 128), ((2, 3, 5, 7), False, 128)])
def test_patch_merge_fused_vs_oracle(dims, emit, C, half):
    """PatchMerging as one launch (concat + LayerNorm(4C) folded around the reduction GEMM [+ the next norm1 in window order])
    against the fp32 oracle: the launch rounds (x - K) and W diag(gamma) to 16 bits where the reference rounds nothing, so the
    bound is the operand rounding over K = 4C terms — and against the three-launch sequence, which rounds LN(x) and W.  C = 96: the
    matrix is LDS-resident; C = 128 / 192: it streams through two chunk buffers (4 / 8 chunks)."""
    B, D, H, W = dims
    g = rng(sum(dims) + C)
    x = torch.from_numpy((g.standard_normal((B, D, H, W, C)) * 1.5 + 0.3 * g.standard_normal((1, 1, 1, 1, C))).astype(np.float32))
    # the launch's statistics are a shifted one-pass: tokens far from zero with a small spread, and constant tokens, must hold
    x[:, 0, :2] = 50.0 + 0.05 * x[:, 0, :2]
    x[:, -1, -2:, -2:] = -3.25
    p = {"m.norm.weight": torch.from_numpy((1 + 0.2 * g.standard_normal(4 * C)).astype(np.float32)),
         "m.norm.bias": torch.from_numpy((0.2 * g.standard_normal(4 * C)).astype(np.float32)),
         "m.reduction.weight": torch.from_numpy((g.standard_normal((2 * C, 4 * C)) / np.sqrt(4 * C)).astype(np.float32))}
    ref = O.patch_merge(x, p, "m.")                                        # (B, D, Hn, Wn, 2C) fp32
    mp, Hn, Wn = _merge_map(D, H, W)
    Ln = D * Hn * Wn
    kw = {}
    if emit:
        lay = O.window_layout(D, Hn, Wn, (8, 7, 7), (0, 0, 0))
        if (lay["src"] < 0).any():
            kw = {}
            emit = False
        else:
            dst = np.empty(Ln, np.int32)
            dst[lay["src"]] = np.arange(Ln, dtype=np.int32)
            gn = torch.from_numpy((1 + 0.2 * g.standard_normal(2 * C)).astype(np.float32))
            bn = torch.from_numpy((0.2 * g.standard_normal(2 * C)).astype(np.float32))
            kw = dict(next_norm=(dev(gn), dev(bn)), next_dst=dev(torch.from_numpy(dst)), next_rows=Ln)
    out, nxt = kernels.patch_merge(dev(x.reshape(-1, C)), dev(torch.from_numpy(mp)), B, dev(p["m.reduction.weight"]),
                                   dev(p["m.norm.weight"]), dev(p["m.norm.bias"]), out_dtype=half, **kw)
    refm = ref.reshape(-1, 2 * C)
    scale = max(1.0, refm.abs().max().item())
    assert (out.cpu() - refm).abs().max().item() <= 4 * EPS[half] * scale
    # the three launches it replaces: gather-LayerNorm (16-bit) -> GEMM on 16-bit weights
    ln = kernels.layernorm_rows(dev(x.reshape(-1, C)), dev(p["m.norm.weight"]), dev(p["m.norm.bias"]), index_map=dev(torch.from_numpy(mp)),
                                nparts=4, n_batch=B, rows_out=Ln, out_dtype=half)
    chain = kernels.gemm(ln, dev(p["m.reduction.weight"], half), None, _abi.EPI_STORE_F32)
    assert (out - chain).abs().max().item() <= 4 * EPS[half] * scale
    if emit:
        lnn = torch.nn.functional.layer_norm(out.cpu(), (2 * C,), gn, bn).reshape(B, D, Hn, Wn, 2 * C)
        ref_ln = O.gather_windows(lnn, lay).reshape(-1, 2 * C)
        assert (nxt.float().cpu() - ref_ln).abs().max().item() <= 2 * EPS[half] * ref_ln.abs().max().item() + 1e-5
    # a token's result does not depend on the batch around it
    one, _ = kernels.patch_merge(dev(x[1:2].reshape(-1, C)), dev(torch.from_numpy(mp)), 1, dev(p["m.reduction.weight"]),
                                 dev(p["m.norm.weight"]), dev(p["m.norm.bias"]), out_dtype=half)
    assert torch.equal(one, out[Ln:2 * Ln])


@pytest.mark.parametrize("C", [96, 128])
def test_patch_merge_fused_outlier_channel(C, half):
    """A massive-activation channel (x[..., 0] = 50 on otherwise N(0, 1.5) tokens) must not cost the other 4C - 1 operands their
    precision: the shift K of the one-pass statistics is a 96-channel mean, not the token's first value.  Same bound as
    the three-launch sequence holds against the fp32 oracle."""
    B, D, H, W = 2, 4, 8, 8
    g = rng(77 + C)
    x = torch.from_numpy((g.standard_normal((B, D, H, W, C)) * 1.5).astype(np.float32))
    x[..., 0] = 50.0
    x[0, 0, 0, 0, 0] = -80.0
    p = {"m.norm.weight": torch.from_numpy((1 + 0.2 * g.standard_normal(4 * C)).astype(np.float32)),
         "m.norm.bias": torch.from_numpy((0.2 * g.standard_normal(4 * C)).astype(np.float32)),
         "m.reduction.weight": torch.from_numpy((g.standard_normal((2 * C, 4 * C)) / np.sqrt(4 * C)).astype(np.float32))}
    ref = O.patch_merge(x, p, "m.").reshape(-1, 2 * C)
    mp, Hn, Wn = _merge_map(D, H, W)
    Ln = D * Hn * Wn
    out, _ = kernels.patch_merge(dev(x.reshape(-1, C)), dev(torch.from_numpy(mp)), B, dev(p["m.reduction.weight"]),
                                 dev(p["m.norm.weight"]), dev(p["m.norm.bias"]), out_dtype=half)
    ln = kernels.layernorm_rows(dev(x.reshape(-1, C)), dev(p["m.norm.weight"]), dev(p["m.norm.bias"]), index_map=dev(torch.from_numpy(mp)),
                                nparts=4, n_batch=B, rows_out=Ln, out_dtype=half)
    chain = kernels.gemm(ln, dev(p["m.reduction.weight"], half), None, _abi.EPI_STORE_F32)
    scale = max(1.0, ref.abs().max().item())
    e_fused = (out.cpu() - ref).abs().max().item()
    e_chain = (chain.cpu() - ref).abs().max().item()
    # the outlier itself is rounded at |50| in both forms; everything else must stay at the operand-rounding level
    assert e_fused <= 4 * EPS[half] * scale, (e_fused, e_chain)
    assert e_fused <= 3 * e_chain + 1e-4, (e_fused, e_chain)


def test_patch_merge_rejects_other_widths():
    with pytest.raises(_abi.KvqError, match="unsupported width"):
        kernels.patch_merge(torch.zeros(16, 256, device=DEV), torch.zeros(4, 4, dtype=torch.int32, device=DEV), 1,
                            torch.zeros(512, 1024, device=DEV), torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV))


def test_layernorm_merge_gather_odd_dims():
    g = rng(9)
    B, D, H, W, C = 2, 3, 5, 7, 96
    x = torch.from_numpy(g.standard_normal((B, D, H, W, C)).astype(np.float32))
    p = {"m.norm.weight": torch.from_numpy((1 + 0.1 * g.standard_normal(4 * C)).astype(np.float32)),
         "m.norm.bias": torch.from_numpy((0.1 * g.standard_normal(4 * C)).astype(np.float32)),
         "m.reduction.weight": torch.eye(4 * C)}
    ref = O.patch_merge(x, p, "m.").reshape(-1, 4 * C)          # identity reduction -> the normalised concat
    Hn, Wn = 3, 4
    mp = np.full((D, Hn, Wn, 4), -1, np.int32)
    for d in range(D):
        for h2 in range(Hn):
            for w2 in range(Wn):
                for part, (dh, dw) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
                    hh, ww = 2 * h2 + dh, 2 * w2 + dw
                    if hh < H and ww < W:
                        mp[d, h2, w2, part] = (d * H + hh) * W + ww
    out = kernels.layernorm_rows(dev(x.reshape(-1, C)), dev(p["m.norm.weight"]), dev(p["m.norm.bias"]),
                                 index_map=dev(torch.from_numpy(mp.reshape(-1, 4))), nparts=4, n_batch=B,
                                 rows_out=D * Hn * Wn, out_dtype=torch.float32)
    assert (out.cpu() - ref).abs().max().item() <= 3e-5


# ------------------------------------------------------------------------------- window attention
def _tok_table(lay, window):
    N, nW = lay["N"], lay["nW"]
    Wd, Wh, Ww = window
    n = np.arange(N)
    code = (n // (Wh * Ww)) * (2 * Wh - 1) * (2 * Ww - 1) + ((n // Ww) % Wh) * (2 * Ww - 1) + n % Ww
    desc = lay["frag"][:, 0] | (lay["frag"][:, 1] << 8) | (lay["region"] << 16)
    tok = np.stack([np.tile(code, nW), desc], -1).astype(np.int32)
    center = (Wd - 1) * (2 * Wh - 1) * (2 * Ww - 1) + (Wh - 1) * (2 * Ww - 1) + (Ww - 1)
    return tok, center


def test_window_attention32_cooperative_last_qblock():
    """KVQ_ATTN_COOP=2 (csrc/attn32.hip, round 6: the 13th q-block of an N = 392 window cut along the keys into four ranges, merged in range
    order through LDS) holds the same oracle gates, tile_skip semantics and run-to-run bit-equality as the default form: the library reads
    the variable once, so the attention tests run again in a child interpreter with it set."""
    import subprocess
    import sys
    env = dict(os.environ, KVQ_ATTN_COOP="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "window_attention32_vs_oracle or window_attention32_fused or padded_partition"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


@pytest.mark.parametrize("dims,window,shifted,gated,nH", [
    ((8, 14, 14), (8, 7, 7), False, True, 3),
    ((8, 14, 14), (8, 7, 7), True, True, 3),
    ((16, 7, 7), (8, 7, 7), True, False, 2),      # stage-3 like: spatial shift clamped away, no fragment table
    ((4, 10, 9), (8, 7, 7), True, True, 1),       # clamped depth (N=196) + padding
    ((8, 8, 8), (4, 4, 4), True, False, 2),       # swin_tiny_grpb_m window
    ((8, 14, 14), (8, 7, 7), False, False, 6),
])
def test_window_attention(dims, window, shifted, gated, nH, half):
    g = rng(sum(dims) + nH)
    shift = tuple(w // 2 for w in window) if shifted else (0, 0, 0)
    lay = O.window_layout(*dims, window, shift)
    N, nW, B = lay["N"], lay["nW"], 2
    BW = B * nW
    tl = (2 * window[0] - 1) * (2 * window[1] - 1) * (2 * window[2] - 1)
    q = rnd(torch.from_numpy(g.standard_normal((BW, nH, N, 32)).astype(np.float32)) * 0.6, half)
    k = rnd(torch.from_numpy(g.standard_normal((BW, nH, N, 32)).astype(np.float32)), half)
    v = rnd(torch.from_numpy(g.standard_normal((BW, nH, N, 32)).astype(np.float32)), half)
    rpb = torch.from_numpy(g.standard_normal((tl, nH)).astype(np.float32))
    fpb = torch.from_numpy(g.standard_normal((tl, nH)).astype(np.float32)) if gated else None
    ref = O.attention_core(q, k, v, rpb, fpb, window, lay).reshape(BW * N, nH * 32)
    tok, center = _tok_table(lay, window)
    qkv = torch.stack([q, k, v]).permute(0, 2, 1, 3, 4).reshape(3, nH, BW * N, 32).contiguous()
    use_mask = any(s > 0 for s in lay["ss"])
    out = kernels.window_attention(dev(qkv, half), dev(torch.from_numpy(tok)), dev(rpb),
                                   None if fpb is None else dev(fpb), center, nW, N, use_mask)
    err = (out.float().cpu() - ref).abs().max().item()
    # P and the output are rounded to 16 bits: EPS relative on O(1) values (|v| up to ~4.5)
    assert err <= 6.4 * EPS[half], err
    assert (out.float().cpu() - ref).abs().mean().item() <= 0.5 * EPS[half]


def test_window_attention_softmax_extremes(half):
    """One key dominating by > 80 in the logits, and the -100 mask, must not produce NaN/Inf."""
    g = rng(3)
    lay = O.window_layout(8, 14, 14, (8, 7, 7), (4, 3, 3))
    N, nW, nH = lay["N"], lay["nW"], 1
    q = torch.zeros(nW, nH, N, 32)
    k = torch.zeros(nW, nH, N, 32)
    q[:, :, :, 0] = 16.0
    k[:, :, 17, 0] = 6.0
    v = rnd(torch.from_numpy(g.standard_normal((nW, nH, N, 32)).astype(np.float32)), half)
    rpb = torch.zeros(2535, 1)
    ref = O.attention_core(q, k, v, rpb, None, (8, 7, 7), lay).reshape(nW * N, 32)
    tok, center = _tok_table(lay, (8, 7, 7))
    qkv = torch.stack([q, k, v]).permute(0, 2, 1, 3, 4).reshape(3, nH, nW * N, 32).contiguous()
    out = kernels.window_attention(dev(qkv, half), dev(torch.from_numpy(tok)), dev(rpb), None, center, nW,
                                   N, True).float().cpu()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= 6.4 * EPS[half]


def test_window_attention32_cooperative_last_qblock():
    """KVQ_ATTN_COOP=2 (csrc/attn32.hip, round 6: the 13th q-block of an N = 392 window cut along the keys into four ranges, merged in range
    order through LDS) holds the same oracle gates, tile_skip semantics and run-to-run bit-equality as the default form: the library reads
    the variable once, so the attention tests run again in a child interpreter with it set."""
    import subprocess
    import sys
    env = dict(os.environ, KVQ_ATTN_COOP="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "window_attention32_vs_oracle or window_attention32_fused or padded_partition"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


@pytest.mark.parametrize("dims,window,shifted,gated,nH", [
    ((8, 14, 14), (8, 7, 7), False, True, 3),
    ((8, 14, 14), (8, 7, 7), True, True, 3),
    ((16, 7, 7), (8, 7, 7), True, False, 2),
    ((4, 10, 9), (8, 7, 7), True, True, 1),       # clamped depth (N = 196: 7 key blocks, the last one 4 keys wide) + padding
    ((8, 8, 8), (4, 4, 4), True, False, 2),       # N = 64: two key blocks
    ((16, 14, 7), (8, 7, 7), False, True, 2),     # two windows deep, un-shifted: the depth copies share one bias
    ((16, 28, 28), (8, 7, 7), True, True, 6),     # stage-1 like: 32 windows x 6 heads, several entries per workgroup
])
def test_window_attention32_vs_oracle_and_gather_path(dims, window, shifted, gated, nH, half):
    """The pre-built-bias kernel (32 x 32 score blocks, running maximum; csrc/attn32.hip) against the fp32 oracle and against the per-score
    gather kernel (the bias differs by the image's fp16 rounding).  q reaches it scaled by log2(e) (scores in log2 units): the oracle
    gets the same rounded q divided by log2(e) in fp32, the gather kernel that quotient rounded to 16 bits again."""
    g = rng(sum(dims) + nH + 200)
    shift = tuple(w // 2 for w in window) if shifted else (0, 0, 0)
    lay = O.window_layout(*dims, window, shift)
    N, nW, B = lay["N"], lay["nW"], 3
    BW = B * nW
    tl = (2 * window[0] - 1) * (2 * window[1] - 1) * (2 * window[2] - 1)
    q2 = rnd(torch.from_numpy(g.standard_normal((BW, nH, N, 32)).astype(np.float32)) * (0.6 * kernels.LOG2E), half)
    k = rnd(torch.from_numpy(g.standard_normal((BW, nH, N, 32)).astype(np.float32)), half)
    v = rnd(torch.from_numpy(g.standard_normal((BW, nH, N, 32)).astype(np.float32)), half)
    rpb = torch.from_numpy((0.5 * g.standard_normal((tl, nH))).astype(np.float32))
    fpb = torch.from_numpy((0.5 * g.standard_normal((tl, nH))).astype(np.float32)) if gated else None
    ref = O.attention_core(q2 / kernels.LOG2E, k, v, rpb, fpb, window, lay).reshape(BW * N, nH * 32)
    tok, center = _tok_table(lay, window)
    qkv = dev(torch.stack([q2, k, v]).permute(0, 2, 1, 3, 4).reshape(3, nH, BW * N, 32).contiguous(), half)
    use_mask = any(s > 0 for s in lay["ss"])
    tokd, rpbd, fpbd = dev(torch.from_numpy(tok)), dev(rpb), None if fpb is None else dev(fpb)
    n_types = nW if use_mask else nW // (-(-dims[0] // lay["ws"][0]))
    image = kernels.attn_bias32(tokd[: n_types * N], rpbd, fpbd, center, n_types, N, use_mask)
    out = kernels.window_attention32(qkv, image, nW, N, n_types).float().cpu()
    assert torch.isfinite(out).all()
    assert 0 < float(image.max_abs_bias) <= 32.0
    # against the oracle fed the bias as the image holds it (fp16 of bias - row maximum): the 16-bit rounding of the probabilities and of
    # the output only; the image's own rounding against the exact fp32 bias is bounded separately
    ref_img = O.attention_core(q2 / kernels.LOG2E, k, v, rpb, fpb, window, lay, image=True).reshape(BW * N, nH * 32)
    assert (out - ref_img).abs().max().item() <= 6.4 * EPS[half]
    assert (ref_img - ref).abs().max().item() <= 2.0 ** -7
    assert (out - ref).abs().mean().item() <= 0.5 * EPS[half]
    # the exact per-score path on the same inputs: the two kernels' roundings + the image's + ONE MORE 16-bit rounding of q (the gather
    # kernel takes q un-scaled; re-rounding q moves every logit by up to 2^-9 (bf16) / 2^-12 (fp16) of its size)
    qg = rnd(q2 / kernels.LOG2E, half)
    qkv_g = dev(torch.stack([qg, k, v]).permute(0, 2, 1, 3, 4).reshape(3, nH, BW * N, 32).contiguous(), half)
    gather = kernels.window_attention(qkv_g, tokd, rpbd, fpbd, center, nW, N, use_mask).float().cpu()
    assert (out - gather).abs().max().item() <= 12.0 * EPS[half] + (ref_img - ref).abs().max().item()
    # q-blocks whose two 16-row tiles are both marked in tile_skip are passed over: their rows keep the sentinel
    skip = np.zeros(nW, np.int32)
    nqt = -(-N // 16)
    for wv in range(nW):
        skip[wv] = int(g.integers(0, 1 << nqt)) & ~3                       # q-block 0 always runs
    sentinel = torch.full((BW * N, nH * 32), 7.0, dtype=half, device=qkv.device)
    part = kernels.window_attention32(qkv, image, nW, N, n_types, tile_skip=dev(torch.from_numpy(skip)), out=sentinel).float().cpu()
    rows = np.arange(BW * N)
    t0 = 2 * ((rows % N) // 32)
    sk = skip[(rows // N) % nW]
    both = ((sk >> t0) & 1) & (((sk >> (t0 + 1)) & 1) | (16 * (t0 + 1) >= N))
    skipped = torch.from_numpy(both.astype(bool))
    assert torch.equal(part[~skipped], out[~skipped]) and bool((part[skipped] == 7.0).all())


def test_window_attention32_extremes_and_rescale(half):
    """Rows whose maximum grows by far more than the rescale threshold in the middle of the key range (a dominant key in a late
    block), rows that START masked (shifted windows: the first key blocks of some queries hold -100 only), and a key that dominates
    by > 80: finite, and within the rounding budget of the oracle."""
    g = rng(31)
    window, shift = (8, 7, 7), (4, 3, 3)
    lay = O.window_layout(16, 14, 14, window, shift)
    N, nW, nH = lay["N"], lay["nW"], 2
    q = torch.from_numpy(g.standard_normal((nW, nH, N, 32)).astype(np.float32)) * 0.6
    k = torch.from_numpy(g.standard_normal((nW, nH, N, 32)).astype(np.float32))
    q[:, 1, :, 0] = 16.0
    k[:, 1, :, 0] = 0.0
    k[:, 1, 300, 0] = 6.0                       # head 1: key 300 (block 9) wins every un-masked row by ~96
    k[:, 0, 40::57] *= 9.0                      # head 0: a few loud keys spread over the blocks
    q2, k, v = rnd(q * kernels.LOG2E, half), rnd(k, half), rnd(torch.from_numpy(g.standard_normal((nW, nH, N, 32)).astype(np.float32)), half)
    rpb = torch.from_numpy((0.5 * g.standard_normal((2535, nH))).astype(np.float32))
    rpb[:, 1] = 0.0          # head 1: biases 0 / -100 only — exact in the fp16 image (a +96 logit against a rounded -100.3 would make
                             # the test measure the image's rounding under cancellation, which the dense kernel shares)
    ref = O.attention_core(q2 / kernels.LOG2E, k, v, rpb, None, window, lay).reshape(nW * N, nH * 32)
    tok, center = _tok_table(lay, window)
    qkv = dev(torch.stack([q2, k, v]).permute(0, 2, 1, 3, 4).reshape(3, nH, nW * N, 32).contiguous(), half)
    image = kernels.attn_bias32(dev(torch.from_numpy(tok)), dev(rpb), None, center, nW, N, True)
    out = kernels.window_attention32(qkv, image, nW, N).float().cpu()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= 6.4 * EPS[half] + 2.0 ** -7


@pytest.mark.parametrize("dims", [(16, 14, 14), (24, 7, 7)])
def test_window_attention32_depth_split(dims, half):
    """dsplit_from: depth-split windows pass over the other half's 32-key blocks.  Their scores are the image's -100 and leave the
    exponential as zeros against any maximum the row's own half produces.  Windows that are not split and the first-half rows of split
    windows (their skipped blocks come LAST: exact zeros added) agree bit for bit; second-half rows start their running maximum at
    another block (the full launch rescales once, from the masked blocks' level), so their exponent arguments round differently in the
    last fp32 bit: equal to one 16-bit rounding of the output."""
    g = rng(sum(dims) + 7)
    window, shift = (8, 7, 7), (4, 3, 3)
    lay = O.window_layout(*dims, window, shift)
    N, nW, nH, B = lay["N"], lay["nW"], 3, 2
    BW = B * nW
    qkv = dev(rnd(torch.from_numpy(g.standard_normal((3, nH, BW * N, 32)).astype(np.float32)) * 0.7, half), half)
    rpb = torch.from_numpy((0.5 * g.standard_normal((2535, nH))).astype(np.float32))
    fpb = torch.from_numpy((0.5 * g.standard_normal((2535, nH))).astype(np.float32))
    tok, center = _tok_table(lay, window)
    image = kernels.attn_bias32(dev(torch.from_numpy(tok)), dev(rpb), dev(fpb), center, nW, N, True)
    full = kernels.window_attention32(qkv, image, nW, N)
    first = nW - nW // (-(-dims[0] // 8))
    split = kernels.window_attention32(qkv, image, nW, N, dsplit_from=first)
    rows = torch.arange(BW * N)
    same = (((rows // N) % nW) < first) | ((rows % N) < 192)
    assert torch.equal(split.cpu()[same], full.cpu()[same])
    assert (split.float() - full.float()).abs().max().item() <= 2.0 * EPS[half] * float(full.float().abs().max())
    assert not torch.equal(split, torch.zeros_like(split))
    with pytest.raises(RuntimeError):
        kernels.window_attention32(qkv[:, :, : BW * 98].contiguous(), image, nW, 98, dsplit_from=0)     # not the (8,7,7) window


@pytest.mark.parametrize("dims,shift", [((16, 14, 14), (0, 0, 0)), ((16, 14, 14), (4, 3, 3)), ((8, 21, 14), (0, 0, 0))])
def test_window_attention32_fused_qkv_projection(dims, shift, half):
    """The attention launch that computes its own q | k | v from the norm1 rows (the un-padded C = 96 stage) against the qkv GEMM (q scaled
    by head_dim^-0.5 * log2(e)) followed by the plain launch: the same fp32 accumulation over C and the same 16-bit rounding of
    q | k | v, a different MFMA shape — the outputs agree to the output's own rounding."""
    C = 96
    g = rng(C + sum(dims) + sum(shift) + 5)
    window = (8, 7, 7)
    lay = O.window_layout(*dims, window, shift)
    N, nW, nH = lay["N"], lay["nW"], C // 32
    B = 2
    BW = B * nW
    x = rnd(torch.from_numpy(g.standard_normal((BW * N, C)).astype(np.float32)), half)
    Wq = rnd(torch.from_numpy((g.standard_normal((3 * C, C)) / np.sqrt(C)).astype(np.float32)), half)
    bq = torch.from_numpy(0.3 * g.standard_normal(3 * C).astype(np.float32))
    scale = 32 ** -0.5 * kernels.LOG2E
    rpb = torch.from_numpy((0.5 * g.standard_normal((2535, nH))).astype(np.float32))
    tok, center = _tok_table(lay, window)
    use_mask = any(s > 0 for s in lay["ss"])
    n_types = nW if use_mask else nW // (-(-dims[0] // lay["ws"][0]))
    image = kernels.attn_bias32(dev(torch.from_numpy(tok))[: n_types * N], dev(rpb), None, center, n_types, N, use_mask)
    qkv = kernels.gemm(dev(x, half), dev(Wq, half), dev(bq), _abi.EPI_QKV_BF16, num_heads=nH, q_scale=scale)
    slabs = -(-dims[0] // 8)
    ds = nW - nW // slabs if use_mask else -1
    ref = kernels.window_attention32(qkv, image, nW, N, n_types, dsplit_from=ds).float().cpu()
    scratch = torch.full((1, nH, BW * N, 32), float("nan"), dtype=half, device=DEV)
    out = kernels.window_attention32(scratch, image, nW, N, n_types, dsplit_from=ds, x_ln=dev(x, half), w_qkv=dev(Wq, half),
                                     b_qkv=dev(bq), q_scale=scale).float().cpu()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= 3.0 * EPS[half] * max(1.0, ref.abs().max().item())
    assert (out - ref).abs().mean().item() <= 0.2 * EPS[half]
    assert (scratch[0].float() - qkv[0].float()).abs().max().item() <= 2.0 * EPS[half] * qkv[0].float().abs().max().item()
    with pytest.raises(RuntimeError, match="C = 96"):
        kernels.window_attention32(torch.empty(1, 6, BW * N, 32, dtype=half, device=DEV), image, nW, N, n_types,
                                   x_ln=torch.empty(BW * N, 192, dtype=half, device=DEV), w_qkv=torch.empty(576, 192, dtype=half, device=DEV),
                                   b_qkv=torch.empty(576, device=DEV), q_scale=scale)


def test_window_attention32_writes_padding_rows_itself(half):
    """Padded partitions: with ``pad_mask`` the kernel puts k | v = the qkv bias into the padding rows of its own K | V images and takes their
    q as zero, instead of reading rows a separate launch (kvq_qkv_fill_pad) wrote into the buffer.  Same output, bit for bit, on every row
    that is not a padding row — whatever the buffer holds in the padding rows (here: NaN)."""
    g = rng(77)
    dims, window = (8, 10, 9), (8, 7, 7)                        # H, W pad to 14: most windows hold padding rows
    lay = O.window_layout(*dims, window, (4, 3, 3))
    N, nW, nH, B = lay["N"], lay["nW"], 2, 2
    BW = B * nW
    C = 32 * nH
    pad = torch.from_numpy(lay["src"].reshape(nW, N) < 0)
    assert bool(pad.any()) and not bool(pad.all(1).any())
    bq = torch.from_numpy(0.4 * g.standard_normal(3 * C).astype(np.float32))
    qkv = rnd(torch.from_numpy(g.standard_normal((3, nH, BW, N, 32)).astype(np.float32)) * 0.7, half)
    padb = pad[None, None].expand(nH, B, nW, N).reshape(nH, BW, N)
    filled = qkv.clone()
    for which in (1, 2):                                        # what kvq_qkv_fill_pad writes: the 16-bit rounding of the bias
        bias = rnd(bq[which * C:(which + 1) * C].reshape(nH, 1, 1, 32), half).expand(nH, BW, N, 32)
        filled[which][padb] = bias[padb]
    filled[0][padb] = 0.0
    holes = qkv.clone()
    holes[:, padb] = float("nan")
    rpb = torch.from_numpy((0.5 * g.standard_normal((2535, nH))).astype(np.float32))
    tok, center = _tok_table(lay, window)
    image = kernels.attn_bias32(dev(torch.from_numpy(tok)), dev(rpb), None, center, nW, N, True)
    mask = np.zeros((nW, 13), np.uint32)
    for wv in range(nW):
        for r in np.nonzero(pad[wv].numpy())[0]:
            mask[wv, r >> 5] |= np.uint32(1) << np.uint32(r & 31)
    ref = kernels.window_attention32(dev(filled.reshape(3, nH, BW * N, 32), half), image, nW, N).float().cpu()
    out = kernels.window_attention32(dev(holes.reshape(3, nH, BW * N, 32), half), image, nW, N, b_qkv=dev(bq),
                                     pad_mask=dev(torch.from_numpy(mask.view(np.int32)))).float().cpu()
    keep = ~pad[None].expand(B, nW, N).reshape(-1)
    assert torch.isfinite(out[keep]).all() and torch.equal(out[keep], ref[keep])
    with pytest.raises(RuntimeError, match="b_qkv"):
        kernels.window_attention32(dev(holes.reshape(3, nH, BW * N, 32), half), image, nW, N, pad_mask=dev(torch.from_numpy(mask.view(np.int32))))


def test_window_attention_rejects_large_window():
    with pytest.raises(_abi.KvqError, match="unsupported"):
        kernels.window_attention(torch.zeros(3, 1, 512, 32, dtype=torch.float16, device=DEV),
                                 torch.zeros(512, 2, dtype=torch.int32, device=DEV),
                                 torch.zeros(10, 1, device=DEV), None, 0, 1, 512, False)


# ------------------------------------------------------------------------- embed / heads / sampler
@pytest.mark.parametrize("shape", [(2, 3, 8, 32, 32), (1, 3, 7, 30, 27)])
def test_patch_im2col(shape, half):
    g = rng(sum(shape))
    x = torch.from_numpy(g.standard_normal(shape).astype(np.float32))
    B, Cin, T, H, W = shape
    out = kernels.patch_im2col(dev(x), (2, 4, 4), out_dtype=half).float().cpu()
    xp = torch.nn.functional.pad(x, (0, (-W) % 4, 0, (-H) % 4, 0, (-T) % 2))
    D, Hh, Ww = xp.shape[2] // 2, xp.shape[3] // 4, xp.shape[4] // 4
    ref = xp.reshape(B, Cin, D, 2, Hh, 4, Ww, 4).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * D * Hh * Ww, -1)
    assert torch.equal(out, rnd(ref, half))


@pytest.mark.parametrize("shape,hidden", [((3, 768, 4, 7, 7), 64), ((2, 1024, 1, 5, 5), 64), ((1, 768, 2, 3, 3), 32), ((2, 96, 2, 4, 4), 64)])
def test_vqa_head_layouts(shape, hidden):
    """channels-first features take the VALU kernel, channels-last ones with 64 hidden units and C % 64 == 0 the fp32-MFMA
    kernel (ragged last 16-token tile included); both are fp32 arithmetic on the oracle's numbers."""
    g = rng(77 + shape[1])
    Cc = shape[1]
    feat = torch.from_numpy(g.standard_normal(shape).astype(np.float32))
    hw = synth.synth_vqa_head_weights(Cc, hidden, 5, "stress")
    ref = O.vqa_head(feat, hw)
    w = {k: dev(torch.from_numpy(v)) for k, v in hw.items()}
    args = (w["fc_hid.weight"].reshape(hidden, Cc), w["fc_hid.bias"], w["fc_last.weight"].reshape(-1), w["fc_last.bias"])
    out_cf = kernels.vqa_head(dev(feat), *args)                                         # (B,C,D,H,W) contiguous
    cl = dev(feat).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)          # channels-last view
    out_cl = kernels.vqa_head(cl, *args)
    assert (out_cf.cpu() - ref).abs().max().item() <= 2e-5
    assert (out_cl.cpu() - ref).abs().max().item() <= 2e-5
    # a clip's score does not depend on its neighbours in the batch (a 16-token tile may straddle two clips)
    one = kernels.vqa_head(cl[-1:], *args)
    assert torch.equal(one, out_cl[-1:])


@pytest.mark.parametrize("K,pool,hidden", [(1, True, 64), (3, False, 64), (5, True, 64), (4, False, 96)])
def test_vqa_head_pre_pool_and_classes(K, pool, hidden, golden):
    """VQAHead's pre_pool / num_class > 1 branches (head.py:61-62, :66-67) through the module with the reference's constructor
    arguments: against the oracle on both feature layouts, and (64 hidden units) against the reference's own stored outputs."""
    from kvq_amd.models.head import VQAHead
    g = rng(77)
    feat = torch.from_numpy(g.standard_normal((3, 768, 4, 7, 7)).astype(np.float32))
    hw = synth.synth_vqa_head_weights(768, hidden, 6, "stress", num_class=K)
    ref = O.vqa_head(feat, hw, pre_pool=pool)
    head = VQAHead(in_channels=768, hidden_channels=hidden, num_class=K, pre_pool=pool).eval()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in hw.items()})
    head = head.to("cuda")
    out_cf = head(dev(feat))
    out_cl = head(dev(feat).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3))
    assert out_cf.shape == (3, K)
    assert (out_cf.cpu() - ref).abs().max().item() <= 2e-5 and (out_cl.cpu() - ref).abs().max().item() <= 2e-5
    tag = {(1, True): "pool", (3, False): "k3", (5, True): "k5pool"}.get((K, pool))
    if tag and hidden == 64:
        assert np.abs(out_cf.cpu().numpy() - golden("heads.npz")[f"vqa/{tag}/score"]).max() <= 2e-5


def test_simple_vqa_head():
    g = rng(78)
    feat = torch.from_numpy(g.standard_normal((2, 8, 9472)).astype(np.float32))
    hw = synth.synth_simple_head_weights(9472, 128, 5, "stress")
    ref = O.simple_vqa_head(feat, hw)
    w = {k: dev(torch.from_numpy(v)) for k, v in hw.items()}
    out = kernels.simple_vqa_head(dev(feat), w["quality.0.weight"], w["quality.0.bias"],
                                  w["quality.1.weight"].reshape(-1), w["quality.1.bias"])
    assert (out.cpu() - ref).abs().max().item() <= 5e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("T,H,W,Fh,Fw,fs,al", [(16, 400, 720, 9, 9, 32, 8), (32, 270, 480, 7, 7, 32, 8),
                                               (8, 224, 230, 7, 7, 32, 4)])
def test_fragment_gather_bit_exact(T, H, W, Fh, Fw, fs, al):
    g = rng(T + H)
    video = g.integers(0, 256, size=(3, T, H, W), dtype=np.uint8)
    rh, rw = synth.synth_fragment_offsets(T, T, H, W, Fh, Fw, fs, fs, al)
    ref_raw = SO.spatial_fragments(video.astype(np.float32), rh, rw, Fh, Fw, fs, fs, al)
    ref = SO.normalize(ref_raw, synth.KVQ_MEAN, synth.KVQ_STD)
    hoff = rh + SO.fragment_grid(H, Fh, fs)[:, None, None].astype(np.int32)
    woff = rw + SO.fragment_grid(W, Fw, fs)[None, :, None].astype(np.int32)
    args = (dev(torch.from_numpy(hoff)), dev(torch.from_numpy(woff)), Fh, Fw, fs, fs, al)
    out_u8 = kernels.fragment_gather(dev(torch.from_numpy(video)), *args, mean=synth.KVQ_MEAN, std=synth.KVQ_STD)
    out_f32 = kernels.fragment_gather(dev(torch.from_numpy(video.astype(np.float32))), *args, mean=synth.KVQ_MEAN,
                                      std=synth.KVQ_STD)
    raw = kernels.fragment_gather(dev(torch.from_numpy(video)), *args)
    assert np.array_equal(raw.cpu().numpy(), ref_raw)
    assert np.array_equal(out_u8.cpu().numpy(), ref)          # (v-mean)/std in fp32: bit-exact
    assert np.array_equal(out_f32.cpu().numpy(), ref)


def test_fragment_gather_reference_assert():
    v = torch.zeros(3, 10, 224, 224, dtype=torch.uint8, device=DEV)
    z = torch.zeros(7, 7, 1, dtype=torch.int32, device=DEV)
    with pytest.raises(AssertionError, match="Please provide match vclip and align index"):
        kernels.fragment_gather(v, z, z, 7, 7, 32, 32, 8)


# ------------------------------------------------------------------ fused proj + norm2 + Mlp (tail.hip)
def _tail_reference(A, x, wts, half, lay, B, dims):
    """fp32 composition with the 16-bit operand roundings the launch applies (norm2 output, GELU output)."""
    Wp, bp, g2, b2n, W1, b1, W2, b2 = wts
    y = A @ Wp.t() + bp
    if lay is not None:
        D, H, W = dims
        L, C = D * H * W, x.shape[1]
        y = O.scatter_windows(y.reshape(B, -1, C), lay, B, D, H, W).reshape(B * L, C)
    x1 = x + y
    h = rnd(torch.nn.functional.layer_norm(x1, (x.shape[1],), g2, b2n), half)
    hid = rnd(torch.nn.functional.gelu(h @ W1.t() + b1), half)
    return x1 + hid @ W2.t() + b2


@pytest.mark.parametrize("C,dims,shift,nxt_shift", [(96, (8, 14, 14), (0, 0, 0), (4, 3, 3)),
                                                   (96, (4, 10, 9), (4, 3, 3), None),      # padded: no emission
                                                   (128, (8, 7, 14), (4, 3, 3), (0, 0, 0)),
                                                   (192, (8, 14, 7), (0, 0, 0), (4, 3, 3)),
                                                   (192, (3, 5, 7), (0, 0, 0), None),      # clamped window, ragged tile
                                                   (384, (8, 14, 14), (0, 0, 0), (4, 3, 3)),   # wide rows: csrc/tailmm.hip
                                                   (384, (8, 7, 7), (0, 0, 0), (0, 0, 0)),
                                                   (384, (4, 10, 9), (4, 3, 3), None),
                                                   (256, (8, 14, 7), (4, 3, 3), (0, 0, 0)),    # stage 1 of Swin-B: tailmm, CF = 2
                                                   (256, (4, 10, 9), (4, 3, 3), None),
                                                   (512, (8, 14, 7), (0, 0, 0), (4, 3, 3)),    # stage 2 of Swin-B: tailmm, CF = 4
                                                   (512, (4, 10, 9), (4, 3, 3), None),
                                                   (768, (16, 7, 7), (0, 0, 0), (4, 0, 0))])
def test_block_tail(C, dims, shift, nxt_shift, half):
    g = rng(C + sum(dims))
    D, H, W = dims
    B, hidden = 2, 4 * C
    lay = O.window_layout(D, H, W, (8, 7, 7), shift)
    Lp, L = lay["nW"] * lay["N"], D * H * W
    t = lambda *s, sc=1.0: torch.from_numpy((g.standard_normal(s) * sc).astype(np.float32))
    A = rnd(t(B * Lp, C), half)
    x = t(B * L, C, sc=2.0)
    Wp, W1, W2 = rnd(t(C, C, sc=0.15), half), rnd(t(hidden, C, sc=0.15), half), rnd(t(C, hidden, sc=0.08), half)
    bp, b1, b2 = t(C, sc=0.3), t(hidden, sc=0.3), t(C, sc=0.3)
    g2, b2n = 1 + 0.2 * t(C), 0.2 * t(C)
    wts = (Wp, bp, g2, b2n, W1, b1, W2, b2)
    pack = kernels.block_tail_pack(dev(Wp, half), dev(bp), dev(g2), dev(b2n), dev(W1, half), dev(b1), dev(W2, half), dev(b2))
    ref = _tail_reference(A, x, wts, half, lay, B, dims)
    xd = dev(x.clone())
    kw = {}
    if nxt_shift is not None:
        lay2 = O.window_layout(D, H, W, (8, 7, 7), nxt_shift)
        assert (lay2["src"] >= 0).all()
        dst = np.empty(L, np.int32)
        dst[lay2["src"]] = np.arange(L, dtype=np.int32)
        gn, bn = 1 + 0.2 * t(C), 0.2 * t(C)
        kw = dict(next_norm=(dev(gn), dev(bn)), next_dst=dev(torch.from_numpy(dst)), next_rows=L)
    out_ln = kernels.block_tail(dev(A, half), xd, pack, hidden, scatter_map=dev(torch.from_numpy(lay["src"].astype(np.int32))),
                                map_rows=Lp, out_rows=L, **kw)
    got = xd.cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 6 * EPS[half] * scale + 1e-4, ((got - ref).abs().max().item(), scale)
    if Lp != L:
        # padded windows: the launch that walks the TOKENS (attn_gather = the inverse of the scatter map) instead of the window
        # rows computes every real token exactly as before
        src = lay["src"].astype(np.int64)
        inv = np.zeros(L, np.int32)
        inv[src[src >= 0]] = np.nonzero(src >= 0)[0].astype(np.int32)
        xg = dev(x.clone())
        kernels.block_tail(dev(A, half), xg, pack, hidden, attn_gather=dev(torch.from_numpy(inv)), map_rows=Lp, out_rows=L)
        assert torch.equal(xg.cpu(), got)
    if nxt_shift is not None:
        ln = torch.nn.functional.layer_norm(got, (C,), gn, bn).reshape(B, D, H, W, C)
        ref_ln = O.gather_windows(ln, lay2).reshape(-1, C)
        d = (out_ln.float().cpu() - ref_ln).abs().max().item()
        assert d <= 2 * EPS[half] * ref_ln.abs().max().item() + 1e-5, d


@pytest.mark.parametrize("C,dims,shift,nxt_shift", [(384, (8, 14, 14), (0, 0, 0), (4, 3, 3)), (384, (16, 14, 14), (4, 3, 3), (0, 0, 0)),
                                                    (256, (8, 7, 7), (0, 0, 0), (0, 0, 0)), (512, (8, 14, 7), (0, 0, 0), (4, 3, 0)),
                                                    (192, (8, 14, 14), (0, 0, 0), (4, 3, 3)), (192, (16, 28, 28), (4, 3, 3), (0, 0, 0)),
                                                    (128, (8, 14, 7), (0, 0, 0), (4, 3, 0)),
                                                    (768, (16, 7, 7), (0, 0, 0), (4, 0, 0))])
def test_block_tail_emits_next_qkv(C, dims, shift, nxt_shift, half):
    """The fused tail (C = 128 / 192: token per lane; C = 256 / 384 / 512: feature-sliced) writing the NEXT block's q | k | v itself (swin_backbone.py:252-260 of block b + 1:
    norm1 -> qkv Linear -> head split, q scaled) against LayerNorm + the qkv GEMM launch on the launch's own residual output: the
    same 16-bit operands (norm1 rows, weights), so the two agree to the rounding of one 16-bit result; and x is what the launch
    without the emission leaves."""
    g = rng(C + sum(dims) + 5)
    D, H, W = dims
    B, hidden, nH = 2, 4 * C, C // 32
    lay = O.window_layout(D, H, W, (8, 7, 7), shift)
    lay2 = O.window_layout(D, H, W, (8, 7, 7), nxt_shift)
    Lp, L = lay["nW"] * lay["N"], D * H * W
    assert Lp == L and (lay2["src"] >= 0).all()
    t = lambda *s, sc=1.0: torch.from_numpy((g.standard_normal(s) * sc).astype(np.float32))
    A, x = rnd(t(B * Lp, C), half), t(B * L, C, sc=2.0)
    Wp, W1, W2 = rnd(t(C, C, sc=0.15), half), rnd(t(hidden, C, sc=0.15), half), rnd(t(C, hidden, sc=0.08), half)
    bp, b1, b2, g2, b2n = t(C, sc=0.3), t(hidden, sc=0.3), t(C, sc=0.3), 1 + 0.2 * t(C), 0.2 * t(C)
    Wq, bq = rnd(t(3 * C, C, sc=0.1), half), t(3 * C, sc=0.3)
    gn, bn = 1 + 0.2 * t(C), 0.2 * t(C)
    qs = 0.25
    pack = kernels.block_tail_pack(dev(Wp, half), dev(bp), dev(g2), dev(b2n), dev(W1, half), dev(b1), dev(W2, half), dev(b2))
    qpack = kernels.block_tail_qkv_pack(dev(Wq, half), hidden)
    assert qpack is not None
    dst = np.empty(L, np.int32)
    dst[lay2["src"]] = np.arange(L, dtype=np.int32)
    common = dict(scatter_map=dev(torch.from_numpy(lay["src"].astype(np.int32))), map_rows=Lp, out_rows=L,
                  next_norm=(dev(gn), dev(bn)), next_dst=dev(torch.from_numpy(dst)), next_rows=L)
    x_ln, x_qkv = dev(x.clone()), dev(x.clone())
    ln_rows = kernels.block_tail(dev(A, half), x_ln, pack, hidden, **common)
    qkv = kernels.block_tail(dev(A, half), x_qkv, pack, hidden, next_qkv=(qpack, dev(bq), qs), **common)
    assert torch.equal(x_ln, x_qkv)
    assert qkv.shape == (3, nH, B * L, 32)
    ref = kernels.gemm(ln_rows, dev(Wq, half), dev(bq), _abi.EPI_QKV_BF16, num_heads=nH, q_scale=qs).reshape(3, nH, B * L, 32)
    d = (qkv.float() - ref.float()).abs().max().item()
    assert d <= 2 * EPS[half] * ref.float().abs().max().item() + 1e-5, d
    # and against the fp32 composition on the launch's residual output
    ln = torch.nn.functional.layer_norm(x_qkv.cpu(), (C,), gn, bn).reshape(B, D, H, W, C)
    rows = rnd(O.gather_windows(ln, lay2).reshape(-1, C), half)
    full = rows @ Wq.t() + bq
    full[:, :C] *= qs
    want = full.reshape(B * L, 3, nH, 32).permute(1, 2, 0, 3)
    assert (qkv.float().cpu() - want).abs().max().item() <= 6 * EPS[half] * want.abs().max().item() + 1e-4


def test_block_tail_identity_map_and_unsupported(half):
    g = rng(77)
    C, hidden, M = 96, 384, 333
    t = lambda *s, sc=1.0: torch.from_numpy((g.standard_normal(s) * sc).astype(np.float32))
    A, x = rnd(t(M, C), half), t(M, C)
    Wp, W1, W2 = rnd(t(C, C, sc=0.15), half), rnd(t(hidden, C, sc=0.15), half), rnd(t(C, hidden, sc=0.08), half)
    bp, b1, b2, g2, b2n = t(C), t(hidden), t(C), 1 + 0.1 * t(C), 0.1 * t(C)
    pack = kernels.block_tail_pack(dev(Wp, half), dev(bp), dev(g2), dev(b2n), dev(W1, half), dev(b1), dev(W2, half), dev(b2))
    xd = dev(x.clone())
    kernels.block_tail(dev(A, half), xd, pack, hidden)
    ref = _tail_reference(A, x, (Wp, bp, g2, b2n, W1, b1, W2, b2), half, None, 1, None)
    assert (xd.cpu() - ref).abs().max().item() <= 6 * EPS[half] * ref.abs().max().item() + 1e-4
    assert _abi.lib().kvq_block_tail_pack_bytes(1024, 4096) == 0
    with pytest.raises(_abi.KvqError, match="unsupported"):
        kernels.block_tail_pack(*(dev(torch.zeros(s), half if len(s) == 2 else None) for s in
                                  [(1024, 1024), (1024,), (1024,), (1024,), (4096, 1024), (4096,), (1024, 4096), (1024,)]))


# ------------------------------------------------------------------ fused PatchEmbed3D (embed.hip)
@pytest.mark.parametrize("shape,E,with_ln,emit", [((2, 3, 8, 56, 56), 96, True, True), ((1, 3, 6, 28, 44), 96, True, False),
                                                 ((2, 3, 4, 28, 56), 128, False, True)])
def test_patch_embed_fused(shape, E, with_ln, emit, half):
    """conv3d(k=s=(2,4,4)) + bias + LayerNorm [+ norm1 of the first block in window order] against torch fp32 with the
    operand roundings of the launch (pixels and weights to 16 bits)."""
    g = rng(sum(shape) + E)
    B, Cin, T, H, W = shape
    x = torch.from_numpy((g.standard_normal(shape) * 1.5).astype(np.float32))
    w = rnd(torch.from_numpy((g.standard_normal((E, Cin, 2, 4, 4)) * 0.1).astype(np.float32)), half)
    b = torch.from_numpy((g.standard_normal(E) * 0.2).astype(np.float32))
    lw = torch.from_numpy((1 + 0.2 * g.standard_normal(E)).astype(np.float32)) if with_ln else None
    lb = torch.from_numpy((0.2 * g.standard_normal(E)).astype(np.float32)) if with_ln else None
    y = torch.nn.functional.conv3d(rnd(x, half), w, b, stride=(2, 4, 4)).permute(0, 2, 3, 4, 1)     # (B,D,H0,W0,E)
    if with_ln:
        y = torch.nn.functional.layer_norm(y, (E,), lw, lb)
    D, H0, W0 = y.shape[1:4]
    kw = {}
    if emit:
        lay = O.window_layout(D, H0, W0, (8, 7, 7), (0, 0, 0))
        if (lay["src"] < 0).any():
            pytest.skip("padded layout: the plan keeps the separate norm1 launch")
        L = D * H0 * W0
        dst = np.empty(L, np.int32)
        dst[lay["src"]] = np.arange(L, dtype=np.int32)
        gn = torch.from_numpy((1 + 0.2 * g.standard_normal(E)).astype(np.float32))
        bn = torch.from_numpy((0.2 * g.standard_normal(E)).astype(np.float32))
        kw = dict(next_norm=(dev(gn), dev(bn)), next_dst=dev(torch.from_numpy(dst)), next_rows=L)
    out, nxt = kernels.patch_embed(dev(x), dev(w.reshape(E, -1), half), dev(b), None if lw is None else dev(lw),
                                   None if lb is None else dev(lb), (2, 4, 4), **kw)
    ref = y.reshape(-1, E)
    assert (out.cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    if emit:
        ln = torch.nn.functional.layer_norm(out.cpu(), (E,), gn, bn).reshape(B, D, H0, W0, E)
        ref_ln = O.gather_windows(ln, lay).reshape(-1, E)
        assert (nxt.float().cpu() - ref_ln).abs().max().item() <= 2 * EPS[half] * ref_ln.abs().max().item() + 1e-5


def _fragment_source(seed, n_clips, T=16, Hs=270, Ws=480, grid=7, fs=32, aligned=8, normalise=True):
    """random uint8 clips + the sampler's draws (grid origin + offset inside the cell, fusion_datasets.py:64-98)"""
    g = torch.Generator().manual_seed(seed)
    vids = [torch.randint(0, 256, (3, T, Hs, Ws), dtype=torch.uint8, generator=g).to(DEV) for _ in range(n_clips)]
    gh = torch.tensor([min(Hs // grid * i, Hs - fs) for i in range(grid)]).view(grid, 1, 1)
    gw = torch.tensor([min(Ws // grid * i, Ws - fs) for i in range(grid)]).view(1, grid, 1)
    hs = [(torch.randint(max(1, Hs // grid - fs), (grid, grid, T // aligned), generator=g) + gh).int().to(DEV) for _ in range(n_clips)]
    ws = [(torch.randint(max(1, Ws // grid - fs), (grid, grid, T // aligned), generator=g) + gw).int().to(DEV) for _ in range(n_clips)]
    mean, std = ((123.675, 116.28, 103.53), (58.395, 57.12, 57.375)) if normalise else (None, None)
    return kernels.FragmentSource(vids, hs, ws, grid, grid, fs, fs, aligned, mean=mean, std=std)


@pytest.mark.parametrize("normalise", [True, False])
@pytest.mark.parametrize("emit", [False, True])
def test_patch_embed_reads_through_the_sampler(normalise, emit, half):
    """The embedding launch fed by a FragmentSource (uint8 frames + the sampler's patch origins: K1 fused into its operand
    read) against the same launch on the materialised fp32 clip (kvq_fragment_gather, pinned to get_spatial_fragments by
    test_fragment_gather_*): the 16-bit operands are the same numbers, so every output bit is."""
    src = _fragment_source(11 + emit, 3, normalise=normalise)
    assert src.shape == (3, 3, 16, 224, 224)
    g = rng(5)
    E = 96
    w = dev(torch.from_numpy((g.standard_normal((E, 96)) * 0.1).astype(np.float32)), half)
    b = dev(torch.from_numpy((g.standard_normal(E) * 0.2).astype(np.float32)))
    lw = dev(torch.from_numpy((1 + 0.2 * g.standard_normal(E)).astype(np.float32)))
    lb = dev(torch.from_numpy((0.2 * g.standard_normal(E)).astype(np.float32)))
    kw = {}
    if emit:
        lay = O.window_layout(8, 56, 56, (8, 7, 7), (0, 0, 0))
        L = 8 * 56 * 56
        dst = np.empty(L, np.int32)
        dst[lay["src"]] = np.arange(L, dtype=np.int32)
        kw = dict(next_norm=(lw, lb), next_dst=dev(torch.from_numpy(dst)), next_rows=L)
    clip = src.materialise()
    if not normalise:                                   # raw pixel values: exactly the bytes
        v, h, wo = src.videos[1], src.hoffs[1].cpu(), src.woffs[1].cpu()
        assert torch.equal(clip[1, :, 9, 32:64, 64:96].cpu(),
                           v[:, 9, h[1, 2, 1]:h[1, 2, 1] + 32, wo[1, 2, 1]:wo[1, 2, 1] + 32].float().cpu())
    out_a, nxt_a = kernels.patch_embed(clip, w, b, lw, lb, (2, 4, 4), **kw)
    out_b, nxt_b = kernels.patch_embed(src, w, b, lw, lb, (2, 4, 4), **kw)
    assert torch.equal(out_a, out_b)
    if emit:
        assert torch.equal(nxt_a, nxt_b)


@pytest.mark.parametrize("geom", [dict(T=8, Hs=300, Ws=420, grid=8, fs=32, aligned=8),      # Swin-B's 256 x 256 canvas, E = 128
                                  dict(T=4, Hs=90, Ws=130, grid=5, fs=16, aligned=2),      # 16-pixel mini-patches, frame groups of 2
                                  dict(T=6, Hs=64, Ws=64, grid=2, fs=32, aligned=1),       # the source IS the canvas; a draw per frame
                                  dict(T=16, Hs=80, Ws=100, grid=2, fs=32, aligned=4)])    # cut in two along time: clips as views
def test_patch_embed_reads_through_the_sampler_geometries(geom, half):
    """other grids / mini-patch sizes / frame groups, E = 128 (four channel tiles), clips that are frame runs of a longer video"""
    E = 128
    src = _fragment_source(41 + geom["grid"], 2, **geom)
    g = rng(6)
    w = dev(torch.from_numpy((g.standard_normal((E, 96)) * 0.1).astype(np.float32)), half)
    b = dev(torch.from_numpy((g.standard_normal(E) * 0.2).astype(np.float32)))
    lw = dev(torch.from_numpy((1 + 0.2 * g.standard_normal(E)).astype(np.float32)))
    lb = dev(torch.from_numpy((0.2 * g.standard_normal(E)).astype(np.float32)))
    L0 = (geom["T"] // 2) * (geom["grid"] * geom["fs"] // 4) ** 2
    if L0 % 32:
        assert src.c_struct() is not None
        with pytest.raises(_abi.KvqError, match="does not fit the fused read"):
            kernels.patch_embed(src, w, b, lw, lb, (2, 4, 4))
        return
    out_a, _ = kernels.patch_embed(src.materialise(), w, b, lw, lb, (2, 4, 4))
    out_b, _ = kernels.patch_embed(src, w, b, lw, lb, (2, 4, 4))
    assert torch.equal(out_a, out_b)
    if geom["T"] % (2 * geom["aligned"]) == 0:          # the same clips cut in two along time: views, channel stride of the parent
        halves = src.split_clips(2)
        assert halves.shape[0] == 4 and halves.shape[2] == geom["T"] // 2 and not halves.videos[1].is_contiguous()
        th = geom["T"] // 2
        if th % 2 == 0 and ((th // 2) * (geom["grid"] * geom["fs"] // 4) ** 2) % 32 == 0:
            o_view, _ = kernels.patch_embed(halves, w, b, lw, lb, (2, 4, 4))
            o_mat, _ = kernels.patch_embed(halves.materialise(), w, b, lw, lb, (2, 4, 4))
            assert torch.equal(o_view, o_mat)


@pytest.mark.parametrize("f32", [False, True])
def test_fragment_gather_batch_equals_per_clip_gather(f32):
    """FragmentSource.materialise() is ONE launch (kvq_fragment_gather_batch): bit-equal to kvq_fragment_gather clip by clip
    (which is pinned to get_spatial_fragments), for uint8 and fp32 frames, contiguous clips and frame-run views of a longer video."""
    src = _fragment_source(21, 3, T=16, Hs=150, Ws=190, grid=4, fs=32, aligned=4)
    if f32:
        src = kernels.FragmentSource([v.float() for v in src.videos], src.hoffs, src.woffs, *src.geometry, mean=src.mean, std=src.std)
    for s_ in (src, src.split_clips(2)):
        per_clip = torch.stack([kernels.fragment_gather(v.contiguous(), h, w, *s_.geometry, mean=s_.mean, std=s_.std)
                                for v, h, w in zip(s_.videos, s_.hoffs, s_.woffs)])
        assert torch.equal(s_.materialise(), per_clip)
    raw = kernels.FragmentSource(src.videos, src.hoffs, src.woffs, *src.geometry)          # no normalisation: the pixel values
    assert torch.equal(raw.materialise()[2, 1, 5, :32, :32].cpu(),
                       src.videos[2][1, 5, int(src.hoffs[2][0, 0, 1]):int(src.hoffs[2][0, 0, 1]) + 32,
                                     int(src.woffs[2][0, 0, 1]):int(src.woffs[2][0, 0, 1]) + 32].float().cpu())


def test_patch_embed_fragment_source_guards():
    """no fused read: mini-patches that do not hold whole 4 x 4 patches, fp32 frames, a source smaller than the canvas"""
    src = _fragment_source(3, 1, T=8, Hs=100, Ws=120, grid=2, fs=32, aligned=8)
    f = src.c_struct()
    lib = _abi.lib()
    assert lib.kvq_patch_embed_fragments_supported(f, 1, 3, 2, 8, 64, 64) == 1
    assert lib.kvq_patch_embed_fragments_supported(f, 2, 3, 2, 8, 64, 64) == 0          # batch != clips
    assert lib.kvq_patch_embed_fragments_supported(f, 1, 3, 2, 8, 64, 96) == 0          # canvas != grid * fs
    f.fs_h, f.Fh = 16, 4
    assert lib.kvq_patch_embed_fragments_supported(f, 1, 3, 2, 8, 64, 64) == 1
    f.fs_h, f.Fh = 2, 32
    assert lib.kvq_patch_embed_fragments_supported(f, 1, 3, 2, 8, 64, 64) == 0          # patch rows span mini-patches
    f = src.c_struct()
    f.Hs = 48
    assert lib.kvq_patch_embed_fragments_supported(f, 1, 3, 2, 8, 64, 64) == 0          # source < canvas
    f32 = kernels.FragmentSource([v.float() for v in src.videos], src.hoffs, src.woffs, *src.geometry)
    assert f32.c_struct() is None and f32.materialise().shape == (1, 3, 8, 64, 64)
    with pytest.raises(_abi.KvqError, match="no fused read"):
        kernels.patch_embed(f32, torch.zeros(96, 96, dtype=torch.float16, device=DEV), torch.zeros(96, device=DEV), None, None,
                            (2, 4, 4))


def test_patch_embed_fused_rejects_padded_clip():
    with pytest.raises(_abi.KvqError, match="unsupported"):
        kernels.patch_embed(torch.zeros(1, 3, 7, 30, 27, device=DEV), torch.zeros(96, 96, dtype=torch.float16, device=DEV),
                            torch.zeros(96, device=DEV), None, None, (2, 4, 4))


def test_fp16_narrowing_saturates_instead_of_overflowing():
    """Values past the fp16 range clamp to +/-65504 in every 16-bit epilogue (MODE.FP16_OVFL), never inf."""
    A = torch.full((64, 32), 200.0, dtype=torch.float16, device=DEV)
    W = torch.full((32, 32), 200.0, dtype=torch.float16, device=DEV)
    W[1] = -200.0
    out = kernels.gemm(A, W, None, _abi.EPI_BIAS_BF16)                 # 32 * 200 * 200 = 1.28e6 > 65504
    assert torch.isfinite(out).all()
    assert (out[:, 0] == 65504).all() and (out[:, 1] == -65504).all()
    x = torch.zeros(8, 96, device=DEV)
    x[:, 0] = 1e6
    y = kernels.layernorm_rows(x, torch.full((96,), 1e5, device=DEV), torch.zeros(96, device=DEV), out_dtype=torch.float16)
    assert torch.isfinite(y).all() and (y[:, 0] == 65504).all()


# ------------------------------------------------------------------ implicit-GEMM convolution (gemm.hip, IMPL)
@pytest.mark.parametrize("shape,Cout,k,stride,pad", [
    ((2, 1, 28, 28, 64), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # ResNet 3x3
    ((2, 1, 29, 31, 128), 136, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # strided, odd sizes, ragged M / N tiles
    ((1, 8, 14, 14, 32), 32, (3, 1, 1), (1, 1, 1), (1, 0, 0)),       # SlowFast temporal conv
    ((1, 6, 9, 9, 8), 16, (3, 3, 3), (1, 1, 1), (1, 1, 1)),          # K = 216 -> padded to 224; C = 8 (fast pathway)
    ((1, 4, 7, 7, 8), 8, (1, 1, 1), (1, 1, 1), (0, 0, 0)),           # pointwise with C % 32 != 0: K = 8 -> 32
    ((1, 16, 8, 8, 16), 64, (5, 1, 1), (4, 1, 1), (2, 0, 0)),        # lateral time-strided conv
])
@pytest.mark.parametrize("relu,with_resid", [(True, True), (False, False)])
def test_conv_implicit_equals_im2col_gemm(shape, Cout, k, stride, pad, relu, with_resid, half):
    """kvq_conv_implicit fetches the same operand values in the same K order as kvq_im2col_nd + kvq_gemm_bf16:
    the outputs are bit-identical, and both match an fp32 F.conv3d of the rounded operands."""
    g = rng(sum(shape) + Cout)
    B, D, H, W, Cc = shape
    x = rnd(torch.from_numpy(g.standard_normal(shape).astype(np.float32)), half)
    K = k[0] * k[1] * k[2] * Cc
    kpad = -(-K // 32) * 32
    w5 = rnd(torch.from_numpy((g.standard_normal((Cout, Cc) + k) / np.sqrt(K)).astype(np.float32)), half)   # (N, C, kd, kh, kw)
    wk = torch.zeros(Cout, kpad)
    wk[:, :K] = w5.permute(0, 2, 3, 4, 1).reshape(Cout, K)                                                 # (kd, kh, kw, c) columns
    bias = torch.from_numpy(g.standard_normal(Cout).astype(np.float32))
    xd, wd, bd = dev(x, half), dev(wk, half), dev(bias)
    cols, (Do, Ho, Wo) = kernels.im2col_nd(xd, (B, Cc, D, H, W), (D * H * W * Cc, 1, H * W * Cc, W * Cc, Cc), k, stride, pad,
                                           xd.dtype, kpad)
    M = B * Do * Ho * Wo
    resid = torch.from_numpy(g.standard_normal((M, Cout)).astype(np.float32))
    kw = dict(resid_f32=dev(resid), want_f32=True) if with_resid else {}
    ref = kernels.conv_gemm(cols, wd, bd, relu, **kw)
    got = kernels.conv_implicit(xd, wd, bd, k, stride, pad, relu, **kw)
    if with_resid:
        assert torch.equal(got[1], ref[1])
        got, ref = got[0], ref[0]
    assert got.shape == (B, Do, Ho, Wo, Cout) and torch.equal(got.reshape(M, Cout), ref)
    if not relu:        # the fp32 projection-shortcut form
        assert torch.equal(kernels.conv_implicit(xd, wd, bd, k, stride, pad, False, store_f32=True),
                           kernels.gemm(cols, wd, bd, _abi.EPI_STORE_F32))
    f = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3), w5, bias, stride, pad).permute(0, 2, 3, 4, 1).reshape(M, Cout)
    if with_resid:
        f = f + resid
    f = torch.relu(f) if relu else f
    assert (got.reshape(M, Cout).float().cpu() - f).abs().max().item() <= 4 * EPS[half] * max(1.0, f.abs().max().item())


def test_conv_implicit_rejects_bad_shapes(half):
    x = dev(torch.zeros(1, 1, 4, 4, 12), half)           # C % 8 != 0
    with pytest.raises(_abi.KvqError, match="C % 8"):
        kernels.conv_implicit(x, dev(torch.zeros(8, 128), half), dev(torch.zeros(8)), (1, 3, 3), (1, 1, 1), (0, 1, 1), True)


# ------------------------------------------------------------------------------- the fp16 residual stream of stages 0-1 (round 6, ABI 31)
@pytest.mark.parametrize("C,dims,shift,nxt_shift,qkv", [(96, (8, 14, 14), (0, 0, 0), (4, 3, 3), False), (96, (8, 14, 7), (4, 3, 3), None, False),
                                                        (192, (8, 14, 14), (0, 0, 0), (4, 3, 3), True), (192, (8, 7, 7), (0, 0, 0), None, False),
                                                        (128, (8, 14, 7), (0, 0, 0), (4, 3, 0), True),
                                                        (384, (8, 14, 14), (0, 0, 0), (4, 3, 3), True), (384, (8, 7, 7), (4, 3, 3), None, False),      # csrc/tailmm.hip
                                                        (256, (8, 14, 7), (0, 0, 0), (4, 3, 0), True), (768, (8, 7, 7), (0, 0, 0), (0, 0, 0), True)])
def test_block_tail_fp16_residual_stream(C, dims, shift, nxt_shift, qkv, half):
    """``x_f16``: the launch reads and writes the residual stream as fp16 rows (token-per-lane tails and the wide feature-sliced ones).  Its arithmetic is the fp32-stream launch's on the
    widened input: the result is that launch's result on fp32(fp16(x)) rounded ONCE to fp16 — bit for bit — and the emitted norm1 rows / next
    q | k | v (computed from the un-rounded accumulators in both forms) are identical."""
    g = rng(7 * C + sum(dims))
    D, H, W = dims
    B, hidden = 2, 4 * C
    lay = O.window_layout(D, H, W, (8, 7, 7), shift)
    Lp, L = lay["nW"] * lay["N"], D * H * W
    t = lambda *s, sc=1.0: torch.from_numpy((g.standard_normal(s) * sc).astype(np.float32))      # noqa: E731
    A = rnd(t(B * Lp, C), half)
    x16 = t(B * L, C, sc=2.0).to(torch.float16)
    x16[0, :4] = torch.tensor([70000.0, -70000.0, 65504.0, 1e-7]).to(torch.float16)               # inf, -inf (a saturated producer never writes them), max, a subnormal
    x16[0, :2] = torch.tensor([60000.0, -60000.0]).to(torch.float16)
    Wp, W1, W2 = rnd(t(C, C, sc=0.15), half), rnd(t(hidden, C, sc=0.15), half), rnd(t(C, hidden, sc=0.08), half)
    bp, b1, b2, g2, b2n = t(C, sc=0.3), t(hidden, sc=0.3), t(C, sc=0.3), 1 + 0.2 * t(C), 0.2 * t(C)
    pack = kernels.block_tail_pack(dev(Wp, half), dev(bp), dev(g2), dev(b2n), dev(W1, half), dev(b1), dev(W2, half), dev(b2))
    kw = {}
    if nxt_shift is not None:
        lay2 = O.window_layout(D, H, W, (8, 7, 7), nxt_shift)
        dst = np.empty(L, np.int32)
        dst[lay2["src"]] = np.arange(L, dtype=np.int32)
        kw = dict(next_norm=(dev(1 + 0.2 * t(C)), dev(0.2 * t(C))), next_dst=dev(torch.from_numpy(dst)), next_rows=L)
        if qkv:
            wq = rnd(t(3 * C, C, sc=0.1), half)
            kw["next_qkv"] = (kernels.block_tail_qkv_pack(dev(wq, half), hidden), dev(t(3 * C, sc=0.2)), 0.25)
    smap = dev(torch.from_numpy(lay["src"].astype(np.int32)))
    x32 = dev(x16.float())
    n32 = kernels.block_tail(dev(A, half), x32, pack, hidden, scatter_map=smap, map_rows=Lp, out_rows=L, **kw)
    xh = dev(x16.clone())
    n16 = kernels.block_tail(dev(A, half), xh, pack, hidden, scatter_map=smap, map_rows=Lp, out_rows=L, **kw)
    assert xh.dtype == torch.float16 and torch.equal(xh, x32.clamp(-65504.0, 65504.0).to(torch.float16))
    assert torch.isfinite(xh.float()).all()
    if n32 is not None:
        assert torch.equal(n16, n32)


@pytest.mark.parametrize("C,dims,emit", [(96, (2, 4, 14, 14), True), (96, (1, 3, 5, 7), False), (192, (2, 8, 14, 14), True), (128, (1, 4, 8, 8), False)])
def test_patch_merge_fp16_residual_stream(C, dims, emit, half):
    """``x_f16`` / ``out_f16``: the merge launch on an fp16 stream equals the fp32-stream launch on the widened input (bit for bit: same
    operands, same statistics), its fp16 output is that result rounded once; the emitted norm1 rows are identical."""
    B, D, H, W = dims
    g = rng(sum(dims) + 3 * C)
    x16 = torch.from_numpy((g.standard_normal((B * D * H * W, C)) * 1.5).astype(np.float32)).to(torch.float16)
    wts = (dev(torch.from_numpy((g.standard_normal((2 * C, 4 * C)) / np.sqrt(4 * C)).astype(np.float32))),
           dev(torch.from_numpy((1 + 0.2 * g.standard_normal(4 * C)).astype(np.float32))), dev(torch.from_numpy((0.2 * g.standard_normal(4 * C)).astype(np.float32))))
    mp, Hn, Wn = _merge_map(D, H, W)
    Ln = D * Hn * Wn
    kw = {}
    if emit:
        lay = O.window_layout(D, Hn, Wn, (8, 7, 7), (0, 0, 0))
        assert not (lay["src"] < 0).any()
        dst = np.empty(Ln, np.int32)
        dst[lay["src"]] = np.arange(Ln, dtype=np.int32)
        kw = dict(next_norm=(dev(torch.from_numpy((1 + 0.2 * g.standard_normal(2 * C)).astype(np.float32))),
                             dev(torch.from_numpy((0.2 * g.standard_normal(2 * C)).astype(np.float32)))), next_dst=dev(torch.from_numpy(dst)), next_rows=Ln)
    mpd = dev(torch.from_numpy(mp))
    o32, n32 = kernels.patch_merge(dev(x16.float()), mpd, B, *wts, out_dtype=half, **kw)
    o_in16, n_in16 = kernels.patch_merge(dev(x16), mpd, B, *wts, out_dtype=half, **kw)                      # fp16 in, fp32 out (stage 1 -> 2)
    o16, n16 = kernels.patch_merge(dev(x16), mpd, B, *wts, out_dtype=half, out_f16=True, **kw)              # fp16 in, fp16 out (stage 0 -> 1)
    assert o_in16.dtype == torch.float32 and torch.equal(o_in16, o32)
    assert o16.dtype == torch.float16 and torch.equal(o16, o32.to(torch.float16))
    if emit:
        assert torch.equal(n_in16, n32) and torch.equal(n16, n32)


@pytest.mark.parametrize("frag", [False, True])
def test_patch_embed_fp16_residual_stream(frag, half):
    """``out_f16``: the embedding launch writes the stream as fp16 = its fp32 result rounded once; the norm1 rows are the same rows."""
    g = rng(404)
    E, B, T, Hc, Wc = 96, 2, 8, 64, 96
    w = rnd(torch.from_numpy((g.standard_normal((E, 3 * 2 * 4 * 4)) * 0.1).astype(np.float32)), half)
    b, lw, lb = (torch.from_numpy((g.standard_normal(E) * s_ + o_).astype(np.float32)) for s_, o_ in ((0.2, 0.0), (0.2, 1.0), (0.2, 0.0)))
    if frag:
        from kvq_amd.datasets import KVQ_MEAN, KVQ_STD
        F, Hs, Ws = 2, 150, 200
        vids = [torch.from_numpy(g.integers(0, 256, size=(3, T, Hs, Ws)).astype(np.uint8)).to(DEV) for _ in range(B)]
        gh, gw = SO.fragment_grid(Hs, F, 32).reshape(F, 1, 1), SO.fragment_grid(Ws, F, 32).reshape(1, F, 1)
        ho = [dev(torch.from_numpy(np.ascontiguousarray(g.integers(0, Hs // F - 32, size=(F, F, T // 8)) + gh, dtype=np.int32))) for _ in range(B)]
        wo = [dev(torch.from_numpy(np.ascontiguousarray(g.integers(0, Ws // F - 32, size=(F, F, T // 8)) + gw, dtype=np.int32))) for _ in range(B)]
        x = kernels.FragmentSource(vids, ho, wo, F, F, 32, 32, 8, mean=KVQ_MEAN, std=KVQ_STD)
        Hc = Wc = 64
    else:
        x = dev(torch.from_numpy((g.standard_normal((B, 3, T, Hc, Wc)) * 1.5).astype(np.float32)))
    D0, H0, W0 = T // 2, Hc // 4, Wc // 4
    lay = O.window_layout(D0, H0, W0, (8, 7, 7), (0, 0, 0))
    kw = {}
    if not (lay["src"] < 0).any():
        dst = np.empty(D0 * H0 * W0, np.int32)
        dst[lay["src"]] = np.arange(D0 * H0 * W0, dtype=np.int32)
        kw = dict(next_norm=(dev(lw), dev(lb)), next_dst=dev(torch.from_numpy(dst)), next_rows=D0 * H0 * W0)
    o32, n32 = kernels.patch_embed(x, dev(w, half), dev(b), dev(lw), dev(lb), (2, 4, 4), **kw)
    o16, n16 = kernels.patch_embed(x, dev(w, half), dev(b), dev(lw), dev(lb), (2, 4, 4), out_f16=True, **kw)
    assert o16.dtype == torch.float16 and torch.equal(o16, o32.to(torch.float16))
    if n32 is not None:
        assert torch.equal(n16, n32)
