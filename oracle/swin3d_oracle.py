"""ORACLE (test infrastructure, never shipped in the product path).

CPU fp32 restatement of the reference's Swin-3D(GRPB) trunk + VQAHead, written from
the arithmetic spec in SURVEY.md App. A, *not* from the reference's module code:
all window / shift / fragment-gate / mask logic is expressed as explicit integer
index maps so it doubles as the specification of the index arithmetic the HIP
kernels implement.

Pinned: ``tests/golden/make_golden.py`` imports the real reference (in the build
container only) and checks this file against it, then emits the fixtures under
``tests/golden/`` that ``tests/test_oracle_golden.py`` re-checks everywhere.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.

Reference lines restated (``/root/reference/models/backbones/swin_backbone.py``):
  get_window_size :145-158 · window_partition/reverse :92-142 · global_position_index
  :21-50 · compute_mask :559-586 · WindowAttention3D :161-326 · block :407-516 ·
  PatchMerging :533-556 · PatchEmbed3D :715-733 · BasicLayer :660-687 · trunk forward
  :1044-1080; ``models/head.py:60-68`` (VQAHead), ``:28-31`` (simpleVQAHead).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# integer layout logic
# --------------------------------------------------------------------------------------
def clamp_window(dims: Sequence[int], window: Sequence[int], shift: Sequence[int]):
    """swin_backbone.py:145-158 — a dim no larger than the window uses the whole dim and no shift."""
    ws = [window[i] if dims[i] > window[i] else dims[i] for i in range(3)]
    ss = [shift[i] if dims[i] > window[i] else 0 for i in range(3)]
    return tuple(ws), tuple(ss)


def _nearest_src(dst: np.ndarray, in_size: int, out_size: int) -> np.ndarray:
    """Legacy ``F.interpolate(mode='nearest')`` source index (ATen nearest_idx):
    float32 scale = in/out, floor(dst*scale), clamped — with the two exact fast paths."""
    if out_size == in_size:
        return dst.copy()
    if out_size == 2 * in_size:
        return dst >> 1
    scale = np.float32(in_size) / np.float32(out_size)
    return np.minimum(np.floor(dst.astype(np.float32) * scale).astype(np.int64), in_size - 1)


def window_layout(D: int, H: int, W: int, window: Sequence[int], shift: Sequence[int], adaptive: Optional[Sequence[int]] = None):
    """All index maps of one (stage, block-parity) for a (D,H,W) token grid.

    Returns a dict with
      ws, ss        clamped window / shift
      Dp,Hp,Wp      padded dims, nW windows, N tokens per window
      src           (nW*N,) int64: flat index into the D*H*W token grid feeding each windowed row
                    (after pad + roll(-ss) + partition), or -1 for a zero pad row
      frag          (nW*N, 2) int64: fragment ids (h, w) of each windowed row (global_position_index)
      region        (nW*N,) int64: shift-mask region id of each windowed row (compute_mask)
      sub           None, or — ``adaptive`` = forward(adaptive_window_size=True)'s resized window (swin_backbone.py:54-61, :1050-1055):
                    the resized window partitions (clamped as usual, :408-413) while ``shift`` stays the configured block's — the
                    clamped resized window, whose own coordinates index the bias tables (:266-271)
    """
    ws, ss = clamp_window((D, H, W), window if adaptive is None else adaptive, shift)
    Dp = -(-D // ws[0]) * ws[0]
    Hp = -(-H // ws[1]) * ws[1]
    Wp = -(-W // ws[2]) * ws[2]
    nd, nh, nw = Dp // ws[0], Hp // ws[1], Wp // ws[2]
    N = ws[0] * ws[1] * ws[2]
    nW = nd * nh * nw
    # windowed row -> coordinate in the shifted (rolled) padded frame
    wd, wh, ww, ld, lh, lw = np.meshgrid(np.arange(nd), np.arange(nh), np.arange(nw),
                                         np.arange(ws[0]), np.arange(ws[1]), np.arange(ws[2]),
                                         indexing="ij")
    sd = (wd * ws[0] + ld).reshape(-1)
    sh = (wh * ws[1] + lh).reshape(-1)
    sw = (ww * ws[2] + lw).reshape(-1)
    # roll(-ss): shifted[p] = padded[(p + ss) mod P]
    ud, uh, uw = (sd + ss[0]) % Dp, (sh + ss[1]) % Hp, (sw + ss[2]) % Wp
    valid = (ud < D) & (uh < H) & (uw < W)
    src = np.where(valid, (ud * H + uh) * W + uw, -1).astype(np.int64)
    # fragment ids live on the padded, un-rolled grid: nearest-interpolated (1,ws_h,ws_w) mesh
    fh = _nearest_src(uh, ws[1], Hp)
    fw = _nearest_src(uw, ws[2], Wp)
    frag = np.stack([fh, fw], -1).astype(np.int64)

    # shift-mask regions are painted on the shifted frame (no roll), last writer wins
    def axis_region(s, P, w, sft):
        if sft == 0:
            return np.full_like(s, 2)           # third slice == whole axis
        r = np.zeros_like(s)
        r[s >= P - w] = 1
        r[s >= P - sft] = 2
        return r
    region = (axis_region(sd, Dp, ws[0], ss[0]) * 9 + axis_region(sh, Hp, ws[1], ss[1]) * 3
              + axis_region(sw, Wp, ws[2], ss[2])).astype(np.int64)
    return dict(ws=ws, ss=ss, Dp=Dp, Hp=Hp, Wp=Wp, nW=nW, N=N, src=src, frag=frag, region=region,
                sub=None if adaptive is None else tuple(ws))


def rel_pos_index(window: Sequence[int], N: Optional[int] = None, sub: Optional[Sequence[int]] = None) -> np.ndarray:
    """(N,N) index into the bias tables.  Token n takes the raster coordinate of the
    *configured* window (the reference slices the full table ``[:N,:N]`` when the window
    was clamped, swin_backbone.py:263-264).  ``sub`` (adaptive windows, :266-271): the index is
    ``relative_position_index.reshape(*window, *window)[:d,:h,:w,:d,:h,:w]`` — token n's coordinate in the (d,h,w) sub-window."""
    Wd, Wh, Ww = window
    full = Wd * Wh * Ww
    if sub is not None:
        N = sub[0] * sub[1] * sub[2]
        n = np.arange(N)
        cd, ch, cw = n // (sub[1] * sub[2]), (n // sub[2]) % sub[1], n % sub[2]
    else:
        N = full if N is None else N
        n = np.arange(N)
        cd, ch, cw = n // (Wh * Ww), (n // Ww) % Wh, n % Ww
    dd = cd[:, None] - cd[None, :] + (Wd - 1)
    dh = ch[:, None] - ch[None, :] + (Wh - 1)
    dw = cw[:, None] - cw[None, :] + (Ww - 1)
    return (dd * (2 * Wh - 1) * (2 * Ww - 1) + dh * (2 * Ww - 1) + dw).astype(np.int64)


def frag_gate(layout) -> np.ndarray:
    """(nW,N,N) integer gate g = sum |frag_i - frag_j| (frag_d is identically 0)."""
    f = layout["frag"].reshape(layout["nW"], layout["N"], 2)
    return np.abs(f[:, :, None, :] - f[:, None, :, :]).sum(-1)


def shift_mask(layout) -> Optional[np.ndarray]:
    """(nW,N,N) fp32 additive mask, 0 / -100 (NOT -inf); None when no axis is shifted."""
    if not any(s > 0 for s in layout["ss"]):
        return None
    r = layout["region"].reshape(layout["nW"], layout["N"])
    return np.where(r[:, :, None] == r[:, None, :], 0.0, -100.0).astype(np.float32)


# --------------------------------------------------------------------------------------
# floating-point path
# --------------------------------------------------------------------------------------
def _t(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def _ident(t: torch.Tensor) -> torch.Tensor:
    return t


def operand_rounding(dtype):
    """Rounding applied to MFMA operands by the HIP path (None -> exact fp32 reference arithmetic).
    With a dtype, the oracle becomes an *emulation* of the kernels' 16-bit operand rounding (same
    rounding points, fp32 accumulation) used to separate kernel bugs from format precision."""
    if dtype is None:
        return _ident
    if dtype == torch.float16:
        return lambda t: t.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)
    return lambda t: t.to(dtype).to(torch.float32)


def patch_embed(x: torch.Tensor, p: Dict[str, torch.Tensor], patch=(2, 4, 4), q=_ident) -> torch.Tensor:
    """(B,3,T,H,W) -> channels-last tokens (B,D,H',W',E); zero pad at the end of each axis."""
    pd, ph, pw = patch
    _, _, T, H, W = x.shape
    x = F.pad(x, (0, (-W) % pw, 0, (-H) % ph, 0, (-T) % pd))
    y = F.conv3d(q(x), q(p["patch_embed.proj.weight"]), p["patch_embed.proj.bias"], stride=patch)
    y = y.permute(0, 2, 3, 4, 1)
    if "patch_embed.norm.weight" in p:
        y = F.layer_norm(y, (y.shape[-1],), p["patch_embed.norm.weight"], p["patch_embed.norm.bias"])
    return y.contiguous()


def gather_windows(h: torch.Tensor, layout) -> torch.Tensor:
    """(B,D,H,W,C) -> (B*nW, N, C) windowed rows; pad rows are exact zeros."""
    B, C = h.shape[0], h.shape[-1]
    flat = h.reshape(B, -1, C)
    src = _t(layout["src"])
    rows = flat[:, src.clamp(min=0)]
    rows = rows * (src >= 0).to(rows.dtype)[None, :, None]
    return rows.reshape(B * layout["nW"], layout["N"], C)


def scatter_windows(o: torch.Tensor, layout, B: int, D: int, H: int, W: int) -> torch.Tensor:
    """inverse of gather_windows (window_reverse + roll(+ss) + crop)."""
    C = o.shape[-1]
    src = _t(layout["src"])
    keep = src >= 0
    out = o.new_zeros(B, D * H * W, C)
    out[:, src[keep]] = o.reshape(B, -1, C)[:, keep]
    return out.reshape(B, D, H, W, C)


def window_attention(xw: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, num_heads: int,
                     window, layout, chunk: int = 64, q=_ident, kernel_order: bool = False) -> torch.Tensor:
    """xw (B*nW,N,C) -> (B*nW,N,C).  Implements SURVEY.md App. A item 3.  ``q`` = operand rounding; ``kernel_order`` (emulation only):
    the softmax at csrc/attn32.hip's rounding points (``attention_core_kernel_order``)."""
    BW, N, C = xw.shape
    hd = C // num_heads
    qkv = F.linear(xw, q(p[pre + "qkv.weight"]), p[pre + "qkv.bias"]).reshape(BW, N, 3, num_heads, hd)
    k = q(qkv[:, :, 1].permute(0, 2, 1, 3))
    v = q(qkv[:, :, 2].permute(0, 2, 1, 3))
    if kernel_order:
        qq = q(qkv[:, :, 0].permute(0, 2, 1, 3) * np.float32(hd ** -0.5 * 1.4426950408889634))
        out = q(attention_core_kernel_order(qq, k, v, p[pre + "relative_position_bias_table"],
                                            p.get(pre + "fragment_position_bias_table"), window, layout, q, chunk))
    else:
        qq = q(qkv[:, :, 0].permute(0, 2, 1, 3) * (hd ** -0.5))
        out = q(attention_core(qq, k, v, p[pre + "relative_position_bias_table"],
                               p.get(pre + "fragment_position_bias_table"), window, layout, chunk, q))
    return F.linear(out, q(p[pre + "proj.weight"]), p[pre + "proj.bias"])


def attention_bias(rpb_table: torch.Tensor, fpb_table: Optional[torch.Tensor], window, layout) -> torch.Tensor:
    """(nW,nH,N,N) additive term: gated table bias (+ shift mask)."""
    N, nW, nH = layout["N"], layout["nW"], rpb_table.shape[1]
    rpi = _t(rel_pos_index(window, N, layout.get("sub"))).reshape(-1)
    rpb = rpb_table[rpi].reshape(N, N, nH).permute(2, 0, 1)
    g = _t(frag_gate(layout)).to(torch.float32)                     # (nW,N,N)
    if fpb_table is not None:
        fpb = fpb_table[rpi].reshape(N, N, nH).permute(2, 0, 1)
        bias = rpb[None] * g[:, None] + fpb[None] * (1.0 - g[:, None])
    else:
        bias = rpb[None].expand(nW, -1, -1, -1)
    m = shift_mask(layout)
    if m is not None:
        bias = bias + _t(m)[:, None]
    return bias


def image_bias(rpb_table: torch.Tensor, fpb_table: Optional[torch.Tensor], window, layout) -> torch.Tensor:
    """The additive term as the HIP kernels' pre-built bias image stores it (csrc/attn32.hip bias32_build_kernel): per query row the table bias minus its maximum over the un-masked keys, masked entries REPLACED by
    -100 (minus the same shift), everything rounded to fp16.  Softmax is invariant to the per-row shift, so what this emulates is the
    image's 2^-11 relative rounding of (bias - row maximum) — test infrastructure for the kernels' tolerance, not a reference path."""
    N, nW, nH = layout["N"], layout["nW"], rpb_table.shape[1]
    rpi = _t(rel_pos_index(window, N, layout.get("sub"))).reshape(-1)
    rpb = rpb_table[rpi].reshape(N, N, nH).permute(2, 0, 1)
    if fpb_table is not None:
        g = _t(frag_gate(layout)).to(torch.float32)
        fpb = fpb_table[rpi].reshape(N, N, nH).permute(2, 0, 1)
        bias = rpb[None] * g[:, None] + fpb[None] * (1.0 - g[:, None])
    else:
        bias = rpb[None].expand(nW, -1, -1, -1).clone()
    m = shift_mask(layout)
    masked = torch.zeros(nW, 1, N, N, dtype=torch.bool) if m is None else (_t(m)[:, None] != 0)
    shift = bias.masked_fill(masked, float("-inf")).max(-1, keepdim=True).values
    img = torch.where(masked, torch.full_like(bias, -100.0), bias) - shift
    return img.to(torch.float16).to(torch.float32)


def attention_core(q, k, v, rpb_table, fpb_table, window, layout, chunk: int = 64, rq=_ident, image: bool = False) -> torch.Tensor:
    """q (pre-scaled), k, v: (B*nW, nH, N, hd) -> (B*nW, N, nH*hd).  ``rq`` rounds the un-normalised
    probabilities the way the kernel does (P is an MFMA operand); identity = exact softmax.  ``image``: the bias as the kernels'
    fp16 image holds it (``image_bias``) instead of the exact fp32 term."""
    BW, nH, N, hd = q.shape
    nW = layout["nW"]
    bias = image_bias(rpb_table, fpb_table, window, layout) if image else attention_bias(rpb_table, fpb_table, window, layout)
    out = torch.empty(BW, N, nH * hd, dtype=q.dtype)
    widx = torch.arange(BW) % nW
    for s in range(0, BW, chunk):
        e = min(BW, s + chunk)
        a = q[s:e] @ k[s:e].transpose(-2, -1) + bias[widx[s:e]]
        if rq is _ident:
            a = torch.softmax(a, dim=-1)
            o = a @ v[s:e]
        else:
            pe = rq(torch.exp(a - a.max(-1, keepdim=True).values))
            o = (pe @ v[s:e]) / pe.sum(-1, keepdim=True)
        out[s:e] = o.transpose(1, 2).reshape(e - s, N, nH * hd)
    return out


def attention_core_kernel_order(qk2, k, v, rpb_table, fpb_table, window, layout, rq, chunk: int = 64,
                                block: int = 32, thr: float = 8.0) -> torch.Tensor:
    """The softmax core at the ROUNDING POINTS of csrc/attn32.hip (test infrastructure for the bf16 / fp16 emulation, not a reference
    path): scores in log2 units (``qk2`` = q * head_dim^-0.5 * log2(e), rounded as the qkv epilogue rounds it), the bias as the fp16
    row-max-shifted image times log2(e), keys taken in blocks of 32 against a RUNNING row maximum with the deferred rescale (the
    row's first block is shifted by its exact maximum, a later block only when it grows more than 2^thr past the running one), the
    un-normalised probabilities rounded by ``rq`` at that scale, the normaliser the sum of the ROUNDED probabilities.  At which scale a
    probability is rounded decides which way each 8-bit (bf16) rounding falls: an emulation that rounds exp(a - exact maximum)
    carries the same amount of rounding noise but a different draw of it."""
    LOG2E = 1.4426950408889634
    BW, nH, N, hd = qk2.shape
    nW = layout["nW"]
    img = image_bias(rpb_table, fpb_table, window, layout) * LOG2E                     # (nW | 1, nH, N, N), log2 units
    out = torch.empty(BW, N, nH * hd, dtype=qk2.dtype)
    widx = torch.arange(BW) % nW if img.shape[0] == nW else torch.zeros(BW, dtype=torch.long)
    for s in range(0, BW, chunk):
        e = min(BW, s + chunk)
        a = qk2[s:e] @ k[s:e].transpose(-2, -1) + img[widx[s:e]]                     # relative to nm = 0
        O = torch.zeros(e - s, nH, N, hd)
        l = torch.zeros(e - s, nH, N, 1)
        nm = torch.zeros(e - s, nH, N, 1)                                            # -(running maximum)
        for t0 in range(0, N, block):
            S = a[..., t0:t0 + block] + nm
            mr = S.max(-1, keepdim=True).values
            d = mr if t0 == 0 else torch.where(mr > thr, mr, torch.zeros_like(mr))
            S = S - d
            if t0:
                f = torch.exp2(-d)
                O, l = O * f, l * f
            nm = nm - d
            P = rq(torch.exp2(S))
            l = l + P.sum(-1, keepdim=True)
            O = O + P @ v[s:e, :, t0:t0 + block]
        out[s:e] = (O / l).transpose(1, 2).reshape(e - s, N, nH * hd)
    return out


def swin_block(x: torch.Tensor, p, pre: str, num_heads: int, window, shift, q=_ident, kernel_order: bool = False,
               adaptive=None) -> torch.Tensor:
    """x (B,D,H,W,C) channels-last residual stream (always fp32)."""
    B, D, H, W, C = x.shape
    lay = window_layout(D, H, W, window, shift, adaptive)
    h = q(F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"]))
    o = window_attention(gather_windows(h, lay), p, pre + "attn.", num_heads, window, lay, q=q, kernel_order=kernel_order)
    x = x + scatter_windows(o, lay, B, D, H, W)
    h = q(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"]))
    h = F.linear(h, q(p[pre + "mlp.fc1.weight"]), p[pre + "mlp.fc1.bias"])
    h = q(F.gelu(h))
    h = F.linear(h, q(p[pre + "mlp.fc2.weight"]), p[pre + "mlp.fc2.bias"])
    return x + h


def patch_merge(x: torch.Tensor, p, pre: str, q=_ident) -> torch.Tensor:
    B, D, H, W, C = x.shape
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    cat = q(F.layer_norm(cat, (4 * C,), p[pre + "norm.weight"], p[pre + "norm.bias"]))
    return F.linear(cat, q(p[pre + "reduction.weight"]))


def patch_merge_kernel_order(x: torch.Tensor, p, pre: str, q, eps: float = 1e-5) -> torch.Tensor:
    """PatchMerging at the ROUNDING POINTS of csrc/merge.hip (C <= 192; emulation only, as ``attention_core_kernel_order``): the launch
    folds the LayerNorm around the GEMM — W (gamma (x - mean) rstd + beta) = rstd (W diag(gamma)) (x - mean) + W beta — so its 16-bit
    operands are W' = W diag(gamma) and d = x - K (K = the mean of the concatenated row's first 96 channels), not W and the normalised
    row; the statistics come from the un-rounded d, mean - K leaves through the row sums of the ROUNDED W', W beta stays fp32."""
    B, D, H, W, C = x.shape
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    g, b, w = p[pre + "norm.weight"], p[pre + "norm.bias"], p[pre + "reduction.weight"]
    d = cat - cat[..., :96].mean(-1, keepdim=True)
    md = d.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((d * d).mean(-1, keepdim=True) - md * md).clamp(min=0.0) + eps)
    wg = q(w * g[None, :])
    return (F.linear(q(d), wg) - md * wg.sum(1)) * rstd + F.linear(b[None], w)[0]


def swin3d_trunk(x: torch.Tensor, params, cfg, return_stages: bool = False, operand_dtype=None, kernel_order: bool = False,
                 merge_fold_max_c: int = 192, adaptive_window=None):
    """x (B,3,T,H,W) fp32 -> (B,C_out,D,H/32,W/32) like the reference trunk.  ``cfg`` is a
    ``kvq_amd.utils.synth.SwinCfg``-shaped object (patch, depths, num_heads, window).
    ``operand_dtype`` None = the reference's fp32 arithmetic (this is THE oracle); torch.float16 /
    torch.bfloat16 = emulate the HIP path's MFMA-operand rounding (diagnostic only); ``kernel_order`` (with an operand dtype) also
    takes the attention kernel's softmax order and scales (``attention_core_kernel_order``) and the fused merge launch's operands
    (``patch_merge_kernel_order``, widths up to ``merge_fold_max_c``).  ``adaptive_window``: forward(adaptive_window_size=True)'s
    resized window (``window * clip size // base_x_size``, swin_backbone.py:54-61), see ``window_layout``."""
    p = {k: _t(v).float() for k, v in params.items()}
    q = operand_rounding(operand_dtype)
    shift = tuple(w // 2 for w in cfg.window)
    y = patch_embed(x.float(), p, cfg.patch, q)
    stages = [y]
    for i in range(len(cfg.depths)):
        for b in range(cfg.depths[i]):
            y = swin_block(y, p, f"layers.{i}.blocks.{b}.", cfg.num_heads[i], cfg.window,
                           (0, 0, 0) if b % 2 == 0 else shift, q, kernel_order and operand_dtype is not None, adaptive_window)
        if i < len(cfg.depths) - 1:
            if kernel_order and operand_dtype is not None and y.shape[-1] <= merge_fold_max_c:      # csrc/merge.hip (plan.hip KVQ_MERGE_MAXC)
                y = patch_merge_kernel_order(y, p, f"layers.{i}.downsample.", q)
            else:
                y = patch_merge(y, p, f"layers.{i}.downsample.", q)
        stages.append(y)
    y = F.layer_norm(y, (y.shape[-1],), p["norm.weight"], p["norm.bias"])
    out = y.permute(0, 4, 1, 2, 3).contiguous()
    return (out, stages) if return_stages else out


def vqa_head(feat: torch.Tensor, hp, pre_pool: bool = False) -> torch.Tensor:
    """feat (B,C,D,H,W) -> (B,num_class): mean over tokens of w2·GELU(W1 f + b1) + b2 (head.py:60-68, eval); ``pre_pool``: the token
    grid is averaged first (:61-62); num_class > 1: nn.Softmax() — implicit dim 1, the classes — per token before the mean (:66-67)."""
    p = {k: _t(v).float() for k, v in hp.items()}
    B, C = feat.shape[:2]
    f = feat.reshape(B, C, -1).transpose(1, 2)
    if pre_pool:
        f = f.mean(dim=1, keepdim=True)
    K = p["fc_last.weight"].shape[0]
    h = F.gelu(F.linear(f, p["fc_hid.weight"].reshape(-1, C), p["fc_hid.bias"]))
    s = F.linear(h, p["fc_last.weight"].reshape(K, -1), p["fc_last.bias"])
    if K > 1:
        s = torch.softmax(s, dim=-1)
    return s.mean(dim=1)


def simple_vqa_head(feat: torch.Tensor, hp) -> torch.Tensor:
    """feat (B,T,9472) -> (B,1): two Linears without activation, mean over frames (head.py:28-31)."""
    p = {k: _t(v).float() for k, v in hp.items()}
    h = F.linear(feat, p["quality.0.weight"], p["quality.0.bias"])
    return F.linear(h, p["quality.1.weight"], p["quality.1.bias"]).mean(dim=1)


def swin_flops(cfg, T: int, H: int, W: int) -> float:
    """Algorithmic GEMM FLOPs (2·MAC) per clip, padding as the reference pads."""
    pd, ph, pw = cfg.patch
    D, Hh, Ww = -(-T // pd), -(-H // ph), -(-W // pw)
    fl = 2.0 * D * Hh * Ww * cfg.embed_dim * (cfg.in_chans * pd * ph * pw)
    for i in range(len(cfg.depths)):
        C = cfg.embed_dim * 2 ** i
        ws, _ = clamp_window((D, Hh, Ww), cfg.window, (0, 0, 0))
        Lp = math.prod(-(-d // w) * w for d, w in zip((D, Hh, Ww), ws))
        N, L = math.prod(ws), D * Hh * Ww
        fl += cfg.depths[i] * (2.0 * Lp * C * 3 * C + 4.0 * Lp * N * C + 2.0 * Lp * C * C
                               + 4.0 * cfg.mlp_ratio * L * C * C)
        if i < len(cfg.depths) - 1:
            Hh, Ww = -(-Hh // 2), -(-Ww // 2)
            fl += 2.0 * D * Hh * Ww * 4 * C * 2 * C
    return fl
