"""SlowFast-R50 (8x8) motion feature extractor on libkvq_hip.so — the reference's ``slowfast`` module of
``SlowFast_features.py:137-165`` (blocks 0-4 of ``pytorchvideo.models.hub.slowfast_r50`` + the head's
pools).  pytorchvideo is not vendored / pinned / installed: the architecture below is the
published SlowFast-R50 8x8; parameter names follow pytorchvideo's module tree under
``feature_extraction.`` so that a real ``slowfast_r50`` state_dict loads by name.  **Parity unpinned.**

Execution: channels-last 16-bit activations (B,D,H,W,C); every Conv3d = ``kvq_im2col_nd`` gather + MFMA GEMM
with BatchNorm folded and ReLU / identity add in the epilogue (1x1x1 stride-1 convs skip the gather); the
residual stream between bottlenecks stays fp32; pools = ``kvq_pool_nd`` / ``kvq_mean_std_pool``."""
from __future__ import annotations

import os
from collections import OrderedDict

import weakref

import torch
import torch.nn as nn

from ... import _abi, kernels

DEPTHS = (3, 4, 6, 3)
SLOW = dict(inner=(64, 128, 256, 512), out=(256, 512, 1024, 2048), ka=(1, 1, 3, 3))
FAST = dict(inner=(8, 16, 32, 64), out=(32, 64, 128, 256), ka=(3, 3, 3, 3))
SPATIAL_STRIDE = (1, 2, 2, 2)
FAST_C = (8, 32, 64, 128)          # fast-pathway channels entering fusion 0..3


def conv_table():
    """name -> (weight shape, stride3, pad3): every Conv3d (+ its norm's name) in execution order."""
    t = OrderedDict()
    fe = "feature_extraction."
    t[fe + "0.multipathway_blocks.0"] = ((64, 3, 1, 7, 7), (1, 2, 2), (0, 3, 3), "conv", "norm")
    t[fe + "0.multipathway_blocks.1"] = ((8, 3, 5, 7, 7), (1, 2, 2), (2, 3, 3), "conv", "norm")
    for st in range(4):
        t[fe + f"{st}.multipathway_fusion"] = ((2 * FAST_C[st], FAST_C[st], 7, 1, 1), (4, 1, 1), (3, 0, 0),
                                               "conv_fast_to_slow", "norm")
    slow_in = (64 + 16, 256 + 64, 512 + 128, 1024 + 256)
    for si in range(4):
        for pi, (cfg, cin0) in enumerate(((SLOW, slow_in[si]), (FAST, FAST_C[si]))):
            cin = cin0
            for bi in range(DEPTHS[si]):
                pre = fe + f"{si + 1}.multipathway_blocks.{pi}.res_blocks.{bi}"
                inner, cout, ka = cfg["inner"][si], cfg["out"][si], cfg["ka"][si]
                s = SPATIAL_STRIDE[si] if bi == 0 else 1
                if bi == 0:
                    t[pre + "#1"] = ((cout, cin, 1, 1, 1), (1, s, s), (0, 0, 0), "branch1_conv", "branch1_norm")
                t[pre + ".branch2#a"] = ((inner, cin, ka, 1, 1), (1, 1, 1), (ka // 2, 0, 0), "conv_a", "norm_a")
                t[pre + ".branch2#b"] = ((inner, inner, 1, 3, 3), (1, s, s), (0, 1, 1), "conv_b", "norm_b")
                t[pre + ".branch2#c"] = ((cout, inner, 1, 1, 1), (1, 1, 1), (0, 0, 0), "conv_c", "norm_c")
                cin = cout
    return t


def conv_flops(T=32, H=224, W=224):
    """ALGORITHMIC flops (2 x MAC, convolutions only, padding taps counted as the reference's F.conv3d computes them) of ONE
    clip through blocks 0-4, counted from ``conv_table()`` by walking the forward's shapes: slow pathway T/4 frames, fast
    pathway T frames; stem stride (1,2,2) + max-pool (1,2,2); res2..res5 at H/4 .. H/32.  Returns (total, per-conv dict)."""
    def out_sz(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    per = OrderedDict()
    fe = "feature_extraction."
    tab = conv_table()
    dims = {0: [T // 4, H, W], 1: [T, H, W]}                   # pathway -> current (D, H, W)
    def conv(key, d3):
        (co, ci, kd, kh, kw), st, pd = tab[key][0], tab[key][1], tab[key][2]
        o = [out_sz(d3[0], kd, st[0], pd[0]), out_sz(d3[1], kh, st[1], pd[1]), out_sz(d3[2], kw, st[2], pd[2])]
        per[key] = 2.0 * o[0] * o[1] * o[2] * co * ci * kd * kh * kw
        return o
    for pi in (0, 1):
        d = conv(fe + f"0.multipathway_blocks.{pi}", dims[pi])
        dims[pi] = [d[0], out_sz(d[1], 3, 2, 1), out_sz(d[2], 3, 2, 1)]            # MaxPool3d (1,3,3) / (1,2,2) / (0,1,1)
    conv(fe + "0.multipathway_fusion", dims[1])
    for si in range(4):
        for pi in (0, 1):
            for bi in range(DEPTHS[si]):
                pre = fe + f"{si + 1}.multipathway_blocks.{pi}.res_blocks.{bi}"
                if bi == 0:
                    conv(pre + "#1", dims[pi])
                d = conv(pre + ".branch2#a", dims[pi])
                d = conv(pre + ".branch2#b", d)
                dims[pi] = conv(pre + ".branch2#c", d)
        if si < 3:
            conv(fe + f"{si + 1}.multipathway_fusion", dims[1])
    return sum(per.values()), per


def _set_nested(root: nn.Module, dotted: str, value, buffer=False):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, nn.Module())
        m = getattr(m, p)
    if buffer:
        m.register_buffer(parts[-1], value)
    else:
        m.register_parameter(parts[-1], value)


def pack_pathway_output(frames, device=None):
    """Reference ``pack_pathway_output`` (SlowFast_features.py:112-135): [slow, fast]."""
    idx = torch.linspace(0, frames.shape[2] - 1, frames.shape[2] // 4).long().to(frames.device)
    slow = torch.index_select(frames, 2, idx)
    out = [slow if device is None else slow.to(device), frames if device is None else frames.to(device)]
    # provenance tag: slowfast.forward may re-select the slow frames on the device (one C call for the whole network) ONLY for a
    # slow tensor that IS this selection of that fast tensor; any other slow tensor is consumed as given, like the reference does
    # (the fast tensor by identity — a weak reference, so a freed tensor whose address is reused cannot match — plus both tensors'
    # versions: an in-place edit of either after the packing makes the pair an ordinary one)
    out[0]._kvq_packed_of = (weakref.ref(out[1]), out[1]._version, out[0]._version)
    return out


def _is_packed_pair(slow_in, fast_in):
    tag = getattr(slow_in, "_kvq_packed_of", None)
    return tag is not None and tag[0]() is fast_in and tag[1] == fast_in._version and tag[2] == slow_in._version


def head_pool_kernel(pathway, T):
    """pytorchvideo's slowfast_r50 head: AvgPool3d((8,7,7)) on the slow pathway, ((32,7,7)) on the fast one, stride 1, no padding
    (SlowFast_features.py:150-151), followed by AdaptiveAvgPool3d(1) (:152).  The temporal extents are those of the 32-frame model."""
    return (8, 7, 7) if pathway == 0 else (32, 7, 7)


def check_head_grid(pathway, grid, small="error"):
    """The reference's pool raises when the final grid is smaller than its kernel; equal = a global mean; larger = the mean of the
    overlapping window means (the real pool is run then).  Returns True when the pool is needed.  ``small`` = "mean": a grid
    smaller than the kernel gets the global mean instead of the error (reduced-size parity tests only; ``slowfast.head_small_grid``)."""
    k = head_pool_kernel(pathway, None)
    if any(g < kk for g, kk in zip(grid, k)):
        if small == "mean":
            return False
        raise _abi.KvqError(f"slowfast head: final {'slow' if pathway == 0 else 'fast'} grid {tuple(grid)} is smaller than the reference's "
                            f"AvgPool3d kernel {k} (the reference raises here too): clips must be >= 32 frames of >= 224 x 224")
    return tuple(grid) != k


IMPLICIT_CONV = True     # 0: materialised im2col + GEMM
# Residual stream between bottlenecks: 16-bit (default) or fp32.  The conv_c launches that carry it are HBM-bound — per output
# element an fp32 stream reads 4 B, writes 4 B + the 16-bit copy the next conv_a needs; a 16-bit stream reads 2 B and writes 2 B
# (17 launches, 0.92 -> ~0.5 ms per video of 8 clips).  HIP vs the fp32 CPU restatement stays within the 5e-3 relative-L2 bar of
# tests/test_slowfast.py either way (fp32 accumulation inside every conv; one extra 16-bit rounding per block).
RESIDUAL16 = True
CONVNET = True                  # 0: the layer-by-layer Python sequencing below
FUSE_FAST = True            # 0: the fast pathway's residual blocks as 3-4 conv launches each
FUSE_SLOW = True            # 0: the slow pathway's res2 identity blocks as three conv launches each
TWO_LANES = True                # 0: both pathways on the caller's stream, one op after the other
STEM_MFMA = True             # 0: fast-pathway stem on the fp32 direct kernel
STEM_POOL = True             # 0: the one-call plan runs the fast stem as pack + conv + max-pool launches instead of kvq_conv_stem_pool


def _fragments(w, k_real, row_tiles, accumulator_order=False):
    """[R][>= k_real] 16-bit weights -> MFMA A fragments [row_tiles][ceil(k_real / 16)][64 lanes][8]"""
    dev = w.device
    ks = -(-k_real // 16)
    wp = torch.zeros(row_tiles * 32, ks * 16, dtype=w.dtype, device=dev)
    wp[:w.shape[0], :k_real] = w[:, :k_real]
    lane, e, f = torch.arange(64, device=dev), torch.arange(8, device=dev), torch.arange(ks, device=dev)
    m, h = (lane & 31)[None, :, None], (lane >> 5)[None, :, None]
    if accumulator_order:
        kk = 16 * f[:, None, None] + 8 * (e >> 2)[None, None, :] + 4 * h + (e & 3)[None, None, :]
    else:
        kk = 16 * f[:, None, None] + 8 * h + e[None, None, :]
    rows = 32 * torch.arange(row_tiles, device=dev)[:, None, None, None] + m[None]
    return wp[rows.expand(row_tiles, ks, 64, 8), kk[None].expand(row_tiles, ks, 64, 8)].contiguous()


def pack_fast_bottleneck(wa, ba, wb, bb, wc, bc, cin, ws=None, bs=None, stride=1):
    """The packed image of ``kvq_fast_bottleneck`` (layout: include/kvq_hip.h) from BatchNorm-folded 16-bit weights with
    (kd, kh, kw, c)-ordered columns — conv_a [ci][>= 3 cin], conv_b [ci][>= 9 ci], conv_c [cout][>= ci], optional projection
    [cout][>= cin] — and fp32 biases."""
    ci, cout = wa.shape[0], wc.shape[0]
    parts = [_fragments(wa, 3 * cin, 1), _fragments(wb, 9 * ci, 1), _fragments(wc, ci, cout // 32, True)]
    bias_c = bc
    if ws is not None:
        parts.append(_fragments(ws, cin, cout // 32))
        bias_c = bc + bs
    bias = torch.zeros(-(-(64 + cout) * 4 // 1024) * 256, dtype=torch.float32, device=wa.device)
    bias[:ci], bias[32:32 + ci], bias[64:64 + cout] = ba, bb, bias_c
    blob = torch.cat([t.reshape(-1).view(torch.uint8) for t in parts] + [bias.view(torch.uint8)]).contiguous()
    need = _abi.lib().kvq_fast_bottleneck_pack_bytes(cin, ci, cout, int(ws is not None), stride)
    assert need == blob.numel(), (cin, ci, cout, need, blob.numel())
    return blob


def pack_slow_bottleneck(wa, ba, wb, bb, wc, bc):
    """The packed image of ``kvq_slow_bottleneck`` (layout: include/kvq_hip.h) from BatchNorm-folded 16-bit weights with (kd, kh, kw, c)-
    ordered columns — conv_a [64][>= 256], conv_b [64][>= 576], conv_c [256][>= 64] — and fp32 biases."""
    ci, cout, cin = wa.shape[0], wc.shape[0], 256
    parts = [_fragments(wa, cin, 2), _fragments(wb, 9 * ci, 2), _fragments(wc, ci, cout // 32, True)]
    bias = torch.zeros(512, dtype=torch.float32, device=wa.device)
    bias[:ci], bias[64:64 + ci], bias[128:128 + cout] = ba, bb, bc
    blob = torch.cat([t.reshape(-1).view(torch.uint8) for t in parts] + [bias.view(torch.uint8)]).contiguous()
    need = _abi.lib().kvq_slow_bottleneck_pack_bytes(cin, ci, cout)
    assert need == blob.numel(), (cin, ci, cout, need, blob.numel())
    return blob


class slowfast(nn.Module):  # noqa: N801  (reference spelling)
    def __init__(self, operand_dtype=None, two_lanes=None):
        """``two_lanes``: the fast pathway on the plan's second HIP stream (default: KVQ_SF_LANES, on).  Measured: 4.35 -> 3.91 ms
        per video when SlowFast runs alone, but a loss when another branch (the Swin trunk of config C3) already fills the
        chip from its own stream — that caller passes False."""
        super().__init__()
        self.operand_dtype = _abi.dtype_code(operand_dtype or os.environ.get("KVQ_OPERAND_DTYPE", "fp16"))
        self.two_lanes = TWO_LANES if two_lanes is None else bool(two_lanes)
        # False: no launch cuts K, so a clip's features do not depend on how many clips share its forward (the feature
        # extractor writes per-clip files; the reference runs batch 1) — datasets/slowfast_clips.py::extract_video sets it
        self.split_k = True
        # final grid smaller than the head's AvgPool3d kernel ((8,7,7) / (32,7,7): clips under 32 x 224 x 224): "error" as the
        # reference does, or "mean" = the global mean (what oracle/slowfast_oracle.py restates; reduced-size tests set it)
        self.head_small_grid = "error"
        self.table = conv_table()
        for key, (wshape, _, _, cname, nname) in self.table.items():
            base = key.split("#")[0]
            w = torch.empty(wshape)
            nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
            _set_nested(self, f"{base}.{cname}.weight", nn.Parameter(w))
            c = wshape[0]
            _set_nested(self, f"{base}.{nname}.weight", nn.Parameter(torch.ones(c)))
            _set_nested(self, f"{base}.{nname}.bias", nn.Parameter(torch.zeros(c)))
            _set_nested(self, f"{base}.{nname}.running_mean", torch.zeros(c), buffer=True)
            _set_nested(self, f"{base}.{nname}.running_var", torch.ones(c), buffer=True)
            _set_nested(self, f"{base}.{nname}.num_batches_tracked", torch.tensor(0, dtype=torch.long), buffer=True)
        self._wcache = None

    # ---- weights: BatchNorm folded, (kd,kh,kw,c) column order, K padded to 32, 16-bit -----------------
    def _weights(self, device):
        sd = self.state_dict()
        sig = (self.operand_dtype,) + tuple((t.data_ptr(), t._version) for t in sd.values())
        if self._wcache is not None and self._wcache[0] == sig:
            return self._wcache[1]
        half = _abi.torch_dtype(self.operand_dtype)
        out = {}
        for key, (wshape, stride, pad, cname, nname) in self.table.items():
            base = key.split("#")[0]
            w = sd[f"{base}.{cname}.weight"].to(device, torch.float32)
            g, b = sd[f"{base}.{nname}.weight"].to(device, torch.float32), sd[f"{base}.{nname}.bias"].to(device, torch.float32)
            mu, var = sd[f"{base}.{nname}.running_mean"].to(device, torch.float32), sd[f"{base}.{nname}.running_var"].to(device, torch.float32)
            scale = g / torch.sqrt(var + 1e-5)
            w = (w * scale.view(-1, 1, 1, 1, 1)).permute(0, 2, 3, 4, 1).reshape(wshape[0], -1)
            if wshape[0] in (8, 16) and w.shape[1] > 256:        # few outputs, long patch: direct fp32 stem conv (conv.hip)
                out[key + "/direct"] = w.t().contiguous()
                kk = tuple(wshape[2:])
                if wshape[0] == 8 and wshape[1] <= 4 and kk[2] == 7 and stride[2] == 2 and pad[2] == 3:
                    # the same stem on the matrix cores (kvq_conv_stem_mfma): 16-bit [kd*kh][16][32] weight image
                    out[key + "/mfma"] = kernels.stem_mfma_pack_weight(out[key + "/direct"], kk, wshape[1], half)
            kpad = -(-w.shape[1] // 32) * 32
            if kpad != w.shape[1]:
                w = torch.nn.functional.pad(w, (0, kpad - w.shape[1]))
            if half == torch.float16:
                w = w.clamp(-65504.0, 65504.0)
            out[key] = (w.to(half).contiguous(), (b - mu * scale).contiguous(), tuple(wshape[2:]), stride, pad)
        self._wcache = (sig, out)
        return out

    def _bottleneck_pack(self, Wt, pre, projection, stride):
        key = pre + "/bneck"
        if key not in Wt:
            cin = self.table[pre + ".branch2#a"][0][1]
            proj = Wt[pre + "#1"][:2] if projection else (None, None)
            Wt[key] = pack_fast_bottleneck(*Wt[pre + ".branch2#a"][:2], *Wt[pre + ".branch2#b"][:2], *Wt[pre + ".branch2#c"][:2], cin, *proj,
                                           stride=stride)
        return Wt[key]

    # ---- conv on channels-last 16-bit (B,D,H,W,C) -----------------------------------------------------
    @staticmethod
    def _cols(x, spec):
        wt, _, k, stride, pad = spec
        B, D, H, W, C = x.shape
        if k == (1, 1, 1) and stride == (1, 1, 1) and C % 32 == 0:
            return x.reshape(-1, C), (D, H, W)
        return kernels.im2col_nd(x, (B, C, D, H, W), (D * H * W * C, 1, H * W * C, W * C, C), k, stride, pad, x.dtype,
                                 wt.shape[1])

    def _conv_relu(self, x, spec):
        wt, bias, k, stride, pad = spec
        pointwise = k == (1, 1, 1) and stride == (1, 1, 1) and x.shape[-1] % 32 == 0
        if IMPLICIT_CONV and not pointwise and x.shape[-1] % 8 == 0 and x.is_contiguous():
            return kernels.conv_implicit(x, wt, bias, k, stride, pad, True)     # no patch matrix
        a, (d, h, w) = self._cols(x, spec)
        return kernels.conv_gemm(a, spec[0], spec[1], True).reshape(x.shape[0], d, h, w, spec[0].shape[0])

    def _res_block(self, x16, x32, W, pre, first):
        out = self._conv_relu(x16, W[pre + ".branch2#a"])
        out = self._conv_relu(out, W[pre + ".branch2#b"])
        r16 = RESIDUAL16
        if first:                                             # projection shortcut: conv + BN, no ReLU
            spec = W[pre + "#1"]
            pointwise = spec[2] == (1, 1, 1) and spec[3] == (1, 1, 1) and x16.shape[-1] % 32 == 0
            if IMPLICIT_CONV and not pointwise and x16.shape[-1] % 8 == 0 and x16.is_contiguous():
                identity = kernels.conv_implicit(x16, spec[0], spec[1], spec[2], spec[3], spec[4], False, store_f32=not r16)
                identity = identity.reshape(-1, identity.shape[-1])
            else:
                a, _ = self._cols(x16, spec)
                identity = kernels.gemm(a, spec[0], spec[1], _abi.EPI_BIAS_BF16 if r16 else _abi.EPI_STORE_F32)
        else:
            identity = (x16 if r16 else x32).reshape(-1, x16.shape[-1])
        rkw = dict(resid=identity) if r16 else dict(resid_f32=identity, want_f32=True)
        spec = W[pre + ".branch2#c"]
        if IMPLICIT_CONV and out.shape[-1] % 32 and out.shape[-1] % 8 == 0 and out.is_contiguous():
            # 1x1x1 from 8 / 16 channels (fast pathway): the K padding to 32 lives in the tap table, not in a patch matrix
            y = kernels.conv_implicit(out, spec[0], spec[1], spec[2], spec[3], spec[4], True, **rkw)
            return (y, None) if r16 else (y[0], y[1].reshape(y[0].shape))
        a, (d, h, w) = self._cols(out, spec)
        shape = (x16.shape[0], d, h, w, spec[0].shape[0])
        y = kernels.conv_gemm(a, spec[0], spec[1], True, **rkw)
        return (y.reshape(shape), None) if r16 else (y[0].reshape(shape), y[1].reshape(shape))

    def _stem(self, x, spec, half, direct=None, mfma=None):
        B, C, T, H, W = x.shape
        wt, bias, k, stride, pad = spec
        if mfma is not None and STEM_MFMA:
            # fast pathway: 3 -> 8 channels, k 5x7x7 (the patch matrix would be 4.7 GB for 8 clips): clip packed to 4-channel
            # 16-bit rows, one MFMA k-slice per kernel row (1.51 -> 0.26 ms for 8 clips against the fp32 direct kernel)
            y = kernels.conv_stem_mfma(x.contiguous(), mfma, bias, k, stride, pad, True)
        elif direct is not None:
            y = kernels.conv_stem_direct(x, direct, bias, k, stride, pad, True, half)
        elif IMPLICIT_CONV and C <= 8 and k[0] == 1:
            # slow pathway: 3 -> 64 channels, k 1x7x7: implicit GEMM over the clip packed to 8 channels (no 147-column patch matrix)
            x8 = kernels.pack_channels_last8(x, (B, T, C, H, W), (C * T * H * W, H * W, T * H * W, W, 1), half)
            w8 = self.__dict__.setdefault("_stem8", {})
            key = (wt.data_ptr(), wt._version)
            if key not in w8:
                taps = k[1] * k[2]
                t8 = torch.zeros(wt.shape[0], -(-taps * 8 // 32) * 32, dtype=wt.dtype, device=wt.device)
                t8[:, :taps * 8].view(wt.shape[0], taps, 8)[:, :, :C] = wt[:, :taps * C].reshape(wt.shape[0], taps, C)
                w8.clear()
                w8[key] = t8
            y = kernels.conv_implicit(x8.reshape(B, T, H, W, 8), w8[key], bias, k, stride, pad, True)
        else:
            a, (d, h, w) = kernels.im2col_nd(x, (B, C, T, H, W), (C * T * H * W, T * H * W, H * W, W, 1), k, stride, pad,
                                             half, wt.shape[1])
            y = kernels.conv_gemm(a, wt, bias, True).reshape(B, d, h, w, wt.shape[0])
        return kernels.pool_nd(y, (1, 3, 3), (1, 2, 2), (0, 1, 1), True)

    # ---- the whole network as ONE C call (csrc/convnet.hip): layer table built once per geometry -------------------------
    def _net(self, B, T, H, W, device):
        """kvq_convnet plan of blocks 0-4 + the head pools for (B, 3, T, H, W) clips: pathway packing = a frame-select launch,
        the lateral connections' torch.cat = the producing launches write at a channel offset of the wider tensor, residual
        stream 16-bit.  One plan + workspace per (geometry, stream): forwards on different streams may overlap."""
        import ctypes as C
        Wt = self._weights(device)
        key = (B, T, H, W, str(device), self.operand_dtype, _abi.current_stream(), id(Wt), self.two_lanes, self.split_k, self.head_small_grid)
        hit = self.__dict__.setdefault("_nets", {}).get(key)
        if hit is not None:
            return hit
        fe = "feature_extraction."
        tens, ops, keep, descs = [], [], [], []
        lane = [0]            # ops built while lane[0] == 1 run on the plan's second stream (the fast pathway)
        FAST_LANE = 1 if self.two_lanes else 0

        def tensor(b, d, h, w, c, kind=_abi.NET_T_ACT16):
            tens.append((b, d, h, w, c, kind))
            return len(tens) - 1

        def op(kind, src, dst, k=(1, 1, 1), st=(1, 1, 1), pd=(0, 0, 0), **kw):
            o = _abi.KvqNetOp()
            o.kind, o.src, o.dst, o.src2, o.dst32 = kind, src, dst, kw.get("src2", -1), -1
            o.kernel3[:], o.stride3[:], o.pad3[:] = tuple(k), tuple(st), tuple(pd)
            o.lane = lane[0]
            for f in ("cout", "kpad", "relu", "is_max", "dst_coff", "per_frame", "mean_off", "std_off", "out_stride", "n_index"):
                if f in kw:
                    setattr(o, f, kw[f])
            for f in ("w", "bias"):
                if kw.get(f) is not None:
                    setattr(o, f, kw[f].data_ptr())
            if "t_index" in kw:
                arr = (C.c_int32 * len(kw["t_index"]))(*kw["t_index"])
                keep.append(arr)
                o.t_index, o.n_index = C.cast(arr, C.c_void_p), len(kw["t_index"])
            ops.append(o)
            descs.append(dict(kind=kind, name=kw.get("name", ""), src=tens[src][:5], k=tuple(k), stride=tuple(st),
                              cout=kw.get("cout", 0), kpad=kw.get("kpad", 0)))

        def odim(n, k, st, pd):
            return (n + 2 * pd - k) // st + 1

        def conv(src, key, dst=None, coff=0, relu=True, src2=-1):
            wt, bias, k, st, pd = Wt[key]
            b, d, h, w, _, _ = tens[src]
            if dst is None:
                dst = tensor(b, odim(d, k[0], st[0], pd[0]), odim(h, k[1], st[1], pd[1]), odim(w, k[2], st[2], pd[2]), wt.shape[0])
            op(_abi.NET_CONV, src, dst, k, st, pd, cout=wt.shape[0], kpad=wt.shape[1], relu=int(relu), dst_coff=coff, src2=src2, w=wt, bias=bias,
               name=key.replace(fe, ""))
            descs[-1]["M"] = tens[dst][0] * tens[dst][1] * tens[dst][2] * tens[dst][3]
            return dst

        fast_in = tensor(B, T, H, W, 3, _abi.NET_T_F32_PLANAR)                       # slot 0: the caller's clips
        t_sel = [int(v) for v in torch.linspace(0, T - 1, T // 4).long().tolist()]
        Hs, Ws = odim(H, 7, 2, 3), odim(W, 7, 2, 3)
        Hp, Wp = odim(Hs, 3, 2, 1), odim(Ws, 3, 2, 1)
        # stems: slow 3 -> 64 (1x7x7), fast 3 -> 8 (5x7x7), each followed by the (1,3,3) max-pool
        wt, bias, k, st, pd = Wt[fe + "0.multipathway_blocks.0"]
        slow = tensor(B, T // 4, Hp, Wp, 64 + 2 * FAST_C[0])
        if STEM_POOL and tuple(k) == (1, 7, 7) and tuple(st) == (1, 2, 2) and tuple(pd) == (0, 3, 3) and wt.shape[0] == 64 and W <= 224 and W % 4 == 0:
            # frame selection + stem + max-pool in one launch straight from the fp32 clip (no slow clip, no 64-channel stem map in HBM)
            wimg = kernels.stem64_pack_weight(wt, wt.dtype)
            keep.append(wimg)
            op(_abi.NET_STEM64_POOL, fast_in, slow, k, st, pd, cout=64, relu=1, w=wimg, bias=bias, t_index=t_sel, dst_coff=0)
        else:
            slow_in = tensor(B, T // 4, H, W, 3, _abi.NET_T_F32_PLANAR)
            op(_abi.NET_SELECT_T, fast_in, slow_in, t_index=t_sel)
            taps = k[1] * k[2]
            w8 = torch.zeros(wt.shape[0], -(-taps * 8 // 32) * 32, dtype=wt.dtype, device=device)
            w8[:, :taps * 8].view(wt.shape[0], taps, 8)[:, :, :3] = wt[:, :taps * 3].reshape(wt.shape[0], taps, 3)
            keep.append(w8)
            s_stem = tensor(B, T // 4, Hs, Ws, 64)
            op(_abi.NET_STEM8, slow_in, s_stem, k, st, pd, cout=64, kpad=w8.shape[1], relu=1, w=w8, bias=bias)
            op(_abi.NET_POOL, s_stem, slow, (1, 3, 3), (1, 2, 2), (0, 1, 1), is_max=1, dst_coff=0)
        _, fbias, fk, fst, fpd = Wt[fe + "0.multipathway_blocks.1"]
        lane[0] = FAST_LANE
        fast = tensor(B, T, Hp, Wp, 8)
        if STEM_POOL and tuple(fk[1:]) == (7, 7) and tuple(fst) == (1, 2, 2) and tuple(fpd) == (fk[0] // 2, 3, 3) and W <= 256 and W % 4 == 0:
            # stem + max-pool in one launch straight from the fp32 clip: neither the packed clip nor the stem map touches HBM
            op(_abi.NET_STEM_POOL, fast_in, fast, fk, fst, fpd, cout=8, relu=1, w=Wt[fe + "0.multipathway_blocks.1/mfma"], bias=fbias)
        else:
            f_stem = tensor(B, T, Hs, Ws, 8)
            op(_abi.NET_STEM_MFMA, fast_in, f_stem, fk, fst, fpd, cout=8, relu=1, w=Wt[fe + "0.multipathway_blocks.1/mfma"], bias=fbias)
            op(_abi.NET_POOL, f_stem, fast, (1, 3, 3), (1, 2, 2), (0, 1, 1), is_max=1)
        conv(fast, fe + "0.multipathway_fusion", dst=slow, coff=64)
        lane[0] = 0
        slow_out = (SLOW["out"], FAST["out"])
        for si in range(4):
            for pi in (0, 1):
                lane[0] = FAST_LANE if pi == 1 else 0
                x = slow if pi == 0 else fast
                for bi in range(DEPTHS[si]):
                    pre = fe + f"{si + 1}.multipathway_blocks.{pi}.res_blocks.{bi}"
                    if pi == 1 and FUSE_FAST:                # the whole residual block in one launch where the kernel is built
                        cin, ci, cout = tens[x][4], Wt[pre + ".branch2#a"][0].shape[0], Wt[pre + ".branch2#c"][0].shape[0]
                        proj = int(bi == 0)
                        sb = tuple(Wt[pre + ".branch2#b"][3])
                        if sb[0] == 1 and sb[1] == sb[2] and _abi.lib().kvq_fast_bottleneck_pack_bytes(cin, ci, cout, proj, sb[1]):
                            blob = self._bottleneck_pack(Wt, pre, proj, sb[1])
                            keep.append(blob)
                            bb_, d_, h_, w_ = tens[x][:4]
                            ho_, wo_ = -(-h_ // sb[1]), -(-w_ // sb[1])
                            y = tensor(bb_, d_, ho_, wo_, cout)
                            op(_abi.NET_BOTTLENECK, x, y, st=sb, cout=cout, kpad=ci, n_index=proj, w=blob, name=pre.replace(fe, "") + " (fused block)")
                            m_in, m_ = bb_ * d_ * h_ * w_, bb_ * d_ * ho_ * wo_
                            descs[-1]["flops"] = 2.0 * (m_in * 3 * cin * ci + m_ * (9 * ci * ci + ci * cout + (cin * cout if proj else 0)))
                            descs[-1]["M"] = m_
                            x = y
                            continue
                    if pi == 0 and FUSE_SLOW and bi > 0 and RESIDUAL16:
                        cin, ci, cout = tens[x][4], Wt[pre + ".branch2#a"][0].shape[0], Wt[pre + ".branch2#c"][0].shape[0]
                        ka, kb, sb = tuple(Wt[pre + ".branch2#a"][2]), tuple(Wt[pre + ".branch2#b"][2]), tuple(Wt[pre + ".branch2#b"][3])
                        if ka == (1, 1, 1) and kb == (1, 3, 3) and sb == (1, 1, 1) and _abi.lib().kvq_slow_bottleneck_pack_bytes(cin, ci, cout):
                            skey = pre + "/sneck"
                            if skey not in Wt:
                                Wt[skey] = pack_slow_bottleneck(*Wt[pre + ".branch2#a"][:2], *Wt[pre + ".branch2#b"][:2], *Wt[pre + ".branch2#c"][:2])
                            keep.append(Wt[skey])
                            bb_, d_, h_, w_ = tens[x][:4]
                            last = bi == DEPTHS[si] - 1
                            y = tensor(bb_, d_, h_, w_, SLOW["out"][si] + 2 * FAST_C[si + 1] if (last and si < 3) else cout)
                            op(_abi.NET_BOTTLENECK_S, x, y, cout=cout, kpad=ci, w=Wt[skey], name=pre.replace(fe, "") + " (fused block)")
                            m_ = bb_ * d_ * h_ * w_
                            descs[-1]["flops"] = 2.0 * m_ * (cin * ci + 9 * ci * ci + ci * cout)
                            descs[-1]["M"] = m_
                            x = y
                            continue
                    a = conv(x, pre + ".branch2#a")
                    b = conv(a, pre + ".branch2#b")
                    ident = conv(x, pre + "#1", relu=False) if bi == 0 else x
                    last = bi == DEPTHS[si] - 1
                    if pi == 0 and last and si < 3:          # the stage output IS the first Cs channels of the next stage's input
                        bb, d, h, w, _, _ = tens[b]
                        wide = tensor(bb, d, h, w, SLOW["out"][si] + 2 * FAST_C[si + 1])
                        x = conv(b, pre + ".branch2#c", dst=wide, coff=0, src2=ident)
                    else:
                        x = conv(b, pre + ".branch2#c", src2=ident)
                if pi == 0:
                    slow = x
                else:
                    fast = x
            if si < 3:
                conv(fast, fe + f"{si + 1}.multipathway_fusion", dst=slow, coff=SLOW["out"][si])
            lane[0] = 0
        # head: AvgPool3d((8,7,7)) / ((32,7,7)), stride 1 + AdaptiveAvgPool3d(1).  On the 32 x 224 x 224 geometry the grid IS the
        # kernel and the two are one global mean; on a larger grid (--resize 256: 8 x 8 x 8) the real pool runs first
        for pi, x in ((0, slow), (1, fast)):
            lane[0] = FAST_LANE if pi == 1 else 0
            bb, d, h, w, c, _ = tens[x]
            if check_head_grid(pi, (d, h, w), self.head_small_grid):
                k = head_pool_kernel(pi, T)
                pooled = tensor(bb, d - k[0] + 1, h - k[1] + 1, w - k[2] + 1, c)
                op(_abi.NET_POOL, x, pooled, k, (1, 1, 1), (0, 0, 0), is_max=0)
                x = pooled
            op(_abi.NET_MEAN_STD, x, pi, per_frame=0, mean_off=0, std_off=-1, out_stride=c)
        lane[0] = 0
        ta = (_abi.KvqNetTensor * len(tens))()
        for i, (b, d, h, w, c, kind) in enumerate(tens):
            ta[i].B, ta[i].D, ta[i].H, ta[i].W, ta[i].C, ta[i].kind = b, d, h, w, c, kind
        oa = (_abi.KvqNetOp * len(ops))(*ops)
        handle = C.c_void_p()
        _abi.check(_abi.lib().kvq_convnet_create(oa, len(ops), ta, len(tens), 1, 2, self.operand_dtype, C.byref(handle)), "kvq_convnet_create")
        if not self.split_k:
            _abi.check(_abi.lib().kvq_convnet_splitk(handle, 0), "kvq_convnet_splitk")
        ws = torch.empty(_abi.lib().kvq_convnet_workspace_bytes(handle), dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)       # tap tables / packed weights were built on this stream; other streams may run the plan
        entry = (handle, ws, (tens[slow][4], tens[fast][4]), keep, Wt, descs)
        self._nets[key] = entry
        return entry

    def profile_layers(self, fast_in):
        """measurement only: one forward with every op of the plan bracketed by HIP events -> [{name, kind, M, N, K, ms, tflops}]"""
        import ctypes as C
        B, _, T, H, W = fast_in.shape
        handle, *_rest, descs = self._net(B, T, H, W, fast_in.device)
        _abi.check(_abi.lib().kvq_convnet_profile(handle, 1), "kvq_convnet_profile")
        try:
            self.forward_clips(fast_in)
            ms = (C.c_float * len(descs))()
            n = C.c_int32(0)
            _abi.check(_abi.lib().kvq_convnet_profile_read(handle, ms, len(descs), C.byref(n)), "kvq_convnet_profile_read")
        finally:
            _abi.check(_abi.lib().kvq_convnet_profile(handle, 0), "kvq_convnet_profile")
        kinds = {_abi.NET_CONV: "conv", _abi.NET_POOL: "pool", _abi.NET_STEM8: "stem8", _abi.NET_STEM_MFMA: "stem_mfma", _abi.NET_STEM_POOL: "stem_pool", _abi.NET_STEM64_POOL: "stem64_pool",
                 _abi.NET_MEAN_STD: "mean", _abi.NET_SELECT_T: "select_t", _abi.NET_BOTTLENECK: "bottleneck", _abi.NET_BOTTLENECK_S: "bottleneck"}
        out = []
        for d, t in zip(descs, ms):
            k = d["k"][0] * d["k"][1] * d["k"][2] * d["src"][4]
            fl = d.get("flops", 2.0 * d.get("M", 0) * d["cout"] * k)
            out.append(dict(name=d["name"], kind=kinds.get(d["kind"], "?"), M=d.get("M", 0), N=d["cout"], K=k, kernel=d["k"], ms=float(t),
                            tflops=fl / (t * 1e9) if t > 0 and fl else 0.0))
        return out

    def __del__(self):
        try:
            for handle, *_ in self.__dict__.get("_nets", {}).values():
                _abi.lib().kvq_convnet_destroy(handle)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    @staticmethod
    def _one_call(fast_in):
        return CONVNET and STEM_MFMA and RESIDUAL16 and fast_in.shape[1] == 3 and fast_in.shape[2] % 4 == 0

    def forward_clips(self, fast_in):
        """(B,3,T,H,W) fp32 clips on a HIP device -> (slow_feature, fast_feature), the same values as
        ``forward(pack_pathway_output(clips))``: ONE C call enqueues the whole network, the slow pathway's frames are selected
        from the clip on the device (pack_pathway_output's indices, SlowFast_features.py:112-135) — no torch kernel runs."""
        import ctypes as C
        if not fast_in.is_cuda:
            raise _abi.KvqError("slowfast.forward_clips needs the clips on a HIP device; there is no CPU path")
        if not self._one_call(fast_in):
            return self.forward(pack_pathway_output(fast_in))
        fast_in = fast_in.float().contiguous()
        B, _, T, H, W = fast_in.shape
        handle, ws, (cs, cf), *_ = self._net(B, T, H, W, fast_in.device)
        s_out = torch.empty(B, cs, dtype=torch.float32, device=fast_in.device)
        f_out = torch.empty(B, cf, dtype=torch.float32, device=fast_in.device)
        ins = (C.c_void_p * 1)(fast_in.data_ptr())
        outs = (C.c_void_p * 2)(s_out.data_ptr(), f_out.data_ptr())
        _abi.check(_abi.lib().kvq_convnet_forward(handle, ins, outs, ws.data_ptr(), ws.numel(), _abi.stream_of(fast_in)),
                   "kvq_convnet_forward")
        return s_out.reshape(B, cs, 1, 1, 1), f_out.reshape(B, cf, 1, 1, 1)

    def forward(self, x):
        """x = [slow (B,3,T/4,H,W), fast (B,3,T,H,W)] fp32 on a HIP device (``pack_pathway_output``)
        -> (slow_feature (B,2048,1,1,1), fast_feature (B,256,1,1,1))."""
        slow_in, fast_in = x
        if not fast_in.is_cuda:
            raise _abi.KvqError("slowfast.forward needs the clips on a HIP device; there is no CPU path")
        if self._one_call(fast_in) and _is_packed_pair(slow_in, fast_in):      # slow_in IS pack_pathway_output's selection of fast_in
            return self.forward_clips(fast_in)
        W = self._weights(fast_in.device)
        half = _abi.torch_dtype(self.operand_dtype)
        fe = "feature_extraction."
        slow = self._stem(slow_in.float().contiguous(), W[fe + "0.multipathway_blocks.0"], half)
        fast = self._stem(fast_in.float().contiguous(), W[fe + "0.multipathway_blocks.1"], half,
                          W.get(fe + "0.multipathway_blocks.1/direct"), W.get(fe + "0.multipathway_blocks.1/mfma"))
        slow = torch.cat([slow, self._conv_relu(fast, W[fe + "0.multipathway_fusion"])], dim=-1)
        s32 = f32 = None
        for si in range(4):
            for bi in range(DEPTHS[si]):
                slow, s32 = self._res_block(slow, s32, W, fe + f"{si + 1}.multipathway_blocks.0.res_blocks.{bi}", bi == 0)
            for bi in range(DEPTHS[si]):
                fast, f32 = self._res_block(fast, f32, W, fe + f"{si + 1}.multipathway_blocks.1.res_blocks.{bi}", bi == 0)
            if si < 3:
                slow = torch.cat([slow, self._conv_relu(fast, W[fe + f"{si + 1}.multipathway_fusion"])], dim=-1)
        # AvgPool3d((8,7,7)) / ((32,7,7)), stride 1 + AdaptiveAvgPool3d(1): a global mean when the grid is the kernel
        outs = []
        for pi, y in enumerate((slow, fast)):
            if check_head_grid(pi, y.shape[1:4], self.head_small_grid):
                y = kernels.pool_nd(y.contiguous(), head_pool_kernel(pi, None), (1, 1, 1), (0, 0, 0), False)
            B, D, H, Wd, C = y.shape
            o = torch.empty(B, C, dtype=torch.float32, device=y.device)
            kernels.mean_std_pool(y.reshape(B, D * H * Wd, C), o, 0, -1)
            outs.append(o.reshape(B, C, 1, 1, 1))
        return outs[0], outs[1]
