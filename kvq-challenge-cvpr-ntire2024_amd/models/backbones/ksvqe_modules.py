"""Host-side mirrors of KSVQE's content-distortion modulation (CDM) modules, ``models/backbones/KSVQE_model.py``:
``crossattention1`` (:1553-1586), ``Attention`` (:1508-1551), ``Semantic_Transformation2`` (:817-835) and
``Dist_Transformation3`` (:934-960) — the four the trainer's KSVQE instantiates per tuned stage (:1160-1186) — with the
reference's constructor arguments, ``state_dict`` keys, argument layouts and outputs.  The torch modules only hold
parameters; the arithmetic runs on ``libkvq_hip.so`` (GEMMs, ``kvq_mha_cross``, ``kvq_mean_std_pool``, the modulation
kernels).  Also here: key-frame selection, the QRS region selection (``RegionNet_CLIP``, eval path) and the CONTRIQUE
distortion branch.  Part of SURVEY.md §8 f1; ``KSVQE_model.py`` composes them into the reference's ``KSVQE.forward``."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import _abi, kernels


class _HipModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.operand_dtype = _abi.dtype_code(os.environ.get("KVQ_OPERAND_DTYPE", "fp16"))
        self._wc = None

    def _half(self):
        return _abi.torch_dtype(self.operand_dtype)

    def _cached(self, device, build):
        sig = (self.operand_dtype, str(device)) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._wc is None or self._wc[0] != sig:
            self._wc = (sig, build())
        return self._wc[1]

    def _w16(self, t, device):
        t = t.detach().to(device, torch.float32)
        if self._half() == torch.float16:
            t = t.clamp(-65504.0, 65504.0)
        return t.to(self._half()).contiguous()

    @staticmethod
    def _need(*ts):
        for t in ts:
            if not t.is_cuda:
                raise _abi.KvqError("this module needs its inputs on a HIP device; there is no CPU path")


class crossattention1(_HipModule):  # noqa: N801  (reference spelling)
    """Q (B, Nq, C) attends K (B, Nk, C): heads of C / num_heads = 64 channels, logits scaled by C^-0.5 (the full width, as
    the reference does), no output projection.  Returns ``(O, None)``: the reference's second output (the head-averaged
    attention map) is discarded by every caller (KSVQE_model.py:1451, :1471) and is not produced here."""

    def __init__(self, dim, num_heads, ln=False):
        super().__init__()
        self.dim_V, self.num_heads = dim, num_heads
        self.fc_q, self.fc_k, self.fc_v = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)

    def forward(self, Q, K, keep16=False):
        """``keep16``: return the 16-bit attention output as it leaves the kernel (what the next GEMM would convert to)."""
        self._need(Q, K)
        dev, h = Q.device, self._half()
        w = self._cached(dev, lambda: [(self._w16(m.weight, dev), m.bias.detach().to(dev, torch.float32).contiguous())
                                       for m in (self.fc_q, self.fc_k, self.fc_v)])
        B, Nq, C = Q.shape
        # operands already in the 16-bit operand type (KSVQE.forward hands the adapters' GEMM outputs and one shared copy of
        # the frame tokens over) are used as they are: same values the conversion would produce
        q16 = Q.contiguous().reshape(-1, C) if Q.dtype == h else kernels.to_half(Q.to(torch.float32).contiguous().reshape(-1, C), h)
        k16 = K.contiguous().reshape(-1, C) if K.dtype == h else kernels.to_half(K.to(torch.float32).contiguous().reshape(-1, C), h)
        q = kernels.gemm(q16, *w[0], _abi.EPI_BIAS_BF16)
        k = kernels.gemm(k16, *w[1], _abi.EPI_BIAS_BF16)
        v = kernels.gemm(k16, *w[2], _abi.EPI_BIAS_BF16)
        o = kernels.mha_cross(q, k, v, B, self.num_heads, float(self.dim_V) ** -0.5)
        if keep16:
            return o.reshape(B, Nq, C), None
        return kernels.to_float(o).reshape(B, Nq, C), None


class Attention(_HipModule):
    """Self-attention over x (B, n, C) with a bias-free qkv projection, head_dim^-0.5 scaling and an output Linear."""

    def __init__(self, dim, heads=8, dropout=0.0):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Dropout(dropout))

    def forward(self, x, mask=None, keep16=False):
        if mask is not None:
            raise NotImplementedError("key mask: no caller passes one (KSVQE_model.py:1473)")
        self._need(x)
        dev, h = x.device, self._half()
        w = self._cached(dev, lambda: (self._w16(self.to_qkv.weight, dev), self._w16(self.to_out[0].weight, dev),
                                       self.to_out[0].bias.detach().to(dev, torch.float32).contiguous()))
        B, n, C = x.shape
        x16 = x.contiguous().reshape(-1, C) if x.dtype == h else kernels.to_half(x.to(torch.float32).contiguous().reshape(-1, C), h)
        qkv = kernels.gemm(x16, w[0], None, _abi.EPI_BIAS_BF16)
        o = kernels.mha_small(qkv, B, n, self.heads)
        y = kernels.gemm(o, w[1], w[2], _abi.EPI_BIAS_BF16)
        return y.reshape(B, n, C) if keep16 else kernels.to_float(y).reshape(B, n, C)


class Semantic_Transformation2(_HipModule):  # noqa: N801
    """x, input (N, C, h, w): gama = sigmoid(conv1x1_{C->1}(x)), beta = conv1x1_{C->1}(x); returns gama * input + beta."""

    def __init__(self, inChannels):  # noqa: N803
        super().__init__()
        self.conv_gama = nn.Conv2d(inChannels, 1, 1, padding=0, stride=1)
        self.conv_beta = nn.Conv2d(inChannels, 1, 1, padding=0, stride=1)

    def forward(self, x, input):  # noqa: A002  (reference argument name)
        self._need(x, input)
        dev = x.device
        w = self._cached(dev, lambda: (self.conv_gama.weight.detach().to(dev, torch.float32).reshape(-1).contiguous(),
                                       float(self.conv_gama.bias.detach()), self.conv_beta.weight.detach().to(dev, torch.float32)
                                       .reshape(-1).contiguous(), float(self.conv_beta.bias.detach())))
        N, C, hh, ww = x.shape
        rows = lambda t: t.to(torch.float32).permute(0, 2, 3, 1).reshape(N * hh * ww, C).contiguous()   # noqa: E731  (layout only)
        out = kernels.sem_modulate(rows(x), rows(input), *w)
        return out.reshape(N, hh, ww, C).permute(0, 3, 1, 2)


class Dist_Transformation3(_HipModule):  # noqa: N801
    """x (B, C, T, H, W), input (B, T·H·W, C): gamma = sigmoid(Linear(std_{THW} x)) (unbiased std), beta = Linear(mean_{THW} x);
    returns gamma[:, None] * input + beta[:, None]."""

    def __init__(self, inChannels):  # noqa: N803
        super().__init__()
        self.get_gamma, self.get_beta = nn.Linear(inChannels, inChannels), nn.Linear(inChannels, inChannels)

    def forward(self, x, input):  # noqa: A002
        self._need(x, input)
        dev, h = x.device, self._half()
        w = self._cached(dev, lambda: (self._w16(self.get_gamma.weight, dev), self.get_gamma.bias.detach().to(dev, torch.float32).contiguous(),
                                       self._w16(self.get_beta.weight, dev), self.get_beta.bias.detach().to(dev, torch.float32).contiguous()))
        if x.dtype == h and x.dim() == 3:
            # (B, positions, C) 16-bit rows, any position order: the statistics are over all positions of a sample
            B, C = x.shape[0], x.shape[2]
            x16 = x.contiguous()
        else:
            B, C = x.shape[:2]
            x16 = kernels.to_half(x.to(torch.float32).reshape(B, C, -1).permute(0, 2, 1).contiguous(), h)  # (B, THW, C) channels-last
        stats = torch.empty(B, 2 * C, dtype=torch.float32, device=dev)
        kernels.mean_std_pool(x16, stats, 0, C)
        s16 = kernels.to_half(stats, h)
        g = kernels.gemm(s16[:, C:].contiguous(), w[0], w[1], _abi.EPI_BIAS_BF16)
        b = kernels.gemm(s16[:, :C].contiguous(), w[2], w[3], _abi.EPI_BIAS_BF16)
        return kernels.dist_modulate(input.to(torch.float32).contiguous(), g, b)


# ------------------------------------------------------------------------------------------------------------------
def obtain_keyframes(x):
    """``KSVQE.obtain_keyframes`` (KSVQE_model.py:1352-1376): x (b, c, t, h, w) -> (group_idx (b, t), key_frame (b, 4, c, h, w)):
    the frames 0, t/4-1, t/2-1, 3t/4-1; a frame's group id counts how many of the last three positions are <= its index."""
    b, c, t, h, w = x.shape
    pos = [0, t // 4 - 1, t // 2 - 1, t * 3 // 4 - 1]
    xt = x.permute(0, 2, 1, 3, 4)
    key = torch.stack([xt[:, p] for p in pos], 1)          # slices, not a host index list: no H2D copy (hipGraph-capturable)
    j = torch.arange(t, device=x.device)
    gid = sum((j >= p).to(x.dtype) for p in dict.fromkeys(pos[1:]))      # equal positions (tiny t) bump once, as the elif chain does
    return gid.unsqueeze(0).expand(b, t).contiguous(), key


def extend_by_group(per_key, group_id):
    """``extend_fullcls_attn`` / ``extend_fullcls_indices`` (KSVQE_model.py:1378-1387, patchnet.py:450-460): per_key (B, N_key, ...)
    -> (B, T, ...) with row t = per_key[b, group_id[b, t]] — an index_select instead of the reference's ``.item()`` double loop."""
    idx = group_id.long()
    return torch.gather(per_key, 1, idx.reshape(idx.shape + (1,) * (per_key.dim() - 2)).expand(idx.shape + per_key.shape[2:]))


class RegionNet_CLIP(nn.Module):  # noqa: N801
    """Eval path of the reference's quality-aware region selection (patchnet.py:390-550, ``sample_type`` other than
    'random'): the CLIP CLS-to-patch map of every key frame picks ONE window of sqrt(k) x sqrt(k) anchors (top-1 of the
    window means), frames take the window of their key frame, the window is cut out: x (b, c, t, h, w) -> (b, c, t,
    sqrt(k)·anchor, sqrt(k)·anchor).  Training-time samplers (perturbed top-k, Gumbel, multinomial, random) are not built."""

    def __init__(self, k, anchor_size, stride, num_samples=500, sample_type="topkpertubation"):
        super().__init__()
        self.k, self.stride, self.anchor_size, self.num_samples, self.sample_type = k, stride, anchor_size, num_samples, sample_type
        if stride != 1:
            raise NotImplementedError("window stride other than 1")

    def forward(self, x, score, sigma, group_id, extra_score=None):
        if self.training or self.sample_type == "random" or extra_score is not None:
            raise NotImplementedError("training-time / random region sampling and extra_score: this is an inference engine")
        if not x.is_cuda:
            raise _abi.KvqError("RegionNet_CLIP.forward needs its inputs on a HIP device; there is no CPU path")
        b, c, t, h, w = x.shape
        _, n_key, L = score.shape
        gs, kk = int(round(L ** 0.5)), int(round(self.k ** 0.5))
        idx = kernels.qrs_top_region(score.to(torch.float32).reshape(b * n_key, gs, gs).contiguous(), h // self.anchor_size,
                                     w // self.anchor_size, kk, kk)
        full = extend_by_group(idx.reshape(b, n_key), group_id).reshape(b * t).contiguous()
        return kernels.crop_regions(x.to(torch.float32).contiguous(), full, self.anchor_size, kk, kk)


# ------------------------------------------------------------------------------------------------------------------
def get_network(name, pretrained=False):
    """``get_network`` (KSVQE_model.py:1608-1620): only the ResNet-50 KSVQE asks for; randomly initialised (no downloads)."""
    if name != "resnet50":
        raise KeyError(f"{name}: only 'resnet50' is built")
    from .simpleVQA_model import TorchvisionResNet50
    return TorchvisionResNet50()


class CONTRIQUE_model(_HipModule):  # noqa: N801
    """The reference's distortion branch (KSVQE_model.py:1622-1664): every anchor x anchor patch of x (b, c, t, h, w) goes
    through the ResNet-50 trunk (``encoder`` = the network's children without avgpool / fc, same state_dict keys), the
    2048-vector is L2-normalised and projected (Linear -> BatchNorm1d -> ReLU -> Linear -> BatchNorm1d, eval statistics
    folded into the two GEMMs) to ``projection_dim``: returns (b, t, patches per frame, projection_dim) fp32."""

    def __init__(self, encoder, n_features, anchor_size=32, patch_dim=(2, 2), normalize=True, projection_dim=128, residual16=None):
        """``residual16`` (build option, not a reference argument): the ResNet trunk's residual stream in 16 bits (True: one more 16-bit
        rounding per bottleneck, the HBM-bound 1x1 convs move half the bytes — +9 % on the KSVQE forward) or in fp32 (False: rounds
        3-5).  None: KVQ_CONTRIQUE_R16 (default 1).  Both are pinned to the reference's stored output (tests/test_gpu_clip.py)."""
        super().__init__()
        self.anchor_size, self.normalize, self.n_features, self.patch_dim = anchor_size, normalize, n_features, patch_dim
        if not hasattr(encoder, "features"):
            raise TypeError("encoder must be the network get_network('resnet50') returns")
        self.encoder = nn.Sequential(*list(encoder.children())[:-2])
        object.__setattr__(self, "_net", encoder)              # the same modules, kept for their HIP forward (not re-registered)
        self.residual16 = (os.environ.get("KVQ_CONTRIQUE_R16", "1") != "0") if residual16 is None else bool(residual16)
        encoder.residual16 = self.residual16                   # the encoder object is this model's own (get_network builds one per call)
        self.projector = nn.Sequential(nn.Linear(n_features, n_features, bias=False), nn.BatchNorm1d(n_features), nn.ReLU(),
                                       nn.Linear(n_features, projection_dim, bias=False), nn.BatchNorm1d(projection_dim))

    def _fold(self, lin, bn, device):
        scale = bn.weight.detach().to(device, torch.float32) / torch.sqrt(bn.running_var.to(device, torch.float32) + bn.eps)
        bias = bn.bias.detach().to(device, torch.float32) - bn.running_mean.to(device, torch.float32) * scale
        return self._w16(lin.weight.detach().to(device, torch.float32) * scale[:, None], device), bias.contiguous()

    def forward(self, x):
        self._need(x)
        dev, h = x.device, self._half()
        self._net.operand_dtype = self.operand_dtype
        sig_extra = tuple((t.data_ptr(), t._version) for t in self.projector.buffers())
        w = self._cached(dev, lambda: (sig_extra, self._fold(self.projector[0], self.projector[1], dev),
                                       self._fold(self.projector[3], self.projector[4], dev)))
        if w[0] != sig_extra:                                    # running statistics changed: refold
            self._wc = None
            return self.forward(x)
        b, c, t, hh, ww = x.shape
        a = self.anchor_size
        gh, gw = hh // a, ww // a
        # (b t) c (gh a) (gw a) -> (b t gh gw) c a a: layout only
        p = (x.to(torch.float32).permute(0, 2, 1, 3, 4).reshape(b * t, c, gh, a, gw, a).permute(0, 2, 4, 1, 3, 5)
             .reshape(b * t * gh * gw, c, a, a).contiguous())
        _, f32 = self._net.features(p)
        f = f32.reshape(-1, self.n_features)
        f16 = kernels.l2_normalize_rows(f.contiguous(), h) if self.normalize else kernels.to_half(f.contiguous(), h)
        z = kernels.conv_gemm(f16, *w[1], True)
        z = kernels.gemm(z, *w[2], _abi.EPI_BIAS_BF16)
        return kernels.to_float(z).reshape(b, t, gh * gw, -1)
