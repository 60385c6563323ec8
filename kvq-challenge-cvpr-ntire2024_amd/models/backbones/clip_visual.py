"""Host-side mirror of KSVQE's ``CLIP_tool``: the reference's ``CLIP_extractor_addadapter_cls``
(``models/backbones/CLIP_backbone.py:115-201``) around the vendored CLIP vision transformer
(``models/backbones/clip/model.py:252-267``) — same module tree and ``state_dict`` keys (``visual.conv1.weight``,
``visual.class_embedding``, ``visual.positional_embedding``, ``visual.ln_pre.*``,
``visual.transformer.resblocks.{i}.{attn.in_proj_*, attn.out_proj.*, ln_1.*, mlp.c_fc.*, mlp.c_proj.*, ln_2.*}``,
``visual.ln_post.*``, ``visual.proj``, ``adapter_layer.{j}.{0,2}.*``), same forward signature and outputs
``(cls_attn (B, h·w), cls_token (B, 1, D), pat_token (1, B, h·w, D))``.

Execution: the torch modules below only HOLD parameters.  The forward runs on the HIP kernels of ``libkvq_hip.so``:
16×16 patch gather + MFMA GEMM, token assembly + ``ln_pre`` (``kvq_vit_embed_ln``), per block LayerNorm → in_proj GEMM →
short-sequence attention (``kvq_mha_small``) → out_proj GEMM accumulating into the fp32 residual stream → LayerNorm →
c_fc GEMM with the QuickGELU epilogue → c_proj GEMM into the residual stream; from ``CLIP_location`` on the CLS adapter
(two ReLU GEMMs on the gathered CLS rows) mixed back 0.5 / 0.5; finally the CLS-to-patch cosine map.  This is one of the
KSVQE-only modules; the rest of KSVQE is not built yet (``VQA_Network`` key ``KSVQE`` raises).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import _abi, kernels


class _Block(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(width, heads)         # parameter container: in_proj_weight/bias, out_proj
        self.ln_1 = nn.LayerNorm(width)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(width, 4 * width))
        self.mlp.add_module("c_proj", nn.Linear(4 * width, width))
        self.ln_2 = nn.LayerNorm(width)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[_Block(width, heads) for _ in range(layers)])


class _Visual(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        self.grid_size = input_resolution // patch_size
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class CLIP_extractor_addadapter_cls(nn.Module):  # noqa: N801  (reference spelling)
    def __init__(self, visual=None, CLIP_location=8, cls_use=True, width=768, layers=12, heads=12, patch_size=16,
                 input_resolution=224, output_dim=512):
        super().__init__()
        self.visual = visual if visual is not None else _Visual(input_resolution, patch_size, width, layers, heads, output_dim)
        self.embed_dim = self.visual.transformer.width
        self.heads = heads
        self.prompt_token_num = 1
        self.cls_use, self.CLIP_location = cls_use, CLIP_location
        if cls_use:
            self.adapter_layer = nn.ModuleList([
                nn.Sequential(nn.Linear(self.embed_dim, self.embed_dim // 4), nn.ReLU(inplace=True),
                              nn.Linear(self.embed_dim // 4, self.embed_dim), nn.ReLU(inplace=True))
                for _ in range(11 - CLIP_location + 1)])
        self.grid_size = self.visual.grid_size
        self.operand_dtype = _abi.dtype_code(os.environ.get("KVQ_OPERAND_DTYPE", "fp16"))
        self._wcache = None
        self._pos = {}

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        if self.cls_use:
            for p in self.adapter_layer.parameters():
                p.requires_grad = True
        return {n for n, p in self.named_parameters() if p.requires_grad}

    # ------------------------------------------------------------------ weights (16-bit GEMM operands, fp32 vectors)
    def _weights(self, device):
        sig = (self.operand_dtype,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._wcache is not None and self._wcache[0] == sig:
            return self._wcache[1]
        half = _abi.torch_dtype(self.operand_dtype)

        def h(t):
            t = t.detach().to(device, torch.float32)
            if half == torch.float16:
                t = t.clamp(-65504.0, 65504.0)
            return t.to(half).contiguous()

        def f(t):
            return t.detach().to(device, torch.float32).contiguous()

        v = self.visual
        w = {"conv": h(v.conv1.weight.reshape(v.conv1.weight.shape[0], -1)), "cls": f(v.class_embedding),
             "ln_pre": (f(v.ln_pre.weight), f(v.ln_pre.bias)), "blocks": [], "adapters": []}
        for blk in v.transformer.resblocks:
            w["blocks"].append(dict(
                ln1=(f(blk.ln_1.weight), f(blk.ln_1.bias)), ln2=(f(blk.ln_2.weight), f(blk.ln_2.bias)),
                win=h(blk.attn.in_proj_weight), bin=f(blk.attn.in_proj_bias),
                wout=h(blk.attn.out_proj.weight), bout=f(blk.attn.out_proj.bias),
                wfc=h(blk.mlp.c_fc.weight), bfc=f(blk.mlp.c_fc.bias), wpr=h(blk.mlp.c_proj.weight), bpr=f(blk.mlp.c_proj.bias)))
        if self.cls_use:
            for ad in self.adapter_layer:
                w["adapters"].append((h(ad[0].weight), f(ad[0].bias), h(ad[2].weight), f(ad[2].bias)))
        self._wcache, self._pos = (sig, w), {}
        return w

    def _pos_embed(self, hw, device):
        """``resize_pos_embed2d`` (CLIP_backbone.py:35-70): bicubic resize of the grid rows when the token grid differs;
        a weight-side transformation, cached per grid."""
        key = (hw, str(device), self.visual.positional_embedding._version)
        p = self._pos.get(key)
        if p is None:
            pos = self.visual.positional_embedding.detach().to(device, torch.float32)
            g = self.grid_size
            if (g, g) != hw:
                grid = torch.nn.functional.interpolate(pos[1:].t().reshape(1, -1, g, g), size=hw, mode="bicubic", align_corners=False)
                pos = torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(hw[0] * hw[1], -1)], 0)
            p = self._pos[key] = pos.contiguous()
        return p

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        """x (B, 3, H, W) fp32 on a HIP device, H and W multiples of the patch size."""
        if not x.is_cuda:
            raise _abi.KvqError("CLIP_extractor_addadapter_cls.forward needs the frames on a HIP device; there is no CPU path")
        B, Cin, H, W = x.shape
        ps = self.visual.conv1.kernel_size[0]
        if H % ps or W % ps:
            raise _abi.KvqError(f"frame size {H}x{W} must be a multiple of the {ps}x{ps} patch")
        w = self._weights(x.device)
        half = _abi.torch_dtype(self.operand_dtype)
        hh, ww = H // ps, W // ps
        D, heads = self.embed_dim, self.heads
        cols = kernels.patch_im2col(x.to(torch.float32).reshape(B, Cin, 1, H, W).contiguous(), (1, ps, ps), out_dtype=half)
        tok = kernels.gemm(cols, w["conv"], None, _abi.EPI_STORE_F32)                       # conv1 (no bias)
        xr = kernels.vit_embed_ln(tok, w["cls"], self._pos_embed((hh, ww), x.device), *w["ln_pre"], B)
        L = hh * ww + 1
        x2 = xr.reshape(B * L, D)
        for i, bw in enumerate(w["blocks"]):
            a = kernels.layernorm_rows(x2, *bw["ln1"], out_dtype=half)
            qkv = kernels.gemm(a, bw["win"], bw["bin"], _abi.EPI_BIAS_BF16)
            att = kernels.mha_small(qkv, B, L, heads)
            kernels.gemm(att, bw["wout"], bw["bout"], _abi.EPI_RESID_F32, out=x2)       # x = x + attention(ln_1(x))
            m = kernels.layernorm_rows(x2, *bw["ln2"], out_dtype=half)
            m = kernels.gemm(m, bw["wfc"], bw["bfc"], _abi.EPI_QGELU_BF16)
            kernels.gemm(m, bw["wpr"], bw["bpr"], _abi.EPI_RESID_F32, out=x2)           # x = x + mlp(ln_2(x))
            if i >= self.CLIP_location and self.cls_use:
                w0, b0, w2, b2 = w["adapters"][i - self.CLIP_location]
                c = kernels.cls_gather(xr, half)
                c = kernels.conv_gemm(kernels.conv_gemm(c, w0, b0, True), w2, b2, True)
                kernels.cls_mix(xr, c, 0.5)
        return kernels.cosine_cls(xr), xr[:, :1], xr[:, 1:].unsqueeze(0)


def build_CLIPmodel_basedadapter_cls(backbone_name="ViT-B/16", CLIP_location=None, cls_use=None):  # noqa: N802
    """Reference builder (CLIP_backbone.py:204-214) minus the checkpoint download: a randomly initialised ViT-B/16 tower;
    load the CLIP weights with ``load_state_dict`` (keys as in the reference)."""
    if backbone_name != "ViT-B/16":
        raise NotImplementedError(backbone_name)
    m = CLIP_extractor_addadapter_cls(CLIP_location=CLIP_location, cls_use=cls_use)
    m.freeze()
    return m
