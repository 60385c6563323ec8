"""ORACLE (test infrastructure, never shipped in the product path).

CPU fp32 restatement of the KSVQE "CLIP_tool": the reference's ``CLIP_extractor_addadapter_cls.forward``
(``models/backbones/CLIP_backbone.py:156-201``) over the vendored CLIP vision transformer
(``models/backbones/clip/model.py``: ``VisionTransformer`` :252-267, ``ResidualAttentionBlock`` :184-216 with
``nn.MultiheadAttention`` self-attention, ``QuickGELU`` :179-181, fp32 ``LayerNorm`` :171-176) and
``resize_pos_embed2d`` (:35-70).  Written as plain functional tensor arithmetic over the state_dict.

Pinned: ``tests/golden/make_golden.py clip`` instantiates the reference modules with the same synthetic weights and
checks this file against them, then stores the reference's outputs in ``tests/golden/clip.npz``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _t(v):
    return v if torch.is_tensor(v) else torch.from_numpy(v)


def resize_pos_embed(pos: torch.Tensor, src: int, tgt_hw) -> torch.Tensor:
    """(1 + src*src, C) -> (1 + h*w, C): the class row kept, the grid rows bicubically resized
    (align_corners=False, no antialias) when the token grid differs (CLIP_backbone.py:35-70)."""
    h, w = tgt_hw
    if (src, src) == (h, w):
        return pos
    grid = pos[1:].t().reshape(1, -1, src, src)
    grid = F.interpolate(grid, size=(h, w), mode="bicubic", align_corners=False)
    return torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(h * w, -1)], 0)


def attention(x: torch.Tensor, w_in, b_in, w_out, b_out, heads: int) -> torch.Tensor:
    """nn.MultiheadAttention(x, x, x) without mask / dropout on x (B, L, D): q scaled by head_dim^-0.5."""
    B, L, D = x.shape
    hd = D // heads
    q, k, v = F.linear(x, w_in, b_in).split(D, dim=-1)
    q = q.reshape(B, L, heads, hd).transpose(1, 2) * hd ** -0.5
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    return F.linear((p @ v).transpose(1, 2).reshape(B, L, D), w_out, b_out)


def clip_visual_extractor(x: torch.Tensor, params, heads: int = 12, clip_location: int = 8, cls_use: bool = True):
    """x (B, 3, H, W) fp32 -> (cls_attn (B, h*w), cls_token (B, 1, D), pat_token (1, B, h*w, D))."""
    p = {k: _t(v).float() for k, v in params.items()}
    wc = p["visual.conv1.weight"]
    D, patch = wc.shape[0], wc.shape[-1]
    t = F.conv2d(x.float(), wc, stride=patch)                         # (B, D, h, w)
    B, _, h, w = t.shape
    t = t.reshape(B, D, h * w).transpose(1, 2)
    t = torch.cat([p["visual.class_embedding"].expand(B, 1, D), t], 1)
    src = int(round((p["visual.positional_embedding"].shape[0] - 1) ** 0.5))
    t = t + resize_pos_embed(p["visual.positional_embedding"], src, (h, w))
    t = F.layer_norm(t, (D,), p["visual.ln_pre.weight"], p["visual.ln_pre.bias"])
    layers = sum(1 for k in p if k.endswith("attn.in_proj_weight"))
    for i in range(layers):
        pre = f"visual.transformer.resblocks.{i}."
        a = F.layer_norm(t, (D,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"])
        t = t + attention(a, p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"], p[pre + "attn.out_proj.weight"],
                          p[pre + "attn.out_proj.bias"], heads)
        m = F.layer_norm(t, (D,), p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
        m = F.linear(m, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"])
        m = m * torch.sigmoid(1.702 * m)                                   # QuickGELU
        t = t + F.linear(m, p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"])
        if i >= clip_location and cls_use:                                 # CLS adapter, mixed 0.5 / 0.5 (:183-191)
            j = i - clip_location
            c = t[:, :1]
            a1 = F.relu(F.linear(c, p[f"adapter_layer.{j}.0.weight"], p[f"adapter_layer.{j}.0.bias"]))
            a1 = F.relu(F.linear(a1, p[f"adapter_layer.{j}.2.weight"], p[f"adapter_layer.{j}.2.bias"]))
            t = torch.cat([0.5 * a1 + 0.5 * c, t[:, 1:]], 1)
    cls, pat = t[:, :1], t[:, 1:]
    return torch.cosine_similarity(cls, pat, dim=-1), cls, pat.unsqueeze(0)
