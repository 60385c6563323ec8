"""ORACLE (test infrastructure, never shipped in the product path).

CPU fp32 restatements of KSVQE's content-distortion modulation modules (``models/backbones/KSVQE_model.py``):
``crossattention1`` :1553-1586, ``Attention`` :1508-1551, ``Semantic_Transformation2`` :817-835,
``Dist_Transformation3`` :934-960, as plain functions over their state_dicts.

Pinned: ``tests/golden/make_golden.py cdm`` runs the imported reference modules with the same synthetic weights,
checks these functions against them and stores the reference's outputs in ``tests/golden/cdm.npz``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def cross_attention(Q, K, p, num_heads):
    """(B, Nq, C), (B, Nk, C) -> (O (B, Nq, C), A (B, Nq, Nk) = head-mean of the softmax); logits / sqrt(C)."""
    q = F.linear(Q, p["fc_q.weight"], p["fc_q.bias"])
    k = F.linear(K, p["fc_k.weight"], p["fc_k.bias"])
    v = F.linear(K, p["fc_v.weight"], p["fc_v.bias"])
    B, Nq, C = q.shape
    hd = C // num_heads
    qh, kh, vh = (t.reshape(B, -1, num_heads, hd).transpose(1, 2) for t in (q, k, v))
    a = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(C), dim=-1)
    # second output: the reference stacks the heads HEAD-major ([h*B + b], torch.cat of the channel splits along dim 0) and
    # then views that as (B, heads, ...) before averaging (:1585) — so its "head mean" mixes batch elements.  Reproduced
    # as-is; every caller discards it (:1451, :1471).
    a_cat = a.transpose(0, 1).reshape(num_heads * B, Nq, -1)
    return (a @ vh).transpose(1, 2).reshape(B, Nq, C), a_cat.reshape(B, num_heads, Nq, -1).mean(dim=1)


def self_attention(x, p, heads):
    B, n, C = x.shape
    hd = C // heads
    q, k, v = (t.reshape(B, n, heads, hd).transpose(1, 2) for t in F.linear(x, p["to_qkv.weight"]).chunk(3, dim=-1))
    a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
    return F.linear((a @ v).transpose(1, 2).reshape(B, n, C), p["to_out.0.weight"], p["to_out.0.bias"])


def semantic_transformation2(x, inp, p):
    gama = torch.sigmoid(F.conv2d(x, p["conv_gama.weight"], p["conv_gama.bias"]))
    return gama * inp + F.conv2d(x, p["conv_beta.weight"], p["conv_beta.bias"])


def dist_transformation3(x, inp, p):
    B, C = x.shape[:2]
    flat = x.reshape(B, C, -1)
    gama = torch.sigmoid(F.linear(flat.std(dim=2), p["get_gamma.weight"], p["get_gamma.bias"]))
    beta = F.linear(flat.mean(dim=2), p["get_beta.weight"], p["get_beta.bias"])
    return gama.unsqueeze(1) * inp + beta.unsqueeze(1)


def obtain_keyframes(x):
    """KSVQE.obtain_keyframes (:1352-1376), loops as written there."""
    b, c, t, h, w = x.shape
    xt = x.permute(0, 2, 1, 3, 4)
    key = torch.stack([xt[:, 0], xt[:, t // 4 - 1], xt[:, t // 2 - 1], xt[:, t * 3 // 4 - 1]], 1)
    gid = x.new_zeros((b, t))
    for i in range(b):
        g = 0
        for j in range(t):
            if j == t // 4 - 1:
                g += 1
            elif j == t // 2 - 1:
                g += 1
            elif j == t * 3 // 4 - 1:
                g += 1
            gid[i, j] = g
    return gid, key


def qrs_select(x, score, group_id, k=49, anchor=32):
    """Eval path of RegionNet_CLIP.forward (patchnet.py:461-550): returns (patches (b, c, t, 7*anchor, 7*anchor), region index
    per key frame (b, n_key))."""
    b, c, t, h, w = x.shape
    _, n_key, L = score.shape
    gs, kk = int(round(L ** 0.5)), int(round(k ** 0.5))
    gh, gw = h // anchor, w // anchor
    s = score.reshape(b * n_key, 1, gs, gs)
    if (gs, gs) != (gh, gw):
        s = F.interpolate(s, scale_factor=(gh / gs, gw / gs), mode="nearest")
    m = F.unfold(s, kernel_size=kk, stride=1).mean(dim=1)                       # (b*n_key, regions)
    mn, mx = m.min(-1, keepdim=True).values, m.max(-1, keepdim=True).values
    m = (m - mn) / (mx - mn + 1e-5)
    idx = m.argmax(dim=-1).reshape(b, n_key)
    nx = gw - kk + 1
    out = x.new_zeros((b, c, t, kk * anchor, kk * anchor))
    for i in range(b):
        for j in range(t):
            r = int(idx[i, int(group_id[i, j])])
            ry, rx = r // nx, r % nx
            out[i, :, j] = x[i, :, j, ry * anchor: ry * anchor + kk * anchor, rx * anchor: rx * anchor + kk * anchor]
    return out, idx


def contrique(x, params, anchor=32, normalize=True):
    """CONTRIQUE_model.forward (:1643-1664) over its state_dict (``encoder.{0,1,4..7}`` = conv1, bn1, layer1..4 of a ResNet-50;
    ``projector.{0,1,3,4}``): x (b, c, t, h, w) -> (b, t, patches per frame, projection_dim)."""
    from . import resnet_oracle as R
    p = {k: (v if torch.is_tensor(v) else torch.from_numpy(v)) for k, v in params.items()}
    b, c, t, h, w = x.shape
    gh, gw = h // anchor, w // anchor
    z = (x.permute(0, 2, 1, 3, 4).reshape(b * t, c, gh, anchor, gw, anchor).permute(0, 2, 4, 1, 3, 5)
         .reshape(b * t * gh * gw, c, anchor, anchor))
    q = {}
    for k, v in p.items():                       # encoder.N.* -> the oracle's conv1 / bn1 / layerL names
        if k.startswith("encoder."):
            n, rest = k[len("encoder."):].split(".", 1)
            q[{"0": "conv1", "1": "bn1"}.get(n, f"layer{int(n) - 3}") + "." + rest] = v.float() if v.is_floating_point() else v
    z = F.relu(R._bn(F.conv2d(z, q["conv1.weight"], stride=2, padding=3), q, "bn1"))
    z = F.max_pool2d(z, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(R.LAYERS, 1):
        for bi in range(blocks):
            z = R._bottleneck(z, q, f"layer{li}.{bi}", stride if bi == 0 else 1, bi == 0)
    f = z.reshape(-1, z.shape[1])
    if normalize:
        f = F.normalize(f, dim=1)

    def bn1d(v, pre):
        return F.batch_norm(v, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"], p[pre + ".bias"], False, 0.0, 1e-5)

    f = F.relu(bn1d(F.linear(f, p["projector.0.weight"]), "projector.1"))
    f = bn1d(F.linear(f, p["projector.3.weight"]), "projector.4")
    return f.reshape(b, t, gh * gw, -1)


def _adapter(v, p, pre):
    v = F.relu(F.linear(v, p[pre + "0.weight"], p[pre + "0.bias"]))
    return F.relu(F.linear(v, p[pre + "2.weight"], p[pre + "2.bias"]))


def contrastive_supervised(feat, dis_label):
    """distortion_contrastive_supervised (:1666-1691)."""
    b, t, g, _ = feat.shape
    f = feat.reshape(b * t * g, -1)
    same = (dis_label.unsqueeze(1).repeat(1, b) == dis_label).float()
    labels = same.repeat(1, t * g).view(b * t * g, -1)
    z = F.normalize(f, p=2, dim=1)
    sim = z @ z.t() / 0.1
    n = b * t * g
    off = 1.0 - torch.eye(n)
    pos = (labels @ labels.t()) * off
    return torch.mean(torch.log(torch.sum(torch.exp(sim) * off, dim=1)) - torch.sum(sim * pos, dim=1) / torch.sum(pos, dim=1))


def ksvqe_forward(inputs, params, cfg, clip_location=8, tuning_stage=2, multi=False, layer=-1):
    """KSVQE.forward (:1389-1500), eval: inputs = {resize_video (b,3,t,112,112), fragment (b,3,t,288,288), dis_label (b,)};
    params = the model's state_dict (numpy / tensors); cfg = the trunk's SwinCfg.  Returns (features (b, 768, t/2, 7, 7), loss);
    ``multi``: the trilinear-resized concat of feats[:-1] instead (:1489-1495); ``layer`` > -1: feats[layer] (:1496-1498) — feats =
    [behind the embedding, behind every stage (a tuned stage: after its modulation)]."""
    from . import clip_oracle as CO
    from . import swin3d_oracle as O
    p = {k: ((v if torch.is_tensor(v) else torch.from_numpy(v))) for k, v in params.items()}
    p = {k: (v.float() if v.is_floating_point() else v) for k, v in p.items()}
    sub = lambda pre: {k[len(pre):]: v for k, v in p.items() if k.startswith(pre)}       # noqa: E731
    rv, frag, dis_label = inputs["resize_video"].float(), inputs["fragment"].float(), inputs["dis_label"]
    b, _, t = frag.shape[:3]
    gid, key = obtain_keyframes(rv)
    n_key = key.shape[1]
    cls_attn, _, pat = CO.clip_visual_extractor(key.reshape((b * n_key,) + tuple(key.shape[2:])), sub("CLIP_tool."),
                                                clip_location=clip_location)
    pat = pat.reshape(b, n_key, pat.shape[2], pat.shape[3])
    patch_tokens = torch.stack([torch.stack([pat[i, int(gid[i, j])] for j in range(t)]) for i in range(b)])      # (b, t, 49, 768)
    x_ori, _ = qrs_select(frag, cls_attn.reshape(b, n_key, -1), gid)
    dist = contrique(x_ori[:, :, ::2], sub("distortion_tool."))
    dist = 0.2 * _adapter(dist, p, "dist_adapter.") + 0.8 * dist
    loss = contrastive_supervised(dist, dis_label)
    shift = tuple(w // 2 for w in cfg.window)
    y = O.patch_embed(x_ori, p, cfg.patch)
    feats = [y]
    for i in range(len(cfg.depths)):
        for blk in range(cfg.depths[i]):
            y = O.swin_block(y, p, f"layers.{i}.blocks.{blk}.", cfg.num_heads[i], cfg.window, (0, 0, 0) if blk % 2 == 0 else shift)
        if i < len(cfg.depths) - 1:
            y = O.patch_merge(y, p, f"layers.{i}.downsample.")
        if i >= tuning_stage:                                      # CDM on the stage output (:1436-1482); y channels-last (n, t', h, w, c)
            k = i - tuning_stage
            n, tt, hh, ww, c = y.shape
            frames = y.reshape(n * tt, hh * ww, c)
            pt = _adapter(patch_tokens[:, ::2].reshape(n * tt, -1, patch_tokens.shape[-1]), p, f"semantic_adapter.{k}.")
            heads = cfg.num_heads[min(i, len(cfg.depths) - 2)]
            enh, _ = cross_attention(frames, pt, sub(f"semantic_cross.{k}."), heads)
            cf = lambda v: v.reshape(n * tt, hh, ww, c).permute(0, 3, 1, 2)               # noqa: E731
            x_s = semantic_transformation2(cf(enh), cf(frames), sub(f"semantic_mod.{k}.")).permute(0, 2, 3, 1).reshape(n, tt, hh, ww, c)
            dt = _adapter(dist.reshape(n * tt, -1, dist.shape[-1]), p, f"distortion_adapter.{k}.")
            de, _ = cross_attention(frames, dt, sub(f"distortion_cross.{k}."), heads)
            de = de.reshape(n, tt, hh * ww, c).permute(0, 2, 1, 3).reshape(n * hh * ww, tt, c)
            de = self_attention(de, sub(f"distortion_self.{k}."), heads)
            de = de.reshape(n, hh * ww, tt, c).permute(0, 3, 2, 1).reshape(n, c, tt, hh, ww)
            x_d = dist_transformation3(de, y.reshape(n, tt * hh * ww, c), sub(f"distortion_mod.{k}.")).reshape(n, tt, hh, ww, c)
            y = (p["a1"][k] * x_d + p["a2"][k] * x_s) / 2
        feats.append(y)
    cf5 = lambda v: v.permute(0, 4, 1, 2, 3).contiguous()                                  # noqa: E731
    if multi:
        size = tuple(y.shape[1:4])
        return torch.cat([F.interpolate(cf5(f), size=size, mode="trilinear") for f in feats[:-1]], 1)
    if layer > -1:
        return cf5(feats[layer])
    y = F.layer_norm(y, (y.shape[-1],), p["norm.weight"], p["norm.bias"])
    return y.permute(0, 4, 1, 2, 3).contiguous(), loss
