"""Procedural (platform-stable) weights and clips for parity tests and the bench.

No dataset or checkpoint is reachable offline, so every
parity case and the benchmark use weights drawn from numpy ``PCG64`` streams and
clips drawn the same way.  Names and shapes follow the reference's state_dict
(reference ``models/backbones/swin_backbone.py:194-239,
385-405,530-531,707-711,838`` and ``models/head.py:54-55,23-26``).

Two schemes:
  * ``"init"``   mirrors the reference initialisation (``swin_backbone.py:1017-1024,
    242``): N(0, 0.02) weights, zero biases, unit LayerNorm.
  * ``"stress"`` fan-in scaled weights, non-zero biases, perturbed LayerNorm affine
    and O(1) bias tables, so that every term of the arithmetic (bias adds, gates,
    masks, LN affine) moves the output measurably.  This is the scheme parity is
    judged on.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


@dataclass(frozen=True)
class SwinCfg:
    """Constructor arguments of the reference trunk (``swin_backbone.py:760-783``)."""
    patch: Tuple[int, int, int] = (2, 4, 4)
    in_chans: int = 3
    embed_dim: int = 96
    depths: Tuple[int, ...] = (2, 2, 6, 2)
    num_heads: Tuple[int, ...] = (3, 6, 12, 24)
    window: Tuple[int, int, int] = (8, 7, 7)
    mlp_ratio: int = 4
    frag_biases: Tuple[bool, ...] = (True, True, True, False)

    @property
    def num_stages(self) -> int:
        return len(self.depths)

    def dim(self, stage: int) -> int:
        return self.embed_dim * (2 ** stage)

    @property
    def num_features(self) -> int:
        return self.dim(self.num_stages - 1)

    @property
    def table_len(self) -> int:
        w = self.window
        return (2 * w[0] - 1) * (2 * w[1] - 1) * (2 * w[2] - 1)


SWIN_T_GRPB = SwinCfg()
SWIN_T_PLAIN = SwinCfg(frag_biases=(False, False, False, False))
SWIN_S_PLAIN = SwinCfg(depths=(2, 2, 18, 2), frag_biases=(False, False, False, False))        # model key swin_small
SWIN_T_GRPB_M = SwinCfg(window=(4, 4, 4), frag_biases=(False, False, False, False))             # model key swin_tiny_grpb_m
# BASELINE.json config 5: a parameterisation the build defines (SURVEY.md §0 trap 7)
SWIN_B_GRPB = SwinCfg(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))


def _gen(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def swin_param_shapes(cfg: SwinCfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape, trunk only, in the reference's registration order."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    E = cfg.embed_dim
    s["patch_embed.proj.weight"] = (E, cfg.in_chans) + tuple(cfg.patch)
    s["patch_embed.proj.bias"] = (E,)
    s["patch_embed.norm.weight"] = (E,)
    s["patch_embed.norm.bias"] = (E,)
    for i in range(cfg.num_stages):
        C, nH = cfg.dim(i), cfg.num_heads[i]
        for b in range(cfg.depths[i]):
            p = f"layers.{i}.blocks.{b}."
            s[p + "norm1.weight"] = (C,)
            s[p + "norm1.bias"] = (C,)
            s[p + "attn.relative_position_bias_table"] = (cfg.table_len, nH)
            if cfg.frag_biases[i]:
                s[p + "attn.fragment_position_bias_table"] = (cfg.table_len, nH)
            s[p + "attn.qkv.weight"] = (3 * C, C)
            s[p + "attn.qkv.bias"] = (3 * C,)
            s[p + "attn.proj.weight"] = (C, C)
            s[p + "attn.proj.bias"] = (C,)
            s[p + "norm2.weight"] = (C,)
            s[p + "norm2.bias"] = (C,)
            s[p + "mlp.fc1.weight"] = (cfg.mlp_ratio * C, C)
            s[p + "mlp.fc1.bias"] = (cfg.mlp_ratio * C,)
            s[p + "mlp.fc2.weight"] = (C, cfg.mlp_ratio * C)
            s[p + "mlp.fc2.bias"] = (C,)
        if i < cfg.num_stages - 1:
            p = f"layers.{i}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,)
            s[p + "norm.bias"] = (4 * C,)
    s["norm.weight"] = (cfg.num_features,)
    s["norm.bias"] = (cfg.num_features,)
    return s


def vqa_head_param_shapes(in_channels=768, hidden=64, num_class=1) -> "OrderedDict[str, Tuple[int, ...]]":
    return OrderedDict([
        ("fc_hid.weight", (hidden, in_channels, 1, 1, 1)),
        ("fc_hid.bias", (hidden,)),
        ("fc_last.weight", (num_class, hidden, 1, 1, 1)),
        ("fc_last.bias", (num_class,)),
    ])


def simple_head_param_shapes(in_channels=9472, hidden=128) -> "OrderedDict[str, Tuple[int, ...]]":
    return OrderedDict([
        ("quality.0.weight", (hidden, in_channels)),
        ("quality.0.bias", (hidden,)),
        ("quality.1.weight", (1, hidden)),
        ("quality.1.bias", (1,)),
    ])


def resnet50_param_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict of the reference's simpleVQA ResNet-50 (simpleVQA_model.py:128-218), buffers included."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def bn(prefix, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}.{leaf}"] = (c,)
        s[f"{prefix}.num_batches_tracked"] = ()

    s["conv1.weight"] = (64, 3, 7, 7)
    bn("bn1", 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
        for bi in range(blocks):
            p = f"layer{li}.{bi}"
            s[p + ".conv1.weight"] = (planes, inplanes, 1, 1); bn(p + ".bn1", planes)
            s[p + ".conv2.weight"] = (planes, planes, 3, 3); bn(p + ".bn2", planes)
            s[p + ".conv3.weight"] = (planes * 4, planes, 1, 1); bn(p + ".bn3", planes * 4)
            if bi == 0:
                s[p + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1); bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    s["quality.0.weight"] = (128, 9472); s["quality.0.bias"] = (128,)
    s["quality.1.weight"] = (1, 128); s["quality.1.bias"] = (1,)
    return s


def synth_resnet50_weights(seed: int = 0, scheme: str = "stress"):
    return synth_params(resnet50_param_shapes(), seed, scheme, prefix="resnet.")


def _draw(name: str, shape, seed: int, scheme: str) -> np.ndarray:
    g = _gen(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "running_mean":
        return (0.1 * g.standard_normal(shape)).astype(np.float32) if scheme == "stress" else np.zeros(shape, np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32) if scheme == "stress" else np.ones(shape, np.float32)
    is_norm = (".norm" in name or name.startswith("norm.") or ".bn" in name
               or "downsample.1." in name)
    if scheme == "init":
        if "position_bias_table" in name:
            return (g.standard_normal(shape) * 0.02).astype(np.float32)
        if is_norm:
            return (np.ones(shape) if leaf == "weight" else np.zeros(shape)).astype(np.float32)
        if leaf == "bias":
            return np.zeros(shape, np.float32)
        return np.clip(g.standard_normal(shape) * 0.02, -2.0, 2.0).astype(np.float32)
    if scheme != "stress":
        raise ValueError(f"unknown scheme {scheme!r}")
    if "position_bias_table" in name:
        return (g.standard_normal(shape) * 0.5).astype(np.float32)
    if is_norm:
        if leaf == "weight":
            return (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "bias":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return (g.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


def synth_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, scheme: str = "stress",
                 prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Draw every tensor in ``shapes`` from its own PCG64 stream (keyed by name)."""
    return OrderedDict((k, _draw(prefix + k, shp, seed, scheme)) for k, shp in shapes.items())


def synth_swin_weights(cfg: SwinCfg = SWIN_T_GRPB, seed: int = 0, scheme: str = "stress"):
    return synth_params(swin_param_shapes(cfg), seed, scheme)


def synth_swin2d_checkpoint(cfg: SwinCfg = SWIN_T_GRPB, seed: int = 0, window2d: int = 7):
    """A synthetic 2D-Swin checkpoint body (the ``{"model": ...}`` dict ``inflate_weights`` reads, reference
    swin_backbone.py:858-931): the 3D trunk's own keys with the patch-embed conv collapsed to (E, 3, ph, pw), every
    relative-position table in its 2D size ((2·window2d−1)², nH), no fragment tables, plus the
    ``relative_position_index`` / ``attn_mask`` buffers a 2D checkpoint carries (the loader must drop them)."""
    w3 = synth_swin_weights(cfg, seed, "stress")
    out = OrderedDict()
    for k, v in w3.items():
        if "fragment_position_bias_table" in k:
            continue
        if k == "patch_embed.proj.weight":
            out[k] = np.ascontiguousarray(v[:, :, 0])
        elif "relative_position_bias_table" in k:
            out[k] = _gen(seed, "2d/" + k).standard_normal(((2 * window2d - 1) ** 2, v.shape[1])).astype(np.float32)
            n = window2d * window2d
            out[k.replace("relative_position_bias_table", "relative_position_index")] = \
                _gen(seed, "2di/" + k).integers(0, (2 * window2d - 1) ** 2, (n, n)).astype(np.int64)
        else:
            out[k] = v
    out["layers.0.blocks.1.attn_mask"] = np.zeros((64, window2d * window2d, window2d * window2d), np.float32)
    return out


def synth_swin3d_checkpoint(cfg: SwinCfg = SWIN_T_GRPB, seed: int = 0):
    """A synthetic Video-Swin (mmaction-style) checkpoint body (the ``{"state_dict": ...}`` dict ``load_swin`` reads,
    reference swin_backbone.py:933-1006): trunk keys under ``backbone.`` WITHOUT fragment tables (the loader forks
    them from the relative tables), a classifier head that must be ignored and one key of the wrong shape that
    must be dropped."""
    w3 = synth_swin_weights(cfg, seed, "stress")
    out = OrderedDict()
    for k, v in w3.items():
        if "fragment_position_bias_table" not in k:
            out["backbone." + k] = v
    out["backbone.norm.weight"] = np.ones((cfg.num_features + 1,), np.float32)      # shape mismatch: dropped
    out["cls_head.fc_cls.weight"] = _gen(seed, "cls").standard_normal((400, cfg.num_features)).astype(np.float32)
    out["cls_head.fc_cls.bias"] = np.zeros((400,), np.float32)
    return out


def clip_visual_param_shapes(width=768, layers=12, patch=16, grid=14, out_dim=512, clip_location=8, cls_use=True):
    """state_dict of the reference's ``CLIP_extractor_addadapter_cls`` (CLIP_backbone.py:115-201) around the vendored
    ViT (clip/model.py:252-267): ``visual.*`` + ``adapter_layer.{j}.{0,2}.*`` for the layers from ``clip_location`` on."""
    sh = OrderedDict()
    sh["visual.class_embedding"] = (width,)
    sh["visual.positional_embedding"] = (grid * grid + 1, width)
    sh["visual.proj"] = (width, out_dim)
    sh["visual.conv1.weight"] = (width, 3, patch, patch)
    sh["visual.ln_pre.weight"] = (width,)
    sh["visual.ln_pre.bias"] = (width,)
    for i in range(layers):
        pre = f"visual.transformer.resblocks.{i}."
        sh[pre + "attn.in_proj_weight"] = (3 * width, width)
        sh[pre + "attn.in_proj_bias"] = (3 * width,)
        sh[pre + "attn.out_proj.weight"] = (width, width)
        sh[pre + "attn.out_proj.bias"] = (width,)
        sh[pre + "ln_1.weight"] = (width,)
        sh[pre + "ln_1.bias"] = (width,)
        sh[pre + "mlp.c_fc.weight"] = (4 * width, width)
        sh[pre + "mlp.c_fc.bias"] = (4 * width,)
        sh[pre + "mlp.c_proj.weight"] = (width, 4 * width)
        sh[pre + "mlp.c_proj.bias"] = (width,)
        sh[pre + "ln_2.weight"] = (width,)
        sh[pre + "ln_2.bias"] = (width,)
    sh["visual.ln_post.weight"] = (width,)
    sh["visual.ln_post.bias"] = (width,)
    if cls_use:
        for j in range(layers - clip_location):
            sh[f"adapter_layer.{j}.0.weight"] = (width // 4, width)
            sh[f"adapter_layer.{j}.0.bias"] = (width // 4,)
            sh[f"adapter_layer.{j}.2.weight"] = (width, width // 4)
            sh[f"adapter_layer.{j}.2.bias"] = (width,)
    return sh


def synth_clip_visual_weights(seed: int = 0, **kw):
    """Fan-in scaled Linear / conv weights, LayerNorm weights around 1 ('stress' rules: the ``ln_`` names count as norms),
    embeddings at the ViT's width^-0.5 scale."""
    out = OrderedDict()
    for name, shape in clip_visual_param_shapes(**kw).items():
        g = _gen(seed, "clip/" + name)
        leaf = name.rsplit(".", 1)[-1]
        if ".ln_" in name:
            out[name] = ((1.0 + 0.1 * g.standard_normal(shape)) if leaf == "weight" else 0.1 * g.standard_normal(shape)).astype(np.float32)
        elif "embedding" in name or name == "visual.proj":
            out[name] = (g.standard_normal(shape) * shape[-1] ** -0.5).astype(np.float32)
        elif leaf in ("bias", "in_proj_bias"):
            out[name] = (0.1 * g.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (g.standard_normal(shape) / np.sqrt(int(np.prod(shape[1:])))).astype(np.float32)
    return out


def synth_cdm_weights(seed: int = 0, dim: int = 768):
    """Synthetic state_dicts of the four CDM modules of one tuned KSVQE stage (KSVQE_model.py:1160-1186), keyed by module."""
    shapes = {
        "cross": {"fc_q.weight": (dim, dim), "fc_q.bias": (dim,), "fc_k.weight": (dim, dim), "fc_k.bias": (dim,),
                  "fc_v.weight": (dim, dim), "fc_v.bias": (dim,)},
        "self": {"to_qkv.weight": (3 * dim, dim), "to_out.0.weight": (dim, dim), "to_out.0.bias": (dim,)},
        "sem": {"conv_gama.weight": (1, dim, 1, 1), "conv_gama.bias": (1,), "conv_beta.weight": (1, dim, 1, 1), "conv_beta.bias": (1,)},
        "dist": {"get_gamma.weight": (dim, dim), "get_gamma.bias": (dim,), "get_beta.weight": (dim, dim), "get_beta.bias": (dim,)},
    }
    out = {}
    for mod, sh in shapes.items():
        out[mod] = OrderedDict()
        for name, shape in sh.items():
            g = _gen(seed, f"cdm/{mod}/{name}")
            if name.endswith("bias"):
                out[mod][name] = (0.1 * g.standard_normal(shape)).astype(np.float32)
            else:
                out[mod][name] = (g.standard_normal(shape) / np.sqrt(int(np.prod(shape[1:])))).astype(np.float32)
    return out


def synth_contrique_weights(seed: int = 0, n_features=2048, projection_dim=128):
    """state_dict of the reference's ``CONTRIQUE_model`` (KSVQE_model.py:1622-1641): ``encoder.{0,1,4..7}`` = conv1, bn1,
    layer1..4 of a ResNet-50 (the 'stress' weights of ``synth_resnet50_weights``) + the two-layer projector with its
    BatchNorm1d statistics."""
    out = OrderedDict()
    for k, v in synth_resnet50_weights(seed).items():
        if k.startswith("quality"):
            continue
        head, rest = k.split(".", 1)
        idx = {"conv1": "0", "bn1": "1"}.get(head) or str(int(head[5:]) + 3)
        out[f"encoder.{idx}.{rest}"] = v
    for name, shape in (("projector.0.weight", (n_features, n_features)), ("projector.3.weight", (projection_dim, n_features))):
        out[name] = (_gen(seed, "contrique/" + name).standard_normal(shape) * 4.0 / np.sqrt(shape[1])).astype(np.float32)
    for pre, c in (("projector.1", n_features), ("projector.4", projection_dim)):
        g = _gen(seed, "contrique/" + pre)
        out[pre + ".weight"] = (1.0 + 0.1 * g.standard_normal(c)).astype(np.float32)
        out[pre + ".bias"] = (0.1 * g.standard_normal(c)).astype(np.float32)
        out[pre + ".running_mean"] = (0.02 * g.standard_normal(c)).astype(np.float32)
        out[pre + ".running_var"] = g.uniform(0.005, 0.02, c).astype(np.float32)
        out[pre + ".num_batches_tracked"] = np.zeros((), np.int64)
    return out


def synth_ksvqe_weights(seed: int = 0, clip_location: int = 8):
    """A complete synthetic state_dict of the reference's KSVQE (KSVQE_model.py:1024-1198): trunk ('stress'), CLIP_tool,
    distortion_tool, the Linear-ReLU adapters, the CDM modules of the two tuned stages, and a1 / a2 away from their (1, 0)
    defaults so that both modulation branches reach the output."""
    out = OrderedDict(synth_swin_weights(SWIN_T_GRPB, seed, "stress"))
    for k, v in synth_clip_visual_weights(seed + 1, clip_location=clip_location).items():
        out["CLIP_tool." + k] = v
    for k, v in synth_contrique_weights(seed + 2).items():
        out["distortion_tool." + k] = v

    def adapter(pre, cin, hid, cout):
        for idx, (o, i) in (("0", (hid, cin)), ("2", (cout, hid))):
            g = _gen(seed, "ksvqe/" + pre + idx)
            out[f"{pre}{idx}.weight"] = (g.standard_normal((o, i)) * (2.0 / i) ** 0.5).astype(np.float32)
            out[f"{pre}{idx}.bias"] = (0.1 * g.standard_normal(o)).astype(np.float32)

    adapter("dist_adapter.", 128, 32, 128)
    for k in range(2):
        adapter(f"semantic_adapter.{k}.", 768, 192, 768)
        adapter(f"distortion_adapter.{k}.", 128, 32, 768)
        cdm = synth_cdm_weights(seed + 10 + k)
        for mod, pre in (("cross", "semantic_cross"), ("sem", "semantic_mod"), ("dist", "distortion_mod"), ("self", "distortion_self")):
            for name, v in cdm[mod].items():
                out[f"{pre}.{k}.{name}"] = v
        for name, v in synth_cdm_weights(seed + 20 + k)["cross"].items():
            out[f"distortion_cross.{k}.{name}"] = v
    out["a1"] = np.asarray([[0.9], [1.1]], np.float32)
    out["a2"] = np.asarray([[0.4], [0.3]], np.float32)
    return out


def synth_ksvqe_inputs(seed: int = 0, b: int = 2, t: int = 32):
    """resize_video (b,3,t,112,112) ~ N(0,1) (CLIP-normalised scale), fragment (b,3,t,288,288) ~ N(0,1) (KVQ-normalised), labels."""
    g = _gen(seed, "ksvqe/inputs")
    return {"resize_video": g.standard_normal((b, 3, t, 112, 112)).astype(np.float32),
            "fragment": g.standard_normal((b, 3, t, 288, 288)).astype(np.float32),
            "dis_label": np.arange(b, dtype=np.int64) % 2}


def synth_vqa_head_weights(in_channels=768, hidden=64, seed: int = 0, scheme: str = "stress", num_class=1):
    return synth_params(vqa_head_param_shapes(in_channels, hidden, num_class), seed, scheme, prefix="head.")


def synth_simple_head_weights(in_channels=9472, hidden=128, seed: int = 0, scheme: str = "stress"):
    return synth_params(simple_head_param_shapes(in_channels, hidden), seed, scheme, prefix="shead.")


# ---------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------
KVQ_MEAN = (123.675, 116.28, 103.53)      # reference fusion_datasets.py:953
KVQ_STD = (58.395, 57.12, 57.375)         # reference fusion_datasets.py:954


def synth_clip(seed: int, T=32, H=224, W=224, batch=1) -> np.ndarray:
    """(batch,3,T,H,W) fp32 clip: i.i.d. uint8 pixels, normalised like the dataset
    (reference ``fusion_datasets.py:1017-1020``)."""
    g = _gen(seed, "clip")
    px = g.integers(0, 256, size=(batch, 3, T, H, W), dtype=np.uint8).astype(np.float32)
    mean = np.asarray(KVQ_MEAN, np.float32).reshape(1, 3, 1, 1, 1)
    std = np.asarray(KVQ_STD, np.float32).reshape(1, 3, 1, 1, 1)
    return (px - mean) / std


def synth_video_u8(seed: int, T=256, H=540, W=960) -> np.ndarray:
    """(3,T,H,W) uint8 post-decode frame stack (SURVEY.md §8d synthetic unit of work)."""
    g = _gen(seed, "video")
    return g.integers(0, 256, size=(3, T, H, W), dtype=np.uint8)


def synth_fragment_offsets(seed: int, T: int, H: int, W: int, fragments_h=7, fragments_w=7,
                           fsize_h=32, fsize_w=32, aligned=8):
    """Offsets rnd_h, rnd_w of shape (Fh, Fw, T//aligned) with the reference's ranges
    (``fusion_datasets.py:86-98``: U{0..hlength-fsize-1}, zeros when the cell is not larger
    than the patch)."""
    g = _gen(seed, "offsets")
    nt = T // aligned
    hl, wl = H // fragments_h, W // fragments_w
    shape = (fragments_h, fragments_w, nt)
    rh = g.integers(0, hl - fsize_h, size=shape, dtype=np.int64) if hl > fsize_h else np.zeros(shape, np.int64)
    rw = g.integers(0, wl - fsize_w, size=shape, dtype=np.int64) if wl > fsize_w else np.zeros(shape, np.int64)
    return rh.astype(np.int32), rw.astype(np.int32)
