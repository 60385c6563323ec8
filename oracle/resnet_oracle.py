"""ORACLE (test infrastructure): CPU fp32 restatement of the SimpleVQA spatial branch — the reference's
2D ResNet-50 forward with (avg, unbiased-std) pooling after layer2/3/4 and the SlowFast feature
concatenation (``/root/reference/models/backbones/simpleVQA_model.py:220-264``, Bottleneck ``:104-124``,
``global_std_pool2d`` ``:8-11``), written functionally over a state_dict.  Pinned to the imported reference
by ``tests/golden/make_golden.py`` (section ``resnet``).  Only tests / smoke / bench's cpu_baseline may
import this module."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def _bn(x, p, pre):
    return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"], p[pre + ".bias"],
                        False, 0.0, 1e-5)


def _bottleneck(x, p, pre, stride, has_down):
    out = F.relu(_bn(F.conv2d(x, p[pre + ".conv1.weight"]), p, pre + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, p[pre + ".conv2.weight"], stride=stride, padding=1), p, pre + ".bn2"))
    out = _bn(F.conv2d(out, p[pre + ".conv3.weight"]), p, pre + ".bn3")
    idt = _bn(F.conv2d(x, p[pre + ".downsample.0.weight"], stride=stride), p, pre + ".downsample.1") if has_down else x
    return F.relu(out + idt)


def simplevqa_features(frames: torch.Tensor, feat3d: torch.Tensor, params) -> torch.Tensor:
    """frames (B,3,T,H,W) fp32, feat3d (B,T,2304) -> (B,T,9472)."""
    p = {k: _t(v).float() if _t(v).is_floating_point() else _t(v) for k, v in params.items()}
    B, C, T, H, W = frames.shape
    x = frames.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    x = F.relu(_bn(F.conv2d(x, p["conv1.weight"], stride=2, padding=3), p, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    pooled = []
    for li, (planes, blocks, stride) in enumerate(LAYERS, 1):
        for bi in range(blocks):
            x = _bottleneck(x, p, f"layer{li}.{bi}", stride if bi == 0 else 1, bi == 0)
        if li >= 2:
            flat = x.reshape(x.shape[0], x.shape[1], -1)
            pooled += [flat.mean(-1), flat.std(-1)]            # torch.std: unbiased
    out = torch.cat(pooled + [feat3d.reshape(B * T, -1).float()], dim=1)
    return out.reshape(B, T, -1)
