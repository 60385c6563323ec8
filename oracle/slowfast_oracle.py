"""ORACLE (test infrastructure) — **parity unpinned**.

CPU fp32 restatement of the SlowFast-R50 (8x8) feature extractor as the reference uses it
(``/root/reference/SlowFast_features.py:112-165``: ``pack_pathway_output``, blocks 0-4 of
``pytorchvideo.models.hub.slowfast_r50``, AvgPool3d (8,7,7)/(32,7,7), AdaptiveAvgPool3d(1)).
The network arithmetic lives in the third-party package ``pytorchvideo`` (imported at
``SlowFast_features.py:21``), which is NOT vendored, NOT in ``requirements.txt`` (no version pinned), not
installed here and whose weights need the network; the reference has no tests or golden vectors for this
path.  This file therefore restates the published SlowFast-R50 8x8 architecture (Feichtenhofer et al. 2019;
pytorchvideo ``create_slowfast(model_depth=50)`` defaults) as summarised in SURVEY.md App. B; it cannot be
checked against the reference here.  What IS pinned: the wrapper semantics (pathway packing indices,
pooling, output shapes (1,2048,1,1,1)/(1,256,1,1,1)) and HIP == this oracle.
Only tests / smoke / bench's cpu_baseline may import this module."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

DEPTHS = (3, 4, 6, 3)
SLOW = dict(stem=64, inner=(64, 128, 256, 512), out=(256, 512, 1024, 2048), ka=(1, 1, 3, 3))
FAST = dict(stem=8, inner=(8, 16, 32, 64), out=(32, 64, 128, 256), ka=(3, 3, 3, 3))
SPATIAL_STRIDE = (1, 2, 2, 2)


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def pack_pathway_output(frames: torch.Tensor):
    """(B,3,T,H,W) -> [slow (T/4 frames at linspace(0,T-1,T/4).long()), fast (all frames)]  (:112-135)."""
    T = frames.shape[2]
    idx = torch.linspace(0, T - 1, T // 4).long()
    return [torch.index_select(frames, 2, idx), frames]


def _bn(x, p, pre):
    return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"], p[pre + ".bias"],
                        False, 0.0, 1e-5)


def _res_block(x, p, pre, ka, stride, has_branch1):
    y = F.relu(_bn(F.conv3d(x, p[pre + ".branch2.conv_a.weight"], padding=(ka // 2, 0, 0)), p, pre + ".branch2.norm_a"))
    y = F.relu(_bn(F.conv3d(y, p[pre + ".branch2.conv_b.weight"], stride=(1, stride, stride), padding=(0, 1, 1)),
                   p, pre + ".branch2.norm_b"))
    y = _bn(F.conv3d(y, p[pre + ".branch2.conv_c.weight"]), p, pre + ".branch2.norm_c")
    sc = x
    if has_branch1:
        sc = _bn(F.conv3d(x, p[pre + ".branch1_conv.weight"], stride=(1, stride, stride)), p, pre + ".branch1_norm")
    return F.relu(sc + y)


def slowfast_features(frames: torch.Tensor, params):
    """frames (B,3,T,H,W) fp32 (T % 4 == 0) -> (slow (B,2048,1,1,1), fast (B,256,1,1,1)) as the reference's
    ``slowfast.forward`` returns them; the head pools are global over what remains."""
    p = {k: (_t(v).float() if _t(v).is_floating_point() else _t(v)) for k, v in params.items()}
    slow, fast = pack_pathway_output(frames.float())
    fe = "feature_extraction."
    # stem
    slow = F.relu(_bn(F.conv3d(slow, p[fe + "0.multipathway_blocks.0.conv.weight"], stride=(1, 2, 2), padding=(0, 3, 3)),
                      p, fe + "0.multipathway_blocks.0.norm"))
    fast = F.relu(_bn(F.conv3d(fast, p[fe + "0.multipathway_blocks.1.conv.weight"], stride=(1, 2, 2), padding=(2, 3, 3)),
                      p, fe + "0.multipathway_blocks.1.norm"))
    slow = F.max_pool3d(slow, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    fast = F.max_pool3d(fast, (1, 3, 3), (1, 2, 2), (0, 1, 1))

    def fuse(s, f, stage):
        pre = fe + f"{stage}.multipathway_fusion"
        g = F.relu(_bn(F.conv3d(f, p[pre + ".conv_fast_to_slow.weight"], stride=(4, 1, 1), padding=(3, 0, 0)), p, pre + ".norm"))
        return torch.cat([s, g], 1), f

    slow, fast = fuse(slow, fast, 0)
    for si in range(4):
        for pi, (x, cfg) in enumerate(((slow, SLOW), (fast, FAST))):
            for bi in range(DEPTHS[si]):
                pre = fe + f"{si + 1}.multipathway_blocks.{pi}.res_blocks.{bi}"
                x = _res_block(x, p, pre, cfg["ka"][si], SPATIAL_STRIDE[si] if bi == 0 else 1, bi == 0)
            if pi == 0:
                slow = x
            else:
                fast = x
        if si < 3:
            slow, fast = fuse(slow, fast, si + 1)
    # reference: AvgPool3d((8,7,7)) / ((32,7,7)) then AdaptiveAvgPool3d(1) == global mean at 224^2 / 32 frames;
    # for other input sizes the head pools are restated as global means over what remains.
    # A grid at least as large as the kernels gets the real pools (overlapping windows, then their mean); a smaller grid — where
    # the reference raises — gets the global mean: reduced-size parity cases only.
    outs = []
    for y, k in ((slow, (8, 7, 7)), (fast, (32, 7, 7))):
        if all(g >= kk for g, kk in zip(y.shape[2:], k)) and tuple(y.shape[2:]) != k:
            y = F.avg_pool3d(y, k, stride=1, padding=0)
        outs.append(y.mean((2, 3, 4), keepdim=True))
    return outs[0], outs[1]


def param_shapes():
    """state_dict (pytorchvideo naming under ``feature_extraction.``) -> shape."""
    from collections import OrderedDict
    s = OrderedDict()

    def bn(pre, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{pre}.{leaf}"] = (c,)
        s[f"{pre}.num_batches_tracked"] = ()

    fe = "feature_extraction."
    s[fe + "0.multipathway_blocks.0.conv.weight"] = (64, 3, 1, 7, 7); bn(fe + "0.multipathway_blocks.0.norm", 64)
    s[fe + "0.multipathway_blocks.1.conv.weight"] = (8, 3, 5, 7, 7); bn(fe + "0.multipathway_blocks.1.norm", 8)
    fast_c = [8, 32, 64, 128]
    for st in range(4):
        pre = fe + f"{st}.multipathway_fusion"
        s[pre + ".conv_fast_to_slow.weight"] = (2 * fast_c[st], fast_c[st], 7, 1, 1); bn(pre + ".norm", 2 * fast_c[st])
    slow_in = [64 + 16, 256 + 64, 512 + 128, 1024 + 256]
    fast_in = [8, 32, 64, 128]
    for si in range(4):
        for pi, (cfg, cin0) in enumerate(((SLOW, slow_in[si]), (FAST, fast_in[si]))):
            cin = cin0
            for bi in range(DEPTHS[si]):
                pre = fe + f"{si + 1}.multipathway_blocks.{pi}.res_blocks.{bi}"
                inner, cout, ka = cfg["inner"][si], cfg["out"][si], cfg["ka"][si]
                if bi == 0:
                    s[pre + ".branch1_conv.weight"] = (cout, cin, 1, 1, 1); bn(pre + ".branch1_norm", cout)
                s[pre + ".branch2.conv_a.weight"] = (inner, cin, ka, 1, 1); bn(pre + ".branch2.norm_a", inner)
                s[pre + ".branch2.conv_b.weight"] = (inner, inner, 1, 3, 3); bn(pre + ".branch2.norm_b", inner)
                s[pre + ".branch2.conv_c.weight"] = (cout, inner, 1, 1, 1); bn(pre + ".branch2.norm_c", cout)
                cin = cout
    return s
