"""Host-side mirror of the reference's ``models/head.py`` (``VQAHead`` :33-68, ``simpleVQAHead``
:10-31).  Parameter names/shapes are the reference's; forward calls libkvq_hip.so.
Inference only: dropout is the identity (``model.eval()`` in ``trainer.py:257,303``)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import kernels
from .backbones.swin_backbone import _Affine


class VQAHead(nn.Module):
    def __init__(self, in_channels=768, hidden_channels=64, num_class=1, dropout_ratio=0.5, pre_pool=False,
                 **kwargs):
        super().__init__()
        self.in_channels, self.hidden_channels, self.num_class = in_channels, hidden_channels, num_class
        self.dropout_ratio, self.pre_pool = dropout_ratio, pre_pool
        self.fc_hid = _Affine((hidden_channels, in_channels, 1, 1, 1), (hidden_channels,))
        self.fc_last = _Affine((num_class, hidden_channels, 1, 1, 1), (num_class,))
        with torch.no_grad():
            nn.init.trunc_normal_(self.fc_hid.weight, std=0.02)
            nn.init.trunc_normal_(self.fc_last.weight, std=0.02)

    def _prepared(self, device):
        """fp32 device copies in the layouts the kernels stream (W1 as it is and transposed), rebuilt only when a
        parameter changed (version / data_ptr) — no per-forward conversion kernels."""
        ps = (self.fc_hid.weight, self.fc_hid.bias, self.fc_last.weight, self.fc_last.bias)
        sig = (str(device),) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_cache", None) is None or self._cache[0] != sig:
            f32 = lambda p: p.detach().to(device=device, dtype=torch.float32)  # noqa: E731
            w1 = f32(ps[0]).reshape(self.hidden_channels, -1).contiguous()
            self._cache = (sig, (w1.t().contiguous(), f32(ps[1]).contiguous(), f32(ps[2]).reshape(-1).contiguous(),
                                 f32(ps[3]).contiguous(), w1))
        return self._cache[1]

    def forward(self, x, rois=None):
        """x (B, C, D, H, W) fp32 (any strides) -> (B, num_class)."""
        w1t, b1, w2, b2, w1 = self._prepared(x.device)
        if self.num_class != 1 or self.pre_pool:      # head.py:61-62, :66-67 — no reference config takes these branches
            return kernels.vqa_head_classes(x.to(torch.float32), w1, b1, w2.reshape(self.num_class, -1), b2,
                                            pre_pool=self.pre_pool)
        return kernels.vqa_head(x.to(torch.float32), w1, b1, w2, b2, w1t=w1t)


class simpleVQAHead(nn.Module):  # noqa: N801  (reference spelling)
    def __init__(self, in_channels=4096 + 2048 + 1024 + 2048 + 256, hidden_channels=128):
        super().__init__()
        self.quality = nn.Sequential(_Affine((hidden_channels, in_channels), (hidden_channels,)),
                                     _Affine((1, hidden_channels), (1,)))
        with torch.no_grad():
            nn.init.trunc_normal_(self.quality[0].weight, std=0.02)
            nn.init.trunc_normal_(self.quality[1].weight, std=0.02)

    def forward(self, x):
        """x (B, T, Cin) fp32 -> (B, 1): two Linears, no activation, mean over frames."""
        f32 = lambda p: p.detach().to(device=x.device, dtype=torch.float32).contiguous()  # noqa: E731
        q0, q1 = self.quality[0], self.quality[1]
        return kernels.simple_vqa_head(x.to(torch.float32), f32(q0.weight), f32(q0.bias), f32(q1.weight).reshape(-1),
                                       f32(q1.bias))
