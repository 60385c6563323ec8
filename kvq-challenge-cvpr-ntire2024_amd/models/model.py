"""Host-side mirror of the reference's ``models/model.py`` (``VQA_Network`` :18-121) — THE drop-in
boundary: same constructor (``config`` dict from config/*.yml), same attribute
names (``<key>_backbone`` / ``<key>_head``, ``key_names``, ``multi``, ``layer``), same ``forward``
signature and return structure."""
from __future__ import annotations

from functools import reduce

import torch.nn as nn

from .backbones.swin_backbone import SwinTransformer3D as VideoBackbone
from .backbones.swin_backbone import swin_3d_small, swin_3d_tiny
from .head import VQAHead, simpleVQAHead


class VQA_Network(nn.Module):  # noqa: N801  (reference spelling)
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.key_names = []
        self.multi = False
        self.layer = -1
        for key, hypers in config["model"]["args"].items():
            hypers = hypers or {}
            if key == "swin_tiny":
                backbone = swin_3d_tiny(**(hypers.get("backbone") or {}))
                head = VQAHead(**(hypers.get("head") or {}))
            elif key == "swin_tiny_grpb":
                backbone = VideoBackbone()                      # GRPB trunk = FAST-VQA / the KSVQE trunk
                head = VQAHead(**(hypers.get("head") or {}))
            elif key == "swin_tiny_grpb_m":
                backbone = VideoBackbone(window_size=(4, 4, 4), frag_biases=[0, 0, 0, 0])
                head = VQAHead(**(hypers.get("head") or {}))
            elif key == "swin_small":
                backbone = swin_3d_small(**(hypers.get("backbone") or {}))
                head = VQAHead(**(hypers.get("head") or {}))
            elif key == "simpleVQA":
                from .backbones.simpleVQA_model import resnet50 as simpleVQA_Backbone
                backbone = simpleVQA_Backbone(pretrained=False)
                head = simpleVQAHead(**(hypers.get("head") or {}))
            elif key == "KSVQE":                                # the reference reads these keys one by one (model.py:59-68)
                from .backbones.KSVQE_model import KSVQE as KSVQE_Backbone
                bk = hypers["backbone"]
                backbone = KSVQE_Backbone(num_samples=bk["num_samples"], sample_type=bk["sample_type"],
                                          CLIP_location=bk["CLIP_location"], cls_use=bk["cls_use"],
                                          tuning_stage=bk["tuning_stage"], a1=bk.get("a1", 1), a2=bk.get("a2", 0),
                                          frozen_stages=bk.get("frozen_stages", -1))
                head = VQAHead(**(hypers.get("head") or {}))
            elif key == "conv_tiny":
                raise NotImplementedError("model key 'conv_tiny' (ConvNeXt-3D) is outside the built scope (SURVEY.md §8)")
            else:
                raise NotImplementedError
            self.key_names.append(key)
            setattr(self, key + "_backbone", backbone)
            setattr(self, key + "_head", head)

    def forward(self, inputs, targets=None, inference=True, return_pooled_feats=False, reduce_scores=False,
                pooled=False, clip_return=False, **kwargs):
        scores, feats, dis_contra_loss, with_loss = [], {}, None, False
        for key in self.key_names:
            feat = getattr(self, key + "_backbone")(inputs, multi=self.multi, layer=self.layer, **kwargs)
            if key == "KSVQE":                                   # (features, distortion contrastive loss) (model.py:93-96)
                feat, dis_contra_loss = feat                     # loss is None when the backbone's aux_loss is off
                with_loss = True
            scores += [getattr(self, key + "_head")(feat)]
            if return_pooled_feats:
                feats[key] = feat
        if reduce_scores:
            scores = reduce(lambda a, b: a + b, scores) if len(scores) > 1 else scores[0]
        if return_pooled_feats:
            return (scores, feats, dis_contra_loss) if with_loss else (scores, feats)
        if with_loss:
            return scores, dis_contra_loss
        return scores
