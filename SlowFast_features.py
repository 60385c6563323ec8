#!/usr/bin/env python
"""Drop-in for the reference's ``SlowFast_features.py`` feature-extraction loop (:137-197) on MI355X:
for every clip of a video, ``feature_{i}_slow_feature.npy`` (1,2048,1,1,1) and
``feature_{i}_fast_feature.npy`` (1,256,1,1,1) under ``<feature_save_folder>/<database>/<video_name>/`` —
the on-disk layout ``ViewDecompositionDataset_add_forSimpleVQA`` reads (fusion_datasets.py:878-890).

Video decoding (cv2, :52-107) is outside the hot path; clips enter as a tensor file
``--clips <file.pt>`` = fp32 (n_clips, 32, 3, H, W) already resized + normalised (mean .45 / std .225), or
``--synthetic N`` seeded clips.  The network weights are a pytorchvideo hub download in the reference
(:140); pass ``--weights <state_dict.pth>`` (pytorchvideo key names under ``feature_extraction.``)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kvq_amd  # noqa: E402,F401
from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast  # noqa: E402,F401


def main(config):
    device = torch.device("cuda")
    model = slowfast().to(device).eval()
    if config.weights:
        print(model.load_state_dict(torch.load(config.weights, map_location="cpu"), strict=False))
    if config.clips:
        video = torch.load(config.clips)
    else:
        g = np.random.Generator(np.random.PCG64(config.seed))
        video = torch.from_numpy(g.standard_normal((config.synthetic, 32, 3, config.resize, config.resize)).astype(np.float32))
    out = os.path.join(config.feature_save_folder, config.database, config.video_name)
    os.makedirs(out, exist_ok=True)
    with torch.no_grad():
        for idx in range(video.shape[0]):
            ele = video[idx:idx + 1].permute(0, 2, 1, 3, 4)                  # (1,3,32,H,W)   (:193)
            slow_feature, fast_feature = model(pack_pathway_output(ele, device))
            np.save(os.path.join(out, f"feature_{idx}_slow_feature"), slow_feature.cpu().numpy())
            np.save(os.path.join(out, f"feature_{idx}_fast_feature"), fast_feature.cpu().numpy())
    print("saved", video.shape[0], "clips to", out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--database", default="KVQ")
    ap.add_argument("--resize", type=int, default=224)
    ap.add_argument("--feature_save_folder", default="feat")
    ap.add_argument("--video_name", default="synthetic_00000")
    ap.add_argument("--clips", default=None)
    ap.add_argument("--synthetic", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weights", default=None)
    main(ap.parse_args())
