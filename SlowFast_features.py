#!/usr/bin/env python
"""Drop-in for the reference's ``SlowFast_features.py`` on MI355X: same command line (:200-217), same clip assembly
(:25-107, ``kvq_amd/datasets/slowfast_clips.py``), same output — for every clip of every video of ``--video_csv``,
``feature_{i}_slow_feature.npy`` (1,2048,1,1,1) and ``feature_{i}_fast_feature.npy`` (1,256,1,1,1) under
``<feature_save_folder>/<database>/<video_name>/``, the layout ``ViewDecompositionDataset_add_forSimpleVQA`` reads
(fusion_datasets.py:878-890).

    python SlowFast_features.py --video_root DIR --video_csv FILE.csv [--database kvq] [--feature_save_folder ...]

Differences that are deliberate: frames come through the package's frame reader (decord when installed, ``<video>.npy`` uint8
[T,H,W,3] stacks otherwise — no codec ships in this image; ``--fps`` or a ``<video>.fps`` file gives a stack's frame rate);
the network weights are a pytorchvideo hub download in the reference (:140): pass ``--weights <state_dict.pth>`` (pytorchvideo
key names under ``feature_extraction.``), else seeded random weights; ``--synthetic N`` writes N seeded random clips instead of
reading videos (smoke runs).  The SlowFast-R50 network runs as HIP kernels (``slowfast_model.py``); there is no CPU path."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kvq_amd  # noqa: E402,F401
from kvq_amd.datasets.slowfast_clips import VideoDataset_NR_SlowFast_feature, extract_video, pil_transform  # noqa: E402
from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast  # noqa: E402,F401


def main(config, video_root, videos_csv):
    device = torch.device("cuda")
    model = slowfast().to(device).eval()
    if config.small_grid_mean:
        model.head_small_grid = "mean"
    if config.weights:
        print(model.load_state_dict(torch.load(config.weights, map_location="cpu"), strict=False))
    folder = config.feature_save_folder + "/" + config.database + "/"              # (:179)
    if config.synthetic:
        g = np.random.Generator(np.random.PCG64(config.seed))
        items = [([torch.from_numpy(g.standard_normal((32, 3, config.resize, config.resize)).astype(np.float32))
                   for _ in range(config.synthetic)], config.video_name)]
    else:
        trainset = VideoDataset_NR_SlowFast_feature(config, pil_transform(config.resize), video_root, videos_csv)
        items = torch.utils.data.DataLoader(trainset, batch_size=None, shuffle=False, num_workers=config.num_workers)
    for video, video_name in items:
        print(video_name)
        os.makedirs(folder + video_name, exist_ok=True)
        for idx, (slow_feature, fast_feature) in enumerate(extract_video(model, list(video), device)):
            np.save(folder + video_name + "/" + "feature_" + str(idx) + "_slow_feature", slow_feature)
            np.save(folder + video_name + "/" + "feature_" + str(idx) + "_fast_feature", fast_feature)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--num_workers", type=int, default=6)
    parser.add_argument("--resize", type=int, default=224)
    parser.add_argument("--gpu_ids", type=list, default=None)
    parser.add_argument("--database", type=str, default="kvq")
    parser.add_argument("--video_root", type=str, default=None)
    parser.add_argument("--video_csv", type=str, default=None)
    parser.add_argument("--feature_save_folder", type=str, default="./feature/simpleVQA/")
    # not in the reference
    parser.add_argument("--weights", default=None)
    parser.add_argument("--fps", type=float, default=None, help="frame rate of .npy frame stacks without a .fps file")
    parser.add_argument("--synthetic", type=int, default=0)
    parser.add_argument("--video_name", default="synthetic_00000")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--small_grid_mean", action="store_true",
                        help="reduced-size smoke runs: a final grid under the head's AvgPool3d kernel (--resize < 224) gets the global "
                             "mean instead of the error the reference raises there")
    config = parser.parse_args()
    if not config.synthetic and (config.video_root is None or config.video_csv is None):
        parser.error("--video_root and --video_csv are required (or --synthetic N)")
    main(config, config.video_root, config.video_csv)
