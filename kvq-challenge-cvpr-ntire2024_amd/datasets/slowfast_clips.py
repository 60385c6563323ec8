"""Clip assembly of the motion-feature extractor — the reference's ``VideoDataset_NR_SlowFast_feature``
(``SlowFast_features.py:25-107``): one 32-frame clip per second of video, the last clips padded with the last frame,
at least 8 clips per video (the last one repeated), every frame ``Resize([r, r])`` + ``ToTensor`` + ``Normalize(.45, .225)``.

Host-side index logic and the PIL transform (the reference runs them in DataLoader workers); the frames come from the
package's frame reader (``open_video``: decord when installed, uint8 ``[T, H, W, 3]`` ``.npy`` stacks otherwise).  The
network itself runs on the device (``models/backbones/slowfast_model.py``)."""
from __future__ import annotations

import csv
import os
from typing import List, Optional

import numpy as np
import torch

VIDEO_CLIP_MIN = 8          # SlowFast_features.py:71
VIDEO_LENGTH_CLIP = 32      # :72


def clip_frame_indices(video_length: int, frame_rate: int, readable: Optional[int] = None) -> List[np.ndarray]:
    """Frame index of every slot of every clip, as ``__getitem__`` assembles them (:66-105).

    ``frame_rate`` = ``int(round(fps))``; 0 -> 10 clips that all start at frame 0 (:66-69).  Clip i starts at
    ``i * frame_rate``; a clip that runs past the end keeps its available frames and repeats the LAST of them (:97-100).
    Fewer than 8 clips -> the last clip is repeated (:102-104).  ``readable`` < ``video_length``: frames the decoder
    could not deliver are replaced by the last delivered one (:87-89).  A video shorter than one second has no clip and the
    reference fails with IndexError (list index -1 of an empty list) — so does this."""
    video_clip = 10 if frame_rate == 0 else int(video_length / frame_rate)
    clips = []
    for i in range(video_clip):
        start = i * frame_rate
        if start + VIDEO_LENGTH_CLIP <= video_length:
            idx = np.arange(start, start + VIDEO_LENGTH_CLIP)
        else:
            have = video_length - start
            idx = np.concatenate([np.arange(start, video_length), np.full(VIDEO_LENGTH_CLIP - have, video_length - 1)])
        clips.append(idx.astype(np.int64))
    if video_clip < VIDEO_CLIP_MIN:
        if not clips:
            raise IndexError("list index out of range")          # transformed_video_all[video_clip - 1] of an empty list
        clips += [clips[video_clip - 1]] * (VIDEO_CLIP_MIN - video_clip)
    if readable is not None and readable < video_length:
        clips = [np.minimum(c, readable - 1) for c in clips]
    return clips


def pil_transform(resize: int):
    """``transforms.Compose([Resize([r, r]), ToTensor(), Normalize(.45, .225)])`` on a PIL image (:172-173): PIL's
    antialiased bilinear resize on uint8, /255, (v - .45) / .225 — bit-for-bit what torchvision does around PIL."""
    from PIL import Image

    def apply(frame_rgb_u8: np.ndarray) -> torch.Tensor:
        img = Image.fromarray(frame_rgb_u8).resize((resize, resize), Image.BILINEAR)
        t = torch.from_numpy(np.asarray(img, np.uint8).copy()).permute(2, 0, 1).to(torch.float32).div(255)
        return (t - 0.45) / 0.225
    return apply


def read_video_names(videos_csv: str) -> List[str]:
    """first column of every row behind the header row (:38-46)"""
    with open(videos_csv, newline="") as f:
        rows = csv.reader(f)
        next(rows)
        return [row[0] for row in rows]


def frame_rate_of(reader, path: str, default_fps: Optional[float]) -> int:
    """``int(round(cap.get(CAP_PROP_FPS)))`` (:64): from the decoder when it knows it, else ``<video>.fps`` (one number) beside
    a frame stack, else ``default_fps``."""
    if hasattr(reader, "get_avg_fps"):
        return int(round(float(reader.get_avg_fps())))
    for side in (path + ".fps", os.path.splitext(path)[0] + ".fps"):
        if os.path.exists(side):
            return int(round(float(open(side).read().strip())))
    if default_fps is None:
        raise ValueError(f"{path}: frame rate unknown (no decoder metadata, no {path}.fps): pass --fps")
    return int(round(float(default_fps)))


class VideoDataset_NR_SlowFast_feature(torch.utils.data.Dataset):  # noqa: N801  (reference spelling)
    """``(list of (32, 3, r, r) fp32 clips, video_name)`` per video, like the reference's dataset (:25-107)."""

    def __init__(self, args, transform, video_root, videos_csv):
        super().__init__()
        self.resize, self.args = args.resize, args
        self.transform = transform if transform is not None else pil_transform(args.resize)
        self.video_root, self.videos_csv = video_root, videos_csv
        self.video_infos = read_video_names(videos_csv)

    def __len__(self):
        return len(self.video_infos)

    def __getitem__(self, index):
        from .fusion_datasets import open_video
        name = self.video_infos[index]
        path = os.path.join(self.video_root, name)
        vr = open_video(path)
        length = len(vr)
        rate = frame_rate_of(vr, path, getattr(self.args, "fps", None))
        clips = clip_frame_indices(length, rate)
        done = {}
        for i in np.unique(np.concatenate(clips)):          # every USED frame is transformed once
            f = vr[int(i)]
            done[int(i)] = self.transform(f.asnumpy() if hasattr(f, "asnumpy") else np.asarray(f))
        return [torch.stack([done[int(i)] for i in c]) for c in clips], name


def extract_video(model, clips: List[torch.Tensor], device, batch: int = 8):
    """clips: list of (32, 3, r, r) fp32 -> list of (slow (1,2048,1,1,1), fast (1,256,1,1,1)) numpy pairs, one per clip
    (:189-197).  Clips are stacked ``batch`` at a time (the clips are independent; repeated padding clips are computed once)."""
    uniq, order = [], []
    for c in clips:                       # the padding to 8 clips repeats the LAST clip: compute it once
        if uniq and (uniq[-1] is c or torch.equal(uniq[-1], c)):
            order.append(len(uniq) - 1)
        else:
            order.append(len(uniq))
            uniq.append(c)
    feats = []
    model.split_k = False           # a clip's features must not depend on which clips share its forward (10 clips run as 8 + 2)
    with torch.no_grad():
        for a in range(0, len(uniq), batch):
            ele = torch.stack(uniq[a:a + batch]).permute(0, 2, 1, 3, 4).contiguous().to(device)      # (b, 3, 32, r, r)   (:193)
            slow, fast = model.forward_clips(ele)
            slow, fast = slow.cpu().numpy(), fast.cpu().numpy()
            feats += [(slow[k:k + 1], fast[k:k + 1]) for k in range(slow.shape[0])]
    return [feats[j] for j in order]
