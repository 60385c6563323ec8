"""Host-side mirror of the sampler half of the reference's ``datasets/fusion_datasets.py``.

  get_spatial_fragments   (:22-121)   grid-mini-patch sampler -> ``kvq_fragment_gather`` (HIP)
  UnifiedFrameSampler     (:612-660)  temporal index sampler (host integers, numpy)
  ViewDecompositionDataset_KVQ (:930-1051), ViewDecompositionDataset_add_forSimpleVQA (:786-927)
                          the reference's dataset classes under their own names (annotation parsing, samplers, dict
                          keys), fed by a frame reader (decord if present, uint8 .npy stacks otherwise)
  SyntheticKVQDataset     build-only: seeded post-decode frame stacks (no dataset/decoder is reachable
                          offline — SURVEY.md §8d); same dict keys.

The reference draws its random offsets inside the functions from global RNG state
(``torch.randint`` :87-98, ``np.random.randint`` :632-635).  The same calls are made here, in the
same order, so a seeded run draws the same offsets (SURVEY.md §0 trap 6); every function also accepts
the offsets explicitly, which is what the parity tests use.
Video *decode* (decord / cv2, :379-431) is outside the hot path (the reader selection incl. the cv2 fallback's
padding rule is reproduced, the codecs themselves are not part of this image): frames enter
as uint8/fp32 (C,T,H,W) tensors.
"""
from __future__ import annotations

import random as _pyrandom
from typing import Optional

import os
import threading

import numpy as np
import torch

from .. import kernels

KVQ_MEAN = (123.675, 116.28, 103.53)
KVQ_STD = (58.395, 57.12, 57.375)


def _grid(res: int, fragments: int, fsize: int):
    return [min(res // fragments * i, res - fsize) for i in range(fragments)]


def get_spatial_fragments(video, fragments_h=7, fragments_w=7, fsize_h=32, fsize_w=32, aligned=32, nfrags=1,
                          random=False, random_upsample=False, fallback_type="upsample", rnd_h=None, rnd_w=None,
                          mean=None, std=None, lazy=False, **kwargs):
    """video (C,T,H,W) uint8|fp32 on a HIP device -> fp32 (C,T,Fh*fs,Fw*fs); ``lazy=True`` -> the draws and the frames as a
    one-entry ``kernels.FragmentSource`` (the trunk's embedding launch samples while it reads; ``.materialise()[0]`` is the tensor).

    ``rnd_h``/``rnd_w`` (Fh,Fw,T//aligned): offsets inside each grid cell; drawn with the reference's
    ``torch.randint`` calls when omitted.  ``mean``/``std`` fuse the dataset's normalisation (:1017-1020)."""
    if random or random_upsample:
        raise NotImplementedError("'random' / 'random_upsample' sampling is deprecated in the reference (:75)")
    if video.shape[1] == 1:
        aligned = 1
    T, H, W = video.shape[-3:]
    ratio = min(H / (fragments_h * fsize_h), W / (fragments_w * fsize_w))
    if ratio < 1:
        if fallback_type != "upsample":
            raise NotImplementedError(f"fallback_type {fallback_type!r}: the reference only knows 'upsample' (:43)")
        if H < fsize_h or W < fsize_w:
            raise ValueError(f"a {H}x{W} source is smaller than one {fsize_h}x{fsize_w} mini-patch (the reference indexes it "
                             "with negative offsets, :63-68)")
        # The reference upsamples the frames (F.interpolate(video / 255, scale_factor = 1 / ratio, bilinear) * 255, cast back,
        # :43-50) but keeps res_h / res_w of the ORIGINAL frames (:41) for the grid and the draws below: the patches are cut from
        # the upsampled frames at the small source's offsets.  Reproduced as it is; kvq_upsample_frames is ATen-CPU-exact.
        video = kernels.upsample_frames(video.contiguous(), 1 / ratio)
    assert T % aligned == 0, "Please provide match vclip and align index"
    nt = T // aligned
    hl, wl = H // fragments_h, W // fragments_w
    if rnd_h is None:
        rnd_h = (torch.randint(hl - fsize_h, (fragments_h, fragments_w, nt)) if hl > fsize_h
                 else torch.zeros((fragments_h, fragments_w, nt)).int())
    if rnd_w is None:
        rnd_w = (torch.randint(wl - fsize_w, (fragments_h, fragments_w, nt)) if wl > fsize_w
                 else torch.zeros((fragments_h, fragments_w, nt)).int())
    rnd_h = torch.as_tensor(np.asarray(rnd_h)) if not torch.is_tensor(rnd_h) else rnd_h
    rnd_w = torch.as_tensor(np.asarray(rnd_w)) if not torch.is_tensor(rnd_w) else rnd_w
    hoff = (rnd_h.cpu().long() + torch.tensor(_grid(H, fragments_h, fsize_h)).view(-1, 1, 1)).int()
    woff = (rnd_w.cpu().long() + torch.tensor(_grid(W, fragments_w, fsize_w)).view(1, -1, 1)).int()
    if lazy:
        return kernels.FragmentSource([video.contiguous()], [hoff.to(video.device)], [woff.to(video.device)], fragments_h,
                                      fragments_w, fsize_h, fsize_w, aligned, mean=mean, std=std)
    return kernels.fragment_gather(video.contiguous(), hoff.to(video.device), woff.to(video.device), fragments_h,
                                   fragments_w, fsize_h, fsize_w, aligned, mean=mean, std=std)


def get_resized_video(video, size_h=224, size_w=224, random_crop=False, arp=False, mean=None, std=None, **kwargs):
    """Reference ``get_resized_video`` (:244-252) on the GPU: (C,T,H,W) -> (C,T,size_h,size_w), bilinear."""
    if random_crop:
        raise NotImplementedError("RandomResizedCrop is a training augmentation")
    if arp:
        ratio = video.shape[-2] / video.shape[-1]
        if ratio > 1:
            size_h = int(ratio * size_w)
        elif ratio < 1:
            size_w = int(size_h / ratio)
    return kernels.resize_bilinear(video.contiguous(), size_h, size_w, mean=mean, std=std)


def get_resizecrop_video(video, resize=520, crop=448, phase="test", mean=None, std=None, **kwargs):
    """Reference ``get_resizecrop_video`` (:299-316), test phase: resize to (resize,resize) then the centre
    crop [r//2-crop//2 : r//2+crop//2] — fused with the normalisation in one kernel."""
    if phase == "train":
        raise NotImplementedError("random crop is a training augmentation")
    o = resize // 2 - crop // 2
    n = (resize // 2 + crop // 2) - o
    return kernels.resize_bilinear(video.contiguous(), resize, resize, crop=(o, o, n, n), mean=mean, std=std)


def get_single_view(video, sample_type="aesthetic", **kwargs):
    """Reference ``get_single_view`` (:350-361)."""
    if sample_type.startswith("aesthetic"):
        return get_resized_video(video, **kwargs)
    if sample_type.startswith("technical"):
        return get_spatial_fragments(video, **kwargs)
    if sample_type.startswith("simpleVQA"):
        return get_resizecrop_video(video, **kwargs)
    raise NotImplementedError


SIMPLEVQA_MEAN = (0.485, 0.456, 0.406)      # applied to 0-255 pixels WITHOUT /255, as the reference does
SIMPLEVQA_STD = (0.229, 0.224, 0.225)       # (fusion_datasets.py:811-812, 903)


class SyntheticSimpleVQADataset(torch.utils.data.Dataset):
    """Seeded stand-in for ``ViewDecompositionDataset_add_forSimpleVQA`` (:786-927): ``simpleVQA`` view =
    8 frames (clip_len 8 x frame_interval 10... positional quirk as in the reference) resized 520 -> centre
    crop 448, normalised with the reference's constants; ``feat`` = the (8, 2304) SlowFast features, read from
    ``data_prefix_3D/<video_name>/feature_{i}_{slow,fast}_feature.npy`` (:878-890) or, when
    ``compute_feat`` is set, produced in-process by the HIP SlowFast branch (BASELINE config C3)."""

    def __init__(self, opt, namelist=None, device=None):
        self.opt, self.device = opt, _default_device(device)
        self.n = int(opt.get("num_videos", 4))
        self.frames, self.h, self.w = int(opt.get("frames", 256)), int(opt.get("height", 540)), int(opt.get("width", 960))
        self.sopt = dict(opt["sample_types"]["simpleVQA"])
        s = self.sopt
        # reference: UnifiedFrameSampler(clip_len // t_frag, t_frag, frame_interval, num_clips) (:836-841)
        self.sampler = UnifiedFrameSampler(s["clip_len"] // s["t_frag"], s["t_frag"], s["frame_interval"], s["num_clips"])
        self.data_prefix_3D = opt.get("data_prefix_3D")
        self.slowfast = opt.get("compute_feat")
        g = np.random.Generator(np.random.PCG64(4321))
        self.labels = list(opt.get("labels") or g.uniform(1.0, 5.0, self.n))

    def __len__(self):
        return self.n

    def _feat(self, i, frames_u8):
        import os
        name = f"synthetic_{i:05d}"
        if self.data_prefix_3D and os.path.isdir(os.path.join(self.data_prefix_3D, name)):
            rows = []
            for k in range(8):
                slow = np.load(os.path.join(self.data_prefix_3D, name, f"feature_{k}_slow_feature.npy")).squeeze()
                fast = np.load(os.path.join(self.data_prefix_3D, name, f"feature_{k}_fast_feature.npy")).squeeze()
                rows.append(np.concatenate([slow, fast]))
            return torch.from_numpy(np.stack(rows)).float()
        if self.slowfast is not None:          # 8 clips x 32 frames @224^2, mean .45 / std .225 (SlowFast_features.py:173-174)
            rows = []
            with torch.no_grad():
                for k in range(8):
                    clip = frames_u8[:, k * 32:(k + 1) * 32].to(self.device)
                    x = kernels.resize_bilinear(clip.contiguous(), 224, 224, mean=(0.45 * 255,) * 3, std=(0.225 * 255,) * 3)
                    slow, fast = self.slowfast.forward_clips(x.unsqueeze(0))
                    rows.append(torch.cat([slow.reshape(-1), fast.reshape(-1)]))
            return torch.stack(rows)
        return torch.zeros(8, 2304)

    def __getitem__(self, i):
        from ..utils import synth
        frames = torch.from_numpy(synth.synth_video_u8(1234 + i, self.frames, self.h, self.w))
        inds = self.sampler(self.frames)
        clip = frames[:, torch.from_numpy(inds.astype(np.int64))].to(self.device)
        view = get_resizecrop_video(clip, self.sopt["resize"], self.sopt["crop"], "test", mean=SIMPLEVQA_MEAN,
                                    std=SIMPLEVQA_STD)
        return {"simpleVQA": view, "feat": self._feat(i, frames).unsqueeze(0), "num_clips": {"simpleVQA": self.sopt["num_clips"]},
                "frame_inds": inds, "label": float(self.labels[i]), "name": f"synthetic_{i:05d}",
                "video_name": f"synthetic_{i:05d}.mp4"}


class UnifiedFrameSampler:
    """Reference ``UnifiedFrameSampler`` (:612-660): same constructor, same RNG calls, same indices."""

    def __init__(self, fsize_t, fragments_t, frame_interval=1, num_clips=1, drop_rate=0.0):
        self.fragments_t, self.fsize_t = fragments_t, fsize_t
        self.size_t = fragments_t * fsize_t
        self.frame_interval, self.num_clips, self.drop_rate = frame_interval, num_clips, drop_rate

    def get_frame_indices(self, num_frames, train=False):
        tgrids = np.array([num_frames // self.fragments_t * i for i in range(self.fragments_t)], dtype=np.int32)
        tlength = num_frames // self.fragments_t
        if tlength > self.fsize_t * self.frame_interval:
            rnd_t = np.random.randint(0, tlength - self.fsize_t * self.frame_interval, size=len(tgrids))
        else:
            rnd_t = np.zeros(len(tgrids), dtype=np.int32)
        ranges_t = np.arange(self.fsize_t)[None, :] * self.frame_interval + rnd_t[:, None] + tgrids[:, None]
        drop = _pyrandom.sample(list(range(self.fragments_t)), int(self.fragments_t * self.drop_rate))
        return np.concatenate([rt for i, rt in enumerate(ranges_t) if i not in drop])

    def __call__(self, total_frames, train=False, start_index=0):
        inds = np.concatenate([self.get_frame_indices(total_frames) for _ in range(self.num_clips)])
        return np.mod(inds + start_index, total_frames).astype(np.int32)


class SyntheticKVQDataset(torch.utils.data.Dataset):
    """Seeded stand-in for ``ViewDecompositionDataset_KVQ``: item i is a uint8 frame stack drawn from
    PCG64(1234+i) sampled into the ``technical`` view on the GPU.  ``args``:
    ``num_videos, frames, height, width, labels (optional list), sample_types.technical.{fragments_h,
    fragments_w, fsize_h, fsize_w, aligned, clip_len, frame_interval, num_clips}``."""

    def __init__(self, opt, namelist=None, device=None):
        self.opt, self.device = opt, _default_device(device)
        self.n = int(opt.get("num_videos", 8))
        self.frames, self.h, self.w = int(opt.get("frames", 256)), int(opt.get("height", 540)), int(opt.get("width", 960))
        self.sopt = dict(opt["sample_types"]["technical"])
        s = self.sopt
        # the reference passes (clip_len, num_clips, frame_interval) positionally (fusion_datasets.py:962-964):
        # num_clips lands in fragments_t, so T = clip_len * num_clips frames, split into clips by the harness
        self.sampler = UnifiedFrameSampler(s["clip_len"], s.get("num_clips", 1), s.get("frame_interval", 1))
        g = np.random.Generator(np.random.PCG64(4321))
        self.labels = list(opt.get("labels") or g.uniform(1.0, 5.0, self.n))

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from ..utils import synth
        s = self.sopt
        frames = torch.from_numpy(synth.synth_video_u8(1234 + i, self.frames, self.h, self.w))
        if self.opt.get("seed_per_item"):
            # the samplers draw from the process-global RNGs like the reference's (a1 / a2): seeding them per item makes item i
            # the same whatever rank builds it and whatever was built before (the 1-rank vs N-rank rehearsal of C4)
            np.random.seed(1234 + i)
            _pyrandom.seed(1234 + i)
            torch.manual_seed(1234 + i)
        inds = self.sampler(self.frames)
        clip = frames[:, torch.from_numpy(inds.astype(np.int64))].to(self.device)
        tech = get_spatial_fragments(clip, s["fragments_h"], s["fragments_w"], s["fsize_h"], s["fsize_w"],
                                     aligned=s.get("aligned", 8), mean=KVQ_MEAN, std=KVQ_STD, lazy=bool(s.get("lazy", False)))
        return {"technical": tech, "num_clips": {"technical": s.get("num_clips", 1)}, "frame_inds": inds,
                "label": float(self.labels[i]), "name": f"synthetic_{i:05d}", "video_name": f"synthetic_{i:05d}.mp4"}


# ------------------------------------------------------------------------------------------------------------------
# The reference's dataset classes, under their own names, for its config/*.yml files (``data.val.type``).  Decode is
# outside the built scope: frames come from a reader object — decord when it is installed, a uint8
# ``[T, H, W, 3]`` ``.npy`` stack otherwise — and go to the device as uint8; sampling + normalisation run there.
class NpyFrameReader:
    """``len(reader)`` frames, ``reader[i]`` -> uint8 (H, W, 3): the slice of decord.VideoReader the datasets use."""

    def __init__(self, path):
        self.frames = np.load(path, mmap_mode="r")
        if self.frames.ndim != 4 or self.frames.shape[-1] != 3 or self.frames.dtype != np.uint8:
            raise ValueError(f"{path}: expected a uint8 [T, H, W, 3] frame stack, got {self.frames.dtype} {self.frames.shape}")

    def __len__(self):
        return self.frames.shape[0]

    def __getitem__(self, i):
        return np.asarray(self.frames[int(i)])

    def read_into(self, indices, out):
        """frames[indices] -> out (len(indices), H, W, 3) uint8, one copy straight from the mapped file"""
        np.take(self.frames, np.asarray(indices, np.int64), axis=0, out=out)


class Cv2FrameReader:
    """The reference's OpenCV fallback (fusion_datasets.py:398-431) as a frame reader: every frame of the file read in order by
    ``cv2.VideoCapture.read()`` and kept AS cv2 RETURNS IT (BGR — the reference stacks ``frame`` without a colour conversion, so the
    fallback path feeds BGR where decord feeds RGB; reproduced, not fixed), and a video of 130 frames or fewer padded with copies of
    its LAST frame until it holds 131 (``while len(video_frame_array) <= 130``, :413-415).  A file cv2 cannot read a single frame
    from makes the reference fail in ``np.stack`` on ``None``; here it is a ValueError naming the file.
    ``cv2`` is imported lazily (it is not part of this image: the logic is exercised in tests with a stand-in module)."""
    MIN_FRAMES = 131

    def __init__(self, path, cv2_module=None):
        if cv2_module is None:
            import cv2 as cv2_module  # noqa: N813
        cap = cv2_module.VideoCapture(path)
        frames, last = [], None
        while True:
            ret, frame = cap.read()
            if not ret:
                break
            last = frame
            frames.append(frame)
        if hasattr(cap, "release"):
            cap.release()
        if last is None:
            raise ValueError(f"{path}: OpenCV could not decode a single frame")
        while len(frames) < self.MIN_FRAMES:                 # 'too short' (:413-415)
            frames.append(last)
        self.frames = np.stack(frames, axis=0)
        if self.frames.dtype != np.uint8 or self.frames.ndim != 4:
            raise ValueError(f"{path}: expected uint8 (H, W, 3) frames from OpenCV, got {self.frames.dtype} {self.frames.shape[1:]}")

    def __len__(self):
        return self.frames.shape[0]

    def __getitem__(self, i):
        return self.frames[int(i)]

    def read_into(self, indices, out):
        np.take(self.frames, np.asarray(indices, np.int64), axis=0, out=out)


def open_video(path):
    """Frame reader for ``path``: ``<path>`` itself or ``<path>.npy`` as a frame stack (this build's decode-free entry), else
    decord.VideoReader (fusion_datasets.py:381-383), else — decord missing, or failing on this file: the reference wraps the decord
    branch in a bare ``try`` — the OpenCV fallback (:398-431, ``Cv2FrameReader``)."""
    import os
    if path.endswith(".npy"):
        return NpyFrameReader(path)
    if os.path.exists(path + ".npy"):
        return NpyFrameReader(path + ".npy")
    decord_error = None
    try:
        from decord import VideoReader
        return VideoReader(path)
    except Exception as e:  # noqa: BLE001  (the reference: ``except:`` around the whole decord branch)
        decord_error = e
    try:
        return Cv2FrameReader(path)
    except ImportError as e:
        raise ImportError(f"cannot read {path}: video decode needs decord or OpenCV (neither is part of this image: SURVEY.md §8 f2; "
                          f"decord: {type(decord_error).__name__}: {decord_error}) — or provide the decoded frames as a uint8 "
                          f"[T,H,W,3] array in {path}.npy") from e


class _Staging:
    """Reusable PINNED host buffers for the frames of one video (two per thread, alternating: the H2D copy of one video
    overlaps the host-side gather of the next; a buffer is rewritten only after its copy's event has completed)."""
    _local = threading.local()

    @classmethod
    def get(cls, nbytes):
        st = cls._local.__dict__.setdefault("slots", {"i": 0, "buf": [None, None], "ev": [None, None]})
        k = st["i"] = 1 - st["i"]
        if st["buf"][k] is None or st["buf"][k].numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25), dtype=torch.uint8)
            st["buf"][k] = buf.pin_memory() if torch.cuda.is_available() else buf
        elif st["ev"][k] is not None:
            st["ev"][k].synchronize()
        return st, k


_COPY_POOL = None
_COPY_THREADS = 4


def _frames_to_device(vr, uniq, device):
    """The sampled frames ``uniq`` of a reader -> ONE uint8 (n, H, W, 3) device tensor: gathered into pinned staging memory
    by a few host threads (numpy copies release the GIL), then a single asynchronous H2D copy on the current stream."""
    global _COPY_POOL
    first = vr[int(uniq[0])]
    first = first.asnumpy() if hasattr(first, "asnumpy") else np.asarray(first)
    n, shape = len(uniq), tuple(first.shape)
    nbytes = n * int(np.prod(shape))
    st, k = _Staging.get(nbytes)
    stage = st["buf"][k][:nbytes].view((n,) + shape)
    host = stage.numpy()
    if hasattr(vr, "read_into"):
        if _COPY_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _COPY_POOL = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="kvq-copy")
        nt = max(1, min(_COPY_THREADS, n // 8))
        bounds = np.linspace(0, n, nt + 1).astype(int)
        jobs = [_COPY_POOL.submit(vr.read_into, uniq[a:b], host[a:b]) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        for j in jobs:
            j.result()
    elif hasattr(vr, "get_batch"):                      # decord: one decode call for all frames
        host[...] = vr.get_batch([int(i) for i in uniq]).asnumpy()
    else:
        host[0] = first
        for j in range(1, n):
            f = vr[int(uniq[j])]
            host[j] = f.asnumpy() if hasattr(f, "asnumpy") else np.asarray(f)
    dev = stage.to(device, non_blocking=True)
    if dev.is_cuda:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev.device))
        st["ev"][k] = ev
    return dev


def _sampled_clips(path, samplers, is_train, device):
    """Reference ``spatial_temporal_view_decomposition`` (:376-397), decode half: one reader, every frame fetched once and
    sent to the device once, as uint8; per view a uint8 (3, T, H, W) device tensor (frame gather + layout change in HBM) +
    the sampled indices."""
    vr = open_video(path)
    frame_inds = {k: s(len(vr), is_train) for k, s in samplers.items()}
    uniq = np.unique(np.concatenate(list(frame_inds.values()), 0))
    frames = _frames_to_device(vr, uniq, device)                               # (n, H, W, 3) uint8
    video = {}
    for k, inds in frame_inds.items():
        pos = torch.from_numpy(np.searchsorted(uniq, inds).astype(np.int64)).to(frames.device)
        video[k] = frames.index_select(0, pos).permute(3, 0, 1, 2).contiguous()
    return video, frame_inds


def _build_samplers(sample_types, phase):
    samplers = {}
    for stype, sopt in sample_types.items():
        if "t_frag" not in sopt:      # (clip_len, num_clips, frame_interval) positionally, as the reference (:962-964)
            samplers[stype] = UnifiedFrameSampler(sopt["clip_len"], sopt["num_clips"], sopt["frame_interval"])
        else:
            samplers[stype] = UnifiedFrameSampler(sopt["clip_len"] // sopt["t_frag"], sopt["t_frag"], sopt["frame_interval"],
                                                  sopt["num_clips"])
        print(stype + " branch sampled frames:", samplers[stype](40, phase == "train"))       # draws from the RNG, as there
    return samplers


def _default_device(device):
    """``None`` -> the process's CURRENT HIP device (rank r of a torch.distributed.run job has set cuda:r): the K1
    kernels are launched on the current device's stream, so the frames must live there."""
    if device is not None:
        return torch.device(device)
    return torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)


class ViewDecompositionDataset_add_forSimpleVQA(torch.utils.data.Dataset):  # noqa: N801  (reference spelling)
    """Reference class of the same name (fusion_datasets.py:786-927): csv ``filename,label`` with a header row, per
    video the ``simpleVQA`` view (resize 520 -> centre crop 448, normalised with the ImageNet constants on 0-255
    pixels — reproduced as-is) and ``feat`` = 8 rows of SlowFast features from
    ``data_prefix_3D/<video_name>/feature_{i}_{slow,fast}_feature.npy``.  Same dict keys."""

    def __init__(self, opt, namelist=None, device=None):
        import csv
        import os.path as osp
        super().__init__()
        self.opt, self.namelist, self.device = opt, namelist, _default_device(device)
        self.ann_file, self.data_prefix, self.data_prefix_3D = opt["anno_file"], opt["data_prefix"], opt["data_prefix_3D"]
        self.sample_types, self.feature_type, self.phase = opt["sample_types"], opt["feature_type"], opt["phase"]
        self.augment = opt.get("augment", False)
        if self.phase == "train" or self.augment:
            raise NotImplementedError("training-phase sampling / augmentation: this is an inference engine")
        self.samplers = _build_samplers(self.sample_types, self.phase)
        if isinstance(self.ann_file, list):
            self.video_infos = self.ann_file
        else:
            self.video_infos = []
            with open(self.ann_file, newline="") as f:
                rows = csv.reader(f)
                next(rows)
                for row in rows:
                    self.video_infos.append(dict(filename=osp.join(self.data_prefix, row[0]), label=float(row[1]), video_name=row[0]))
            scores = [v["label"] for v in self.video_infos]
            self.max, self.min = max(scores), min(scores)
        self.labels = [v["label"] for v in self.video_infos]
        self.video_names = [v["video_name"] for v in self.video_infos]

    def __len__(self):
        return len(self.video_infos)

    def _features(self, video_name):
        import os
        folder = os.path.join(self.data_prefix_3D, video_name)
        parts = {"Slow": ("slow",), "Fast": ("fast",), "SlowFast": ("slow", "fast")}[self.feature_type]
        rows = []
        for i in range(8):
            rows.append(torch.cat([torch.from_numpy(np.load(os.path.join(folder, f"feature_{i}_{p}_feature.npy"))).squeeze().float().reshape(-1)
                                   for p in parts]))
        return torch.stack(rows)

    def __getitem__(self, index):
        info = self.video_infos[index]
        feat = self._features(info["video_name"])
        video, frame_inds = _sampled_clips(info["filename"], self.samplers, False, self.device)
        data = {}
        for stype, sopt in self.sample_types.items():
            kw = dict(sopt, phase="test")
            data[stype] = get_single_view(video[stype], stype, mean=SIMPLEVQA_MEAN, std=SIMPLEVQA_STD, **kw)
        data["num_clips"] = {k: s["num_clips"] for k, s in self.sample_types.items()}
        data["clip_len"] = {k: s["clip_len"] for k, s in self.sample_types.items()}
        data["frame_inds"], data["label"], data["video_name"] = frame_inds, info["label"], info["video_name"]
        if "simpleVQA" in data:
            data["feat"] = feat
        data["name"] = info["filename"]
        return data


class ViewDecompositionDataset_KVQ(torch.utils.data.Dataset):  # noqa: N801  (reference spelling)
    """Reference class of the same name (fusion_datasets.py:930-1051): lines ``filename,cls_label,dis_label,label``;
    per video the view(s) of ``sample_types`` normalised with the KVQ constants, plus the KSVQE inputs:
    ``resize_video`` (``get_resized_video``, /255 then the CLIP constants), ``fragment`` (= the normalised view),
    ``ori_fragment`` (a second, un-normalised fragment draw, as the reference makes), ``dis_label``,
    ``original_shape``.  Same dict keys."""

    CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
    CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, opt, namelist=None, device=None):
        import os.path as osp
        super().__init__()
        self.opt, self.namelist, self.device = opt, namelist, _default_device(device)
        self.ann_file, self.data_prefix = opt["anno_file"], opt["data_prefix"]
        self.sample_types, self.phase = opt["sample_types"], opt["phase"]
        self.augment = opt.get("augment", False)
        if self.phase == "train" or self.augment:
            raise NotImplementedError("training-phase sampling / augmentation: this is an inference engine")
        self.samplers = _build_samplers(self.sample_types, self.phase)
        if isinstance(self.ann_file, list):
            self.video_infos = self.ann_file
        else:
            self.video_infos = []
            with open(self.ann_file, "r") as f:
                for line in f:
                    filename, cls_label, dis_label, label = line.strip().split(",")
                    self.video_infos.append(dict(filename=osp.join(self.data_prefix, filename), label=float(label),
                                                 cls_label=int(float(cls_label)), dis_label=int(float(dis_label)),
                                                 video_name=filename))
            scores = [v["label"] for v in self.video_infos]
            self.max, self.min = max(scores), min(scores)
        self.labels = [v["label"] for v in self.video_infos]
        self.video_names = [v["video_name"] for v in self.video_infos]

    def __len__(self):
        return len(self.video_infos)

    def __getitem__(self, index):
        info = self.video_infos[index]
        video, frame_inds = _sampled_clips(info["filename"], self.samplers, False, self.device)
        data, k = {}, None
        resize = ori = None
        for stype, sopt in self.sample_types.items():       # order of the reference's three calls per view (:455-459)
            kw = dict(sopt, phase="test")
            data[stype] = get_single_view(video[stype], stype, mean=KVQ_MEAN, std=KVQ_STD, **kw)
            resize = get_resized_video(video[stype], mean=tuple(255.0 * m for m in self.CLIP_MEAN),
                                       std=tuple(255.0 * s for s in self.CLIP_STD),
                                       **{a: b for a, b in kw.items() if a in ("size_h", "size_w", "random_crop", "arp")})
            ori = get_spatial_fragments(video[stype], **{a: b for a, b in kw.items() if a in (
                "fragments_h", "fragments_w", "fsize_h", "fsize_w", "aligned", "nfrags")})
            k = stype
        data["resize_video"], data["fragment"], data["ori_fragment"] = resize, data[k], ori
        data["num_clips"] = {s: o["num_clips"] for s, o in self.sample_types.items()}
        data["clip_len"] = {s: o["clip_len"] for s, o in self.sample_types.items()}
        data["frame_inds"], data["dis_label"] = frame_inds, info["dis_label"]
        data["name"], data["video_name"] = info["filename"], info["video_name"]
        data["original_shape"] = tuple(video[k].shape[1:])
        data["label"] = info["label"]
        return data
