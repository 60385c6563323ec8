"""ORACLE (test infrastructure): numpy restatement of the reference samplers and harness math.

Reference lines restated (``/root/reference``):
  datasets/fusion_datasets.py:22-121   get_spatial_fragments (grid-mini-patch sampler)
  datasets/fusion_datasets.py:612-660  UnifiedFrameSampler
  datasets/fusion_datasets.py:1017-1020 (v - mean) / std
  trainer.py:192-201                   clip reshape (b,c,T,h,w) -> (b*nclips,c,T/nclips,h,w)
  trainer.py:356-361                   rescale

The reference draws its offsets from global RNG state inside the functions
(``torch.randint`` :87-98, ``np.random.randint`` :632-635).  Parity needs the *same
drawn offsets* (SURVEY.md §0 trap 6), so every function here takes the drawn offsets
as explicit inputs; ``draw_*`` replays the reference's RNG calls for a seeded state.

Pinned by ``tests/golden/make_golden.py`` against the imported reference.
Only tests, smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np


def fragment_grid(res: int, fragments: int, fsize: int) -> np.ndarray:
    """Cell origins min(res//F * i, res - fsize)  (fusion_datasets.py:64-69)."""
    return np.asarray([min(res // fragments * i, res - fsize) for i in range(fragments)], np.int64)


def draw_fragment_offsets(T: int, H: int, W: int, fragments_h=7, fragments_w=7, fsize_h=32, fsize_w=32,
                          aligned=32, generator=None):
    """Replays the reference's two ``torch.randint`` calls (non-'random' branch, :86-98)."""
    import torch
    nt = T // aligned
    hl, wl = H // fragments_h, W // fragments_w
    kw = {} if generator is None else {"generator": generator}
    if hl > fsize_h:
        rh = torch.randint(hl - fsize_h, (fragments_h, fragments_w, nt), **kw)
    else:
        rh = torch.zeros((fragments_h, fragments_w, nt)).int()
    if wl > fsize_w:
        rw = torch.randint(wl - fsize_w, (fragments_h, fragments_w, nt), **kw)
    else:
        rw = torch.zeros((fragments_h, fragments_w, nt)).int()
    return rh.numpy().astype(np.int32), rw.numpy().astype(np.int32)


def _fma32(a, b, c):
    """fp32 fused multiply-add of fp32 arrays (exact product + sum in float64, one rounding: |a b| < 2^48 ulp-wise)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def upsample_fallback(video: np.ndarray, scale_factor: float) -> np.ndarray:
    """``F.interpolate(video / 255.0, scale_factor=s, mode="bilinear")`` then ``(v * 255.0).type_as(video)``
    (fusion_datasets.py:43-50) for a (C,T,H,W) uint8 or float32 array.  ATen's upsample_bilinear2d, align_corners=False, with
    the scale factor handed to the op: output size floor(size * s), source coordinate float(1 / s) * (dst + 0.5) - 0.5 clamped
    at 0, weights (1 - l, l).  The roundings are those of ATen's CPU kernel as built in this image (fused multiply-adds where
    its compiler contracts them) — pinned bit-exactly by tests/golden/make_golden.py; the cast back to uint8 truncates."""
    import math
    C, T, H, W = video.shape
    OH, OW = math.floor(float(H * scale_factor)), math.floor(float(W * scale_factor))
    scale = np.float32(1.0 / scale_factor)
    x = (video.astype(np.float32) / np.float32(255.0)).astype(np.float32)

    def axis(n_out, n_in):
        i = np.arange(n_out, dtype=np.float32)
        real = np.maximum(_fma32(scale, i + np.float32(0.5), np.float32(-0.5)), np.float32(0))
        i0 = np.minimum(np.floor(real).astype(np.int64), n_in - 1)
        i1 = i0 + (i0 < n_in - 1)
        l1 = np.clip((real - i0.astype(np.float32)).astype(np.float32), 0, 1).astype(np.float32)
        return i0, i1, (np.float32(1) - l1).astype(np.float32), l1

    y0, y1, ly0, ly1 = axis(OH, H)
    x0, x1, lx0, lx1 = axis(OW, W)
    lx0, lx1 = lx0[None, None, None, :], lx1[None, None, None, :]
    ly0, ly1 = ly0[None, None, :, None], ly1[None, None, :, None]
    top, bot = x[:, :, y0], x[:, :, y1]
    r0 = _fma32(top[..., x0], lx0, (top[..., x1] * lx1).astype(np.float32))
    r1 = _fma32(bot[..., x0], lx0, (bot[..., x1] * lx1).astype(np.float32))
    out = (_fma32(r0, ly0, (r1 * ly1).astype(np.float32)) * np.float32(255.0)).astype(np.float32)
    return out.astype(np.uint8) if video.dtype == np.uint8 else out


def spatial_fragments(video: np.ndarray, rnd_h: np.ndarray, rnd_w: np.ndarray, fragments_h=7,
                      fragments_w=7, fsize_h=32, fsize_w=32, aligned=32) -> np.ndarray:
    """video (C,T,H,W) -> (C,T,Fh*fs,Fw*fs); patch (i,j) of t-block t is copied from
    origin (hgrid[i]+rnd_h[i,j,t], wgrid[j]+rnd_w[i,j,t]).  Sources smaller than the canvas take the
    bilinear-upsample fallback (:43-50, ``upsample_fallback``) first — and, as the reference (:41 before :43),
    keep the ORIGINAL frame size for the grid: the patches are cut from the upsampled frames at the small source's offsets."""
    C, T, H, W = video.shape
    if T == 1:
        aligned = 1
    ratio = min(H / (fragments_h * fsize_h), W / (fragments_w * fsize_w))
    if ratio < 1:
        video = upsample_fallback(video, 1 / ratio)
    assert T % aligned == 0, "Please provide match vclip and align index"
    hg, wg = fragment_grid(H, fragments_h, fsize_h), fragment_grid(W, fragments_w, fsize_w)
    out = np.zeros((C, T, fragments_h * fsize_h, fragments_w * fsize_w), video.dtype)
    for i in range(fragments_h):
        for j in range(fragments_w):
            for t in range(T // aligned):
                ho, wo = int(hg[i] + rnd_h[i, j, t]), int(wg[j] + rnd_w[i, j, t])
                out[:, t * aligned:(t + 1) * aligned, i * fsize_h:(i + 1) * fsize_h,
                    j * fsize_w:(j + 1) * fsize_w] = video[:, t * aligned:(t + 1) * aligned,
                                                           ho:ho + fsize_h, wo:wo + fsize_w]
    return out


def normalize(video: np.ndarray, mean, std) -> np.ndarray:
    """(C,T,H,W) float: (v - mean_c) / std_c, fp32 (fusion_datasets.py:1017-1020)."""
    m = np.asarray(mean, np.float32).reshape(-1, 1, 1, 1)
    s = np.asarray(std, np.float32).reshape(-1, 1, 1, 1)
    return (video.astype(np.float32) - m) / s


def draw_frame_offsets(num_frames: int, fsize_t: int, fragments_t: int, frame_interval: int,
                       num_clips: int, rng=np.random):
    """Replays ``np.random.randint(0, tlength - fsize_t*interval, size=fragments_t)`` per clip."""
    tlength = num_frames // fragments_t
    out = []
    for _ in range(num_clips):
        if tlength > fsize_t * frame_interval:
            out.append(rng.randint(0, tlength - fsize_t * frame_interval, size=fragments_t))
        else:
            out.append(np.zeros(fragments_t, dtype=np.int32))
    return np.stack(out)


def frame_indices(num_frames: int, fsize_t: int, fragments_t: int, frame_interval: int,
                  rnd_t: np.ndarray, start_index: int = 0) -> np.ndarray:
    """rnd_t (num_clips, fragments_t) -> int32 indices (num_clips*fragments_t*fsize_t,)
    = (arange(fsize_t)*interval + rnd + tgrid) mod num_frames  (drop_rate = 0)."""
    tgrid = np.asarray([num_frames // fragments_t * i for i in range(fragments_t)], np.int64)
    clips = []
    for r in np.asarray(rnd_t).reshape(-1, fragments_t):
        clips.append((np.arange(fsize_t)[None, :] * frame_interval + r[:, None] + tgrid[:, None]).reshape(-1))
    return np.mod(np.concatenate(clips) + start_index, num_frames).astype(np.int32)


def split_clips(x: np.ndarray, num_clips: int) -> np.ndarray:
    """(b,c,T,h,w) -> (b*num_clips, c, T/num_clips, h, w)  (trainer.py:192-201)."""
    b, c, T, h, w = x.shape
    return (x.reshape(b, c, num_clips, T // num_clips, h, w).transpose(0, 2, 1, 3, 4, 5)
            .reshape(b * num_clips, c, T // num_clips, h, w))


def rescale(pr, gt=None) -> np.ndarray:
    pr = np.asarray(pr, np.float64)
    z = (pr - pr.mean()) / pr.std()
    if gt is None:
        return z
    gt = np.asarray(gt, np.float64)
    return z * gt.std() + gt.mean()


def quality_metrics(preds, labels):
    """SRCC, PLCC, KRCC, RMSE exactly as trainer.py:287-292 computes them."""
    from scipy.stats import kendalltau, pearsonr, spearmanr
    labels = np.asarray(labels, np.float64)
    p = rescale(preds, labels)
    return (float(spearmanr(labels, p)[0]), float(pearsonr(labels, p)[0]),
            float(kendalltau(labels, p)[0]), float(np.sqrt(((labels - p) ** 2).mean())))


def resize_bilinear(video: np.ndarray, rh: int, rw: int, round_u8: bool = False) -> np.ndarray:
    """(C,T,H,W) -> (C,T,rh,rw).  **Parity unpinned**: the reference resizes with
    ``torchvision.transforms.Resize`` (fusion_datasets.py:229-252), and torchvision is absent here
    (un-pinned in requirements.txt: ``torchvision`` without version next to ``torch~=1.10``).  Restated as
    what torchvision's tensor path does: ``F.interpolate(mode="bilinear", align_corners=False)``, no
    antialias (the torch 1.10-era default for tensors), rounding back for integer inputs."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(video)).float()
    out = torch.nn.functional.interpolate(t.permute(1, 0, 2, 3), size=(rh, rw), mode="bilinear", align_corners=False)
    out = out.permute(1, 0, 2, 3)
    if round_u8:
        out = out.round().clamp(0, 255)
    return out.numpy()


def resizecrop(video: np.ndarray, resize: int = 520, crop: int = 448) -> np.ndarray:
    """test-phase ``get_resizecrop_video`` (fusion_datasets.py:299-316): resize to (resize,resize), centre crop."""
    r = resize_bilinear(video, resize, resize, round_u8=(video.dtype == np.uint8))
    h, w = r.shape[-2:]
    return r[..., h // 2 - crop // 2: h // 2 + crop // 2, w // 2 - crop // 2: w // 2 + crop // 2]
