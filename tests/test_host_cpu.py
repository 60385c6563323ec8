"""CPU: host-side logic that needs no GPU — temporal sampler vs the reference's golden indices, harness
math, config dispatch of the test.py drop-in."""
import random

import numpy as np
import pytest

import kvq_amd  # noqa: F401
from kvq_amd.datasets import UnifiedFrameSampler
from kvq_amd.trainer import Trainer
from oracle import sampler_oracle as SO


def test_unified_frame_sampler_matches_reference(golden):
    g = golden("sampler.npz")
    for tag in ("ksvqe", "simple", "short", "clips3"):
        n, fs_t, ft, iv, nc, seed = (int(v) for v in g[f"frames/{tag}/meta"])
        np.random.seed(seed)
        random.seed(seed)
        idx = UnifiedFrameSampler(fs_t, ft, iv, nc)(n)
        assert idx.dtype == np.int32 and np.array_equal(idx, g[f"frames/{tag}/idx"]), tag


def test_rescale_and_metrics(golden):
    g = golden("sampler.npz")
    rng = np.random.Generator(np.random.PCG64(900))
    labels = rng.uniform(1, 5, 900)
    preds = 0.3 * labels + rng.standard_normal(900) * 0.2 - 1.0
    p = Trainer.rescale(None, list(preds), list(labels))
    assert np.allclose(p[:8], g["metrics/rescaled_head"], rtol=0, atol=1e-12)
    assert np.allclose(Trainer.rescale(None, preds), SO.rescale(preds), rtol=0, atol=1e-12)


def test_fragment_sampler_host_asserts():
    import torch
    from kvq_amd.datasets import get_spatial_fragments
    v = torch.zeros(3, 10, 224, 224)
    with pytest.raises(AssertionError, match="Please provide match vclip and align index"):
        get_spatial_fragments(v, aligned=8)
    from kvq_amd import _abi
    with pytest.raises(_abi.KvqError):                 # the upsample fallback (fusion_datasets.py:43-50) is a HIP launch: no CPU path
        get_spatial_fragments(torch.zeros(3, 8, 100, 100), aligned=8)
    with pytest.raises(NotImplementedError, match="fallback_type"):
        get_spatial_fragments(torch.zeros(3, 8, 100, 100), aligned=8, fallback_type="pad")
    with pytest.raises(ValueError, match="smaller than one"):
        get_spatial_fragments(torch.zeros(3, 8, 20, 100), aligned=8)


# ------------------------------------------------------------------ checkpoint formats
@pytest.mark.parametrize("case", ["inflate_w7", "inflate_w12", "load_swin"])
def test_checkpoint_loaders_match_reference(golden, case, tmp_path, capsys):
    """2D -> 3D inflation (swin_backbone.py:858-931) and the Video-Swin loader with its table fork
    (:933-1006): the trunk's state dict after loading a synthetic checkpoint == what the reference left there."""
    import torch
    from kvq_amd.models.backbones.swin_backbone import SwinTransformer3D
    from kvq_amd.utils import synth
    g = golden("ckpt.npz")
    path = str(tmp_path / (case + ".pth"))
    if case.startswith("inflate"):
        body = {"model": {k: torch.from_numpy(v) for k, v in synth.synth_swin2d_checkpoint(synth.SWIN_T_GRPB, 5, int(case[9:])).items()}}
        torch.save(body, path)
        m = SwinTransformer3D(pretrained=path, pretrained2d=True)          # the constructor path (init_weights)
    else:
        body = {"state_dict": {k: torch.from_numpy(v) for k, v in synth.synth_swin3d_checkpoint(synth.SWIN_T_GRPB, 6).items()}}
        torch.save(body, path)
        m = SwinTransformer3D(pretrained=path)
    capsys.readouterr()
    sd = m.state_dict()
    for k in g[f"{case}/keys"]:
        k = str(k)
        a = sd[k].numpy().astype(np.float32).reshape(-1)
        assert list(sd[k].shape) == list(g[f"{case}/{k}/shape"]), k
        assert np.allclose(a[g[f"{case}/{k}/idx"]], g[f"{case}/{k}/val"], rtol=0, atol=1e-6), k
        assert abs(float(a.astype(np.float64).sum()) - float(g[f"{case}/{k}/sum"])) <= 1e-3 * max(1.0, float(g[f"{case}/{k}/asum"])) * 1e-3, k
    if case == "load_swin":      # the fork: fragment tables are copies of the relative tables; the bad-shape key stayed at init
        t = sd["layers.0.blocks.0.attn.relative_position_bias_table"]
        assert torch.equal(t, sd["layers.0.blocks.0.attn.fragment_position_bias_table"])
        assert torch.equal(sd["norm.weight"], torch.ones(768))


def test_trainer_strips_dataparallel_prefix(tmp_path, capsys):
    """DataParallel / DDP checkpoints carry a ``module.`` prefix (trainer.py:62-74)."""
    import torch
    from kvq_amd.models.model import VQA_Network
    cfg = {"model": {"args": {"swin_tiny_grpb": {"head": {"in_channels": 768, "hidden_channels": 64}}}}}
    src = VQA_Network(cfg)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.25)
    path = str(tmp_path / "dp.pth")
    torch.save({"module." + k: v for k, v in src.state_dict().items()}, path)
    dst = VQA_Network(cfg)
    msg = Trainer.load_checkpoint(dst, path)
    capsys.readouterr()
    assert not msg.unexpected_keys and not msg.missing_keys
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k


# ------------------------------------------------------------------ the reference's dataset classes (host half)
def _write_videos(tmp_path, n=3, T=64, H=60, W=80):
    g = np.random.Generator(np.random.PCG64(77))
    for i in range(n):
        np.save(str(tmp_path / f"v{i}.mp4.npy"), g.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8))
    return [f"v{i}.mp4" for i in range(n)]


def test_reference_named_datasets_parse_like_the_reference(tmp_path, capsys):
    """ViewDecompositionDataset_KVQ / _add_forSimpleVQA (fusion_datasets.py:786-1051): annotation formats, sampler
    construction (positional quirk included), the frame reader; no GPU work (``__getitem__`` is a gpu test)."""
    from kvq_amd.datasets import (NpyFrameReader, ViewDecompositionDataset_add_forSimpleVQA, ViewDecompositionDataset_KVQ,
                                  open_video)
    names = _write_videos(tmp_path)
    (tmp_path / "kvq.txt").write_text("".join(f"{n},{i},{2 * i},{3.5 - 0.5 * i}\n" for i, n in enumerate(names)))
    (tmp_path / "simple.csv").write_text("filename,score\n" + "".join(f"{n},{1.0 + i}\n" for i, n in enumerate(names)))
    kv = ViewDecompositionDataset_KVQ(dict(anno_file=str(tmp_path / "kvq.txt"), data_prefix=str(tmp_path), phase="test",
                                           sample_types={"technical": dict(fragments_h=7, fragments_w=7, fsize_h=8, fsize_w=8,
                                                                           aligned=8, clip_len=32, frame_interval=2, num_clips=3)}))
    assert len(kv) == 3 and kv.video_infos[1] == dict(filename=str(tmp_path / "v1.mp4"), label=3.0, cls_label=1, dis_label=2,
                                                        video_name="v1.mp4")
    assert (kv.max, kv.min) == (3.5, 2.5)
    s = kv.samplers["technical"]            # (clip_len, num_clips, frame_interval) land in (fsize_t, fragments_t, interval)
    assert (s.fsize_t, s.fragments_t, s.frame_interval, s.num_clips) == (32, 3, 2, 1)
    sv = ViewDecompositionDataset_add_forSimpleVQA(dict(
        anno_file=str(tmp_path / "simple.csv"), data_prefix=str(tmp_path), data_prefix_3D=str(tmp_path / "feat"), feature_type="SlowFast",
        phase="test", sample_types={"simpleVQA": dict(resize=520, crop=448, clip_len=8, frame_interval=10, t_frag=8, num_clips=1)}))
    assert len(sv) == 3 and sv.labels == [1.0, 2.0, 3.0] and sv.video_names == names
    s = sv.samplers["simpleVQA"]
    assert (s.fsize_t, s.fragments_t, s.frame_interval, s.num_clips) == (1, 8, 10, 1)
    assert "branch sampled frames" in capsys.readouterr().out
    r = open_video(str(tmp_path / "v0.mp4"))
    assert isinstance(r, NpyFrameReader) and len(r) == 64 and r[5].shape == (60, 80, 3) and r[5].dtype == np.uint8
    with pytest.raises(ImportError, match="decord"):
        open_video(str(tmp_path / "missing.mp4"))
    with pytest.raises(NotImplementedError):
        ViewDecompositionDataset_KVQ(dict(anno_file=[], data_prefix="", phase="train", sample_types={}))


def test_pack_pathway_output_provenance_tag():
    """slowfast.forward re-selects the slow frames on the device only for a slow tensor that IS pack_pathway_output's selection of
    that very fast tensor: an in-place edit of either tensor, or another tensor at the same address, makes the pair an ordinary one
    (the reference consumes slow as given, SlowFast_features.py:112-135)."""
    import torch
    from kvq_amd.models.backbones import slowfast_model as sf
    frames = torch.arange(2 * 3 * 32 * 4 * 4, dtype=torch.float32).reshape(2, 3, 32, 4, 4)
    slow, fast = sf.pack_pathway_output(frames)
    assert sf._is_packed_pair(slow, fast) and slow.shape[2] == 8
    assert not sf._is_packed_pair(slow, fast.clone())                 # same values, another tensor
    other = sf.pack_pathway_output(frames.clone())
    assert not sf._is_packed_pair(slow, other[1]) and not sf._is_packed_pair(other[0], fast)
    slow.mul_(2.0)                                                    # the caller edited the slow pathway: consume it as given
    assert not sf._is_packed_pair(slow, fast)
    slow2, fast2 = sf.pack_pathway_output(frames)
    fast2.add_(1.0)
    assert not sf._is_packed_pair(slow2, fast2)


def test_open_video_reader_order_and_cv2_fallback_padding(tmp_path, monkeypatch):
    """``spatial_temporal_view_decomposition``'s reader selection (fusion_datasets.py:379-431): decord first, and when decord is missing or
    fails on the file, OpenCV — every frame kept as cv2 returns it (BGR, no conversion), a video of <= 130 frames padded with its LAST frame
    to 131 (:413-415).  Neither codec is part of this image: a stand-in ``cv2`` module serves synthetic frames (the logic is what is pinned)."""
    import sys
    import types

    import numpy as np
    import pytest

    import kvq_amd  # noqa: F401
    from kvq_amd.datasets import fusion_datasets as FD

    rng = np.random.Generator(np.random.PCG64(5))
    clips = {"short.mp4": rng.integers(0, 256, size=(17, 6, 8, 3), dtype=np.uint8),
             "long.mp4": rng.integers(0, 256, size=(140, 6, 8, 3), dtype=np.uint8), "empty.mp4": np.zeros((0, 6, 8, 3), np.uint8)}

    class Capture:
        def __init__(self, path=None):
            self.frames, self.i = clips[str(path).split("/")[-1]], 0

        def read(self):
            if self.i >= len(self.frames):
                return False, None
            self.i += 1
            return True, self.frames[self.i - 1]

        def release(self):
            pass

    fake = types.ModuleType("cv2")
    fake.VideoCapture = Capture
    monkeypatch.setitem(sys.modules, "cv2", fake)
    monkeypatch.setitem(sys.modules, "decord", None)                 # `from decord import VideoReader` -> ImportError -> the fallback
    short = FD.open_video(str(tmp_path / "short.mp4"))
    assert isinstance(short, FD.Cv2FrameReader) and len(short) == 131
    assert np.array_equal(short.frames[:17], clips["short.mp4"])                       # BGR kept as delivered
    assert all(np.array_equal(short[i], clips["short.mp4"][-1]) for i in (17, 60, 130))  # padded with the LAST frame
    long = FD.open_video(str(tmp_path / "long.mp4"))
    assert len(long) == 140 and np.array_equal(long.frames, clips["long.mp4"])
    out = np.empty((3, 6, 8, 3), np.uint8)
    long.read_into([0, 139, 7], out)
    assert np.array_equal(out, clips["long.mp4"][[0, 139, 7]])
    with pytest.raises(ValueError, match="could not decode"):
        FD.open_video(str(tmp_path / "empty.mp4"))
    # a decord that FAILS on the file falls through to OpenCV as well (the reference's bare `except:` around the decord branch)
    broken = types.ModuleType("decord")

    def _raise(path):
        raise RuntimeError("decord: unsupported stream")
    broken.VideoReader = _raise
    monkeypatch.setitem(sys.modules, "decord", broken)
    assert len(FD.open_video(str(tmp_path / "short.mp4"))) == 131
    # the decode-free entry of this build still wins, and with neither codec the error says what to provide
    np.save(tmp_path / "x.mp4.npy", clips["short.mp4"])
    assert isinstance(FD.open_video(str(tmp_path / "x.mp4")), FD.NpyFrameReader)
    monkeypatch.setitem(sys.modules, "cv2", None)
    monkeypatch.setitem(sys.modules, "decord", None)
    with pytest.raises(ImportError, match="decord or OpenCV"):
        FD.open_video(str(tmp_path / "short.mp4"))
