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
    with pytest.raises(NotImplementedError):
        get_spatial_fragments(torch.zeros(3, 8, 100, 100), aligned=8)


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
