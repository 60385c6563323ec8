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
