"""GPU parity, end to end: VQA_Network (drop-in boundary) -> libkvq_hip.so vs (a) the golden
fixtures produced by the real reference and (b) the CPU oracle run on the box.

Bar (BASELINE.json north_star): |score_gpu - score_ref| <= 1e-3 per clip.  Feature maps are
compared with a relative-L2 bound (bf16 MFMA operands, fp32 accumulate / LN / softmax / residual)."""
import os

import numpy as np
import pytest
import torch

import kvq_amd  # noqa: F401
from kvq_amd import _abi
from kvq_amd.models import VQA_Network
from kvq_amd.utils import synth
from oracle import swin3d_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCORE_TOL = 1e-3          # MOS units, north_star
FEAT_REL_L2 = 2e-2

KEY_FOR_CFG = {"SWIN_T_GRPB": "swin_tiny_grpb", "SWIN_T_PLAIN": "swin_tiny"}


def build_network(cfgn, wseed, scheme):
    key = KEY_FOR_CFG[cfgn]
    cfg = getattr(synth, cfgn)
    net = VQA_Network({"model": {"args": {key: {"backbone": {}, "head": {"in_channels": cfg.num_features,
                                                                           "hidden_channels": 64}}}}})
    sd = {f"{key}_backbone.{k}": torch.from_numpy(v) for k, v in synth.synth_swin_weights(cfg, wseed, scheme).items()}
    sd.update({f"{key}_head.{k}": torch.from_numpy(v)
               for k, v in synth.synth_vqa_head_weights(cfg.num_features, 64, wseed, scheme).items()})
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("relative_position_index" in k for k in missing.missing_keys)
    return net.to(DEV).eval(), key


CASES = ["t_grpb_stress_8x80", "t_grpb_stress_16x64", "t_plain_stress_16x96", "t_grpb_stress_10x50x70",
         "t_grpb_stress_32x224", "t_grpb_init_32x224"]


@pytest.mark.parametrize("case", CASES)
def test_trunk_and_score_vs_reference_golden(golden, case):
    g = golden("trunk.npz")
    wseed, cseed, B, T, H, W = (int(v) for v in g[f"{case}/meta"])
    cfgn, scheme = str(g[f"{case}/cfg"]), str(g[f"{case}/scheme"])
    net, key = build_network(cfgn, wseed, scheme)
    x = torch.from_numpy(synth.synth_clip(cseed, T, H, W, batch=B)).to(DEV)
    with torch.no_grad():
        score, feats = net(inputs={"technical": x}, reduce_scores=True, return_pooled_feats=True)
    feat = feats[key].cpu().numpy()
    assert tuple(g[f"{case}/feat/shape"]) == feat.shape
    flat = np.ascontiguousarray(feat).reshape(-1)
    ref_vals = g[f"{case}/feat/val"]
    got = flat[g[f"{case}/feat/idx"]]
    rel = np.linalg.norm(got - ref_vals) / np.linalg.norm(ref_vals)
    assert rel <= FEAT_REL_L2, rel
    assert score.shape == (B, 1)
    d = np.abs(score.cpu().numpy() - g[f"{case}/score"]).max()
    assert d <= SCORE_TOL, (d, score.cpu().numpy().ravel(), g[f"{case}/score"].ravel())


def test_full_size_vs_oracle_on_box():
    """Fresh seeds (not in the fixtures): oracle runs on this box's CPU, full 32x224x224, B=2."""
    cfg = synth.SWIN_T_GRPB
    net, key = build_network("SWIN_T_GRPB", 21, "stress")
    x = torch.from_numpy(synth.synth_clip(99, 32, 224, 224, batch=2))
    with torch.no_grad():
        score = net(inputs={"technical": x.to(DEV)}, reduce_scores=True).cpu()
        feat = O.swin3d_trunk(x, synth.synth_swin_weights(cfg, 21, "stress"), cfg)
        ref = O.vqa_head(feat, synth.synth_vqa_head_weights(768, 64, 21, "stress"))
    assert (score - ref).abs().max().item() <= SCORE_TOL, (score.ravel(), ref.ravel())


def test_batch_invariance_and_determinism():
    """A clip's score must not depend on what else is in the batch, nor vary run to run."""
    net, _ = build_network("SWIN_T_GRPB", 0, "stress")
    x = torch.from_numpy(synth.synth_clip(5, 32, 224, 224, batch=3)).to(DEV)
    with torch.no_grad():
        s3 = net(inputs={"technical": x}, reduce_scores=True).cpu()
        s3b = net(inputs={"technical": x}, reduce_scores=True).cpu()
        s1 = net(inputs={"technical": x[1:2].contiguous()}, reduce_scores=True).cpu()
    assert torch.equal(s3, s3b)
    assert torch.equal(s3[1:2], s1)


def test_forward_structure_matches_reference_api():
    """VQA_Network.forward return structure (models/model.py:105-121)."""
    net, key = build_network("SWIN_T_GRPB", 0, "init")
    x = torch.from_numpy(synth.synth_clip(1, 8, 64, 64, batch=1)).to(DEV)
    with torch.no_grad():
        as_list = net(inputs={"technical": x})
        reduced = net(inputs={"technical": x}, reduce_scores=True)
        s, f = net(inputs={"technical": x}, return_pooled_feats=True)
    assert isinstance(as_list, list) and len(as_list) == 1 and as_list[0].shape == (1, 1)
    assert torch.is_tensor(reduced) and reduced.shape == (1, 1)
    assert isinstance(s, list) and f[key].shape == (1, 768, 4, 2, 2)


def test_weights_follow_in_place_updates():
    """load_state_dict after a forward must invalidate the cached bf16 weights."""
    net, _ = build_network("SWIN_T_GRPB", 0, "stress")
    x = torch.from_numpy(synth.synth_clip(1, 8, 64, 64, batch=1)).to(DEV)
    with torch.no_grad():
        a = net(inputs={"technical": x}, reduce_scores=True).cpu()
        net2, _ = build_network("SWIN_T_GRPB", 7, "stress")
        net.load_state_dict(net2.state_dict())
        b = net(inputs={"technical": x}, reduce_scores=True).cpu()
        c = net2(inputs={"technical": x}, reduce_scores=True).cpu()
    assert not torch.equal(a, b) and torch.equal(b, c)


def test_no_cpu_fallback():
    net, _ = build_network("SWIN_T_GRPB", 0, "init")
    with pytest.raises(_abi.KvqError, match="no CPU path"):
        net(inputs={"technical": torch.zeros(1, 3, 8, 64, 64)})


def test_native_library_is_the_loaded_one():
    maps = open(f"/proc/{os.getpid()}/maps").read()
    assert "libkvq_hip.so" in maps
