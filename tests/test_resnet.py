"""SimpleVQA spatial branch (BASELINE config C1): oracle vs the reference's golden outputs on CPU; the
HIP conv path (im2col + MFMA GEMM + pooling kernels) vs the oracle / golden on the GPU."""
import numpy as np
import pytest
import torch

import kvq_amd  # noqa: F401
from kvq_amd.utils import synth
from oracle import resnet_oracle as RO
from oracle import swin3d_oracle as O

CASES = ["r50_2x96", "r50_b2_3x64x80"]


def _inputs(golden):
    g = golden("resnet.npz")
    rng = np.random.Generator(np.random.PCG64(55))
    out = {}
    for name in CASES:                      # same draw order as make_golden.sec_resnet
        B, T, H, W = (int(v) for v in g[f"{name}/meta"])
        frames = torch.from_numpy(rng.standard_normal((B, 3, T, H, W)).astype(np.float32))
        feat3d = torch.from_numpy(rng.standard_normal((B, T, 2304)).astype(np.float32))
        out[name] = (frames, feat3d, g[f"{name}/feat"], g[f"{name}/score"])
    return out


@pytest.mark.parametrize("case", CASES)
def test_resnet_oracle_matches_reference_golden(golden, case):
    frames, feat3d, f_ref, s_ref = _inputs(golden)[case]
    f = RO.simplevqa_features(frames, feat3d, synth.synth_resnet50_weights(4, "stress"))
    assert np.abs(f.numpy() - f_ref).max() <= 1e-5 * np.abs(f_ref).max()
    s = O.simple_vqa_head(f, synth.synth_simple_head_weights(9472, 128, 4, "stress"))
    assert np.abs(s.numpy() - s_ref).max() <= 1e-4


def test_resnet_state_dict_surface():
    from kvq_amd.models.backbones.simpleVQA_model import resnet50
    sd = resnet50().state_dict()
    shapes = synth.resnet50_param_shapes()
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("spec", [
    # B,C,D,H,W, kernel, stride, pad
    (2, 64, 1, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 64, 1, 15, 13, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    (1, 8, 6, 9, 9, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (1, 16, 8, 7, 7, (7, 1, 1), (4, 1, 1), (3, 0, 0)),
    (1, 256, 1, 8, 8, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
])
def test_im2col_nd_channels_last(spec, half):
    from kvq_amd import kernels
    B, C, D, H, W, k, s, p = spec
    g = np.random.Generator(np.random.PCG64(sum(spec[:5])))
    x = torch.from_numpy(g.standard_normal((B, D, H, W, C)).astype(np.float32)).to(half)
    cols, (Do, Ho, Wo) = kernels.im2col_nd(x.cuda(), (B, C, D, H, W), (D * H * W * C, 1, H * W * C, W * C, C), k, s, p, half)
    xp = torch.nn.functional.pad(x.float().permute(0, 4, 1, 2, 3), (p[2], p[2], p[1], p[1], p[0], p[0]))
    u = xp.unfold(2, k[0], s[0]).unfold(3, k[1], s[1]).unfold(4, k[2], s[2])       # B,C,Do,Ho,Wo,kd,kh,kw
    ref = u.permute(0, 2, 3, 4, 5, 6, 7, 1).reshape(B * Do * Ho * Wo, -1)          # (kd,kh,kw,c) columns
    K = ref.shape[1]
    assert cols.shape == (B * Do * Ho * Wo, -(-K // 32) * 32)
    assert torch.equal(cols[:, :K].float().cpu(), ref) and torch.all(cols[:, K:] == 0)


@gpu
def test_im2col_nd_fp32_stem_strides():
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(3))
    x = torch.from_numpy(g.standard_normal((1, 3, 4, 20, 22)).astype(np.float32))        # (b,c,T,h,w)
    T, h, w = 4, 20, 22
    cols, (_, Ho, Wo) = kernels.im2col_nd(x.cuda(), (T, 3, 1, h, w), (h * w, T * h * w, 0, w, 1), (1, 7, 7), (1, 2, 2),
                                          (0, 3, 3), torch.float16)
    fr = x[0].permute(1, 0, 2, 3)                                                         # frames (T,3,h,w)
    ref = torch.nn.functional.unfold(fr, 7, padding=3, stride=2)                           # (T, 3*49, L) cols (c,kh,kw)
    ref = ref.reshape(T, 3, 49, -1).permute(0, 3, 2, 1).reshape(T * Ho * Wo, 147)          # -> (kh,kw,c)
    assert torch.equal(cols[:, :147].float().cpu(), ref.half().float()) and cols.shape[1] == 160


@gpu
def test_pool_and_mean_std():
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(4))
    x = torch.from_numpy(g.standard_normal((2, 3, 13, 15, 40)).astype(np.float32)).half()
    mp = kernels.pool_nd(x.cuda(), (1, 3, 3), (1, 2, 2), (0, 1, 1), True).float().cpu()
    ref = torch.nn.functional.max_pool3d(x.float().permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2), (0, 1, 1)).permute(0, 2, 3, 4, 1)
    assert torch.equal(mp, ref)
    ap = kernels.pool_nd(x.cuda(), (3, 13, 15), (1, 1, 1), (0, 0, 0), False).float().cpu()
    ref = x.float().mean((1, 2, 3), keepdim=True)
    assert (ap - ref).abs().max().item() <= 2e-3
    y = x[:, 0].reshape(2, 13 * 15, 40)
    out = torch.zeros(2, 100, device="cuda")
    kernels.mean_std_pool(y.contiguous().cuda(), out, 5, 50)
    assert (out[:, 5:45].cpu() - y.float().mean(1)).abs().max().item() <= 1e-5
    assert (out[:, 50:90].cpu() - y.float().std(1)).abs().max().item() <= 1e-5     # unbiased, like torch.std


@gpu
@pytest.mark.parametrize("case", CASES)
def test_simplevqa_network_vs_reference_golden(golden, case):
    from kvq_amd.models import VQA_Network
    frames, feat3d, f_ref, s_ref = _inputs(golden)[case]
    net = VQA_Network({"model": {"args": {"simpleVQA": {"backbone": None, "head": {"in_channels": 9472,
                                                                                    "hidden_channels": 128}}}}})
    sd = {f"simpleVQA_backbone.{k}": torch.from_numpy(np.asarray(v)) for k, v in synth.synth_resnet50_weights(4, "stress").items()}
    sd.update({f"simpleVQA_head.{k}": torch.from_numpy(v) for k, v in synth.synth_simple_head_weights(9472, 128, 4, "stress").items()})
    net.load_state_dict(sd)
    net = net.cuda().eval()
    with torch.no_grad():
        score, feats = net(inputs={"simpleVQA": frames.cuda(), "feat": feat3d.cuda()}, reduce_scores=True,
                           return_pooled_feats=True)
    f = feats["simpleVQA"].cpu().numpy()
    assert f.shape == f_ref.shape
    rel = np.linalg.norm(f[..., :7168] - f_ref[..., :7168]) / np.linalg.norm(f_ref[..., :7168])
    assert rel <= 5e-3, rel                                  # 53 conv layers on fp16 operands, fp32 accumulate
    assert np.array_equal(f[..., 7168:], f_ref[..., 7168:])  # the SlowFast features pass through untouched
    assert np.abs(score.cpu().numpy() - s_ref).max() <= 1e-3, (score, s_ref)


@gpu
@pytest.mark.slow
def test_simplevqa_network_at_c1_size_vs_oracle():
    """BASELINE configs[0] at ITS OWN size (config/kwai_simpleVQA_test.yml: 8 frames of 448 x 448, 8 x 2304 SlowFast features): the
    HIP path through VQA_Network against the pinned CPU oracle run here (a few seconds) — score within 1e-3, pooled ResNet features
    within 5e-3 relative L2, the motion features untouched."""
    from kvq_amd.models import VQA_Network
    rng = np.random.Generator(np.random.PCG64(448))
    frames = torch.from_numpy(rng.standard_normal((1, 3, 8, 448, 448)).astype(np.float32))
    feat3d = torch.from_numpy(rng.standard_normal((1, 8, 2304)).astype(np.float32))
    rw, hw = synth.synth_resnet50_weights(4, "stress"), synth.synth_simple_head_weights(9472, 128, 4, "stress")
    f_ref = RO.simplevqa_features(frames, feat3d, rw)
    s_ref = O.simple_vqa_head(f_ref, hw)
    net = VQA_Network({"model": {"args": {"simpleVQA": {"backbone": None, "head": {"in_channels": 9472,
                                                                                    "hidden_channels": 128}}}}})
    sd = {f"simpleVQA_backbone.{k}": torch.from_numpy(np.asarray(v)) for k, v in rw.items()}
    sd.update({f"simpleVQA_head.{k}": torch.from_numpy(v) for k, v in hw.items()})
    net.load_state_dict(sd)
    net = net.cuda().eval()
    with torch.no_grad():
        score, feats = net(inputs={"simpleVQA": frames.cuda(), "feat": feat3d.cuda()}, reduce_scores=True, return_pooled_feats=True)
    f = feats["simpleVQA"].cpu()
    assert f.shape == f_ref.shape == (1, 8, 9472)
    rel = ((f[..., :7168] - f_ref[..., :7168]).norm() / f_ref[..., :7168].norm()).item()
    assert rel <= 5e-3, rel
    assert torch.equal(f[..., 7168:], f_ref[..., 7168:])
    assert (score.cpu() - s_ref).abs().max().item() <= 1e-3, (score, s_ref)


@gpu
@pytest.mark.parametrize("C", [40, 6])          # C % 8 == 0: the 8-channel-per-thread kernel; 6: the scalar one
def test_pool_vector_and_scalar_paths_agree_with_torch(C):
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(40 + C))
    x = torch.from_numpy(g.standard_normal((3, 2, 9, 11, C)).astype(np.float32)).half()
    mp = kernels.pool_nd(x.cuda(), (1, 3, 3), (1, 2, 2), (0, 1, 1), True).float().cpu()
    ref = torch.nn.functional.max_pool3d(x.float().permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2), (0, 1, 1)).permute(0, 2, 3, 4, 1)
    assert torch.equal(mp, ref)                                                      # max of fp16 values: exact
    ap = kernels.pool_nd(x.cuda(), (2, 3, 3), (1, 2, 2), (0, 0, 0), False).float().cpu()
    ref = torch.nn.functional.avg_pool3d(x.float().permute(0, 4, 1, 2, 3), (2, 3, 3), (1, 2, 2)).permute(0, 2, 3, 4, 1)
    assert torch.equal(ap, ref.half().float()) or (ap - ref).abs().max().item() <= 1e-3


@gpu
def test_pack_channels_last8_strided_frames():
    """(b, c, T, h, w) fp32 -> (b*T, h, w, 8) 16-bit, frame index = (b, t) through the strides, channels 3..7 zero: bit-exact."""
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(8))
    b, c, T, h, w = 2, 3, 3, 10, 13
    x = torch.from_numpy(g.standard_normal((b, c, T, h, w)).astype(np.float32))
    out = kernels.pack_channels_last8(x.cuda(), (b, T, c, h, w), (c * T * h * w, h * w, T * h * w, w, 1), torch.float16).cpu()
    ref = torch.zeros(b * T, h, w, 8, dtype=torch.float16)
    ref[..., :3] = x.permute(0, 2, 3, 4, 1).reshape(b * T, h, w, 3).half()
    assert torch.equal(out, ref)
    with pytest.raises(Exception, match="C=9"):
        kernels.pack_channels_last8(torch.zeros(1, 9, 1, 4, 4).cuda(), (1, 1, 9, 4, 4), (144, 0, 16, 4, 1), torch.float16)


@gpu
@pytest.mark.parametrize("hw,stride", [((1, 1), 1), ((2, 2), 2), ((2, 1), 1), ((3, 3), 1)])
def test_conv_implicit_with_pruned_taps_equals_full_conv(hw, stride):
    """Dropping the taps that only ever read the zero border (live_taps / prune_conv_weight) leaves the conv unchanged: same
    products, in the same order, minus exact zeros -> bit-identical to the full tap table whenever K stays slice-aligned,
    and equal to F.conv2d of the rounded operands within fp32 accumulation noise otherwise."""
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(sum(hw) + stride))
    n, c, cout = 70, 64, 32
    x = torch.from_numpy(g.standard_normal((n, 1) + hw + (c,)).astype(np.float32)).half()
    w5 = torch.from_numpy((g.standard_normal((cout, c, 3, 3)) / 24).astype(np.float32)).half()
    wk = w5.permute(0, 2, 3, 1).reshape(cout, 9 * c).contiguous()
    bias = torch.from_numpy(g.standard_normal(cout).astype(np.float32))
    live = kernels.live_taps((1,) + hw, (1, 3, 3), (1, stride, stride), (0, 1, 1))
    if hw == (1, 1):
        assert live == [[0], [1], [1]]
    if hw == (2, 2) and stride == 2:
        assert live == [[0], [1, 2], [1, 2]]
    if hw == (3, 3):
        assert live == [[0], [0, 1, 2], [0, 1, 2]]
    wp = kernels.prune_conv_weight(wk.cuda(), (1, 3, 3), c, live)
    assert wp.shape[1] == len(live[1]) * len(live[2]) * c
    full = kernels.conv_implicit(x.cuda(), wk.cuda(), bias.cuda(), (1, 3, 3), (1, stride, stride), (0, 1, 1), True)
    got = kernels.conv_implicit(x.cuda(), wp, bias.cuda(), (1, 3, 3), (1, stride, stride), (0, 1, 1), True, live=live)
    ref = torch.relu(torch.nn.functional.conv2d(x.float()[:, 0].permute(0, 3, 1, 2), w5.float(), bias, stride, 1)).permute(0, 2, 3, 1)
    assert got.shape == full.shape
    assert (got.float().cpu()[:, 0] - ref).abs().max().item() <= 4e-3
    assert (got.float() - full.float()).abs().max().item() <= 2e-3


@gpu
@pytest.mark.parametrize("rows,HW,C", [(1, 9408, 384), (2, 3136, 768), (3, 300, 40), (1, 1500, 20)])
def test_mean_std_pool_variants(rows, HW, C):
    """(mean, unbiased std) pooling on its three kernels (64-channel tile, 16-channel tile, 8-channel 16-byte tile) vs torch."""
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(rows + HW + C))
    y = torch.from_numpy((g.standard_normal((rows, HW, C)) * 2 + 0.5).astype(np.float32)).half()
    out = torch.zeros(rows, 2 * C + 3, device="cuda")
    kernels.mean_std_pool(y.cuda(), out, 1, C + 2)
    assert (out[:, 1:1 + C].cpu() - y.float().mean(1)).abs().max().item() <= 2e-5
    assert (out[:, C + 2:2 * C + 2].cpu() - y.float().std(1)).abs().max().item() <= 2e-5
    assert out[:, 0].abs().max().item() == 0 and out[:, C + 1].abs().max().item() == 0 and out[:, -1].abs().max().item() == 0
