"""SlowFast-R50 motion branch.  Parity is UNPINNED against the reference (pytorchvideo is absent,
SURVEY.md §8c): what is tested is (CPU) the wrapper semantics + architecture invariants of the oracle,
(GPU) HIP == oracle."""
import numpy as np
import pytest
import torch

import kvq_amd  # noqa: F401
from kvq_amd.utils import synth
from oracle import slowfast_oracle as SF


def test_pathway_packing_matches_reference_indices():
    x = torch.arange(32, dtype=torch.float32).reshape(1, 1, 32, 1, 1).expand(1, 3, 32, 1, 1)
    slow, fast = SF.pack_pathway_output(x)
    # reference: linspace(0, 31, 8).long() = [0,4,8,13,17,22,26,31]  
    assert slow[0, 0, :, 0, 0].tolist() == [0, 4, 8, 13, 17, 22, 26, 31] and fast.shape[2] == 32
    from kvq_amd.models.backbones.slowfast_model import pack_pathway_output
    s2, f2 = pack_pathway_output(x)
    assert torch.equal(s2, slow) and torch.equal(f2, fast)


def test_architecture_invariants():
    sh = SF.param_shapes()
    n = sum(int(np.prod(v)) for k, v in sh.items() if "running" not in k and "num_batches" not in k)
    # SlowFast-R50 8x8: 34.57 M parameters incl. the 2304->400 classifier (0.92 M) the reference drops
    assert abs(n + 2304 * 400 + 400 - 34.57e6) < 0.02e6, n
    from kvq_amd.models.backbones.slowfast_model import slowfast
    sd = slowfast().state_dict()
    assert set(sd) == set(sh) and all(tuple(sd[k].shape) == tuple(sh[k]) for k in sh)
    w = synth.synth_params(sh, 3, "stress", prefix="sf.")
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).standard_normal((1, 3, 8, 32, 32)).astype(np.float32))
    with torch.no_grad():
        s, f = SF.slowfast_features(x, w)
    assert s.shape == (1, 2048, 1, 1, 1) and f.shape == (1, 256, 1, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 16, 64, 64), (2, 3, 8, 96, 64)])
def test_hip_slowfast_matches_oracle(shape):
    from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(sum(shape))).standard_normal(shape).astype(np.float32))
    m = slowfast()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    m = m.cuda().eval()
    with torch.no_grad():
        s_ref, f_ref = SF.slowfast_features(x, w)
        s, f = m(pack_pathway_output(x.cuda()))
    assert s.shape == s_ref.shape and f.shape == f_ref.shape
    for got, ref in ((s, s_ref), (f, f_ref)):
        rel = ((got.cpu() - ref).norm() / ref.norm()).item()
        assert rel <= 5e-3, rel          # ~100 conv layers on fp16 operands, fp32 accumulate / residual stream


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kernel,stride,pad,cout", [
    ((2, 3, 6, 30, 46), (5, 7, 7), (1, 2, 2), (2, 3, 3), 8),      # the fast-pathway stem geometry, Wo = 23
    ((1, 3, 4, 16, 32), (1, 7, 7), (1, 2, 2), (0, 3, 3), 8),
    ((1, 3, 4, 16, 20), (3, 5, 5), (1, 1, 1), (1, 2, 2), 8),      # other kernel width
    ((1, 3, 3, 12, 18), (1, 7, 7), (1, 2, 2), (0, 3, 3), 16),
])
def test_conv_stem_direct_vs_torch_conv3d(shape, kernel, stride, pad, cout):
    """kvq_conv_stem_direct (fp32 arithmetic, 16-bit channels-last output) against F.conv3d + bias + ReLU on the same fp32 data."""
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(sum(shape) + cout))
    x = torch.from_numpy(g.standard_normal(shape).astype(np.float32))
    K = kernel[0] * kernel[1] * kernel[2] * shape[1]
    w5 = torch.from_numpy((g.standard_normal((cout, shape[1]) + kernel) / np.sqrt(K)).astype(np.float32))
    bias = torch.from_numpy(g.standard_normal(cout).astype(np.float32))
    w_kc = w5.permute(2, 3, 4, 1, 0).reshape(K, cout).contiguous()                      # [K][Cout], K ordered (kd, kh, kw, c)
    out = kernels.conv_stem_direct(x.cuda(), w_kc.cuda(), bias.cuda(), kernel, stride, pad, True, torch.float16).float().cpu()
    ref = torch.relu(torch.nn.functional.conv3d(x.double(), w5.double(), bias.double(), stride, pad)).permute(0, 2, 3, 4, 1).float()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 2.0 ** -10 * max(1.0, ref.abs().max().item())      # fp32 accumulate, one fp16 rounding


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,kernel,stride,pad", [
    ((2, 3, 6, 30, 46), (5, 7, 7), (1, 2, 2), (2, 3, 3)),      # fast-pathway stem geometry; Wo = 23: ragged last tile
    ((1, 3, 4, 18, 250), (1, 7, 7), (1, 2, 2), (0, 3, 3)),     # Wo = 125 > 112: second pass of the 7-tile loop
    ((1, 2, 3, 9, 16), (3, 3, 7), (2, 1, 2), (1, 1, 3)),       # two channels, other kd / kh / strides
])
def test_conv_stem_mfma_vs_torch_conv3d(shape, kernel, stride, pad, dtype):
    """kvq_pack_clip_cl4 + kvq_conv_stem_mfma (16-bit operands, fp32 accumulate) against F.conv3d of the ROUNDED operands + bias +
    ReLU; and against the fp32 direct kernel within the operand rounding."""
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(sum(shape)))
    cin, cout = shape[1], 8
    x = torch.from_numpy(g.standard_normal(shape).astype(np.float32))
    K = kernel[0] * kernel[1] * kernel[2] * cin
    w5 = torch.from_numpy((g.standard_normal((cout, cin) + kernel) / np.sqrt(K)).astype(np.float32))
    bias = torch.from_numpy(g.standard_normal(cout).astype(np.float32))
    w_kc = w5.permute(2, 3, 4, 1, 0).reshape(K, cout).contiguous()
    wp = kernels.stem_mfma_pack_weight(w_kc.cuda(), kernel, cin, dtype)
    out = kernels.conv_stem_mfma(x.cuda(), wp, bias.cuda(), kernel, stride, pad, True).float().cpu()
    xr, wr = x.to(dtype).double(), w5.to(dtype).double()
    ref = torch.relu(torch.nn.functional.conv3d(xr, wr, bias.double(), stride, pad)).permute(0, 2, 3, 4, 1).float()
    assert out.shape == ref.shape
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert (out - ref).abs().max().item() <= ulp * max(1.0, ref.abs().max().item())
    if cin == 3:
        direct = kernels.conv_stem_direct(x.cuda(), w_kc.cuda(), bias.cuda(), kernel, stride, pad, True, dtype).float().cpu()
        assert (out - direct).abs().max().item() <= (3e-2 if dtype == torch.float16 else 2e-1)
