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


def test_weight_images_of_the_fused_stems_and_blocks_on_the_host():
    """The packed weight images handed to the fused SlowFast launches, checked entry by entry on the host (no GPU): the slow stem's
    [7][64][32] image (``kvq_conv_stem64_pool``), the fast stem's [kd*7][16][32] image (``kvq_conv_stem_pool`` / ``kvq_conv_stem_mfma``),
    and the MFMA A-fragment layout + bias block of the slow pathway's fused res2 block (``kvq_slow_bottleneck``): fragment f of row tile
    t holds, for lane (m = lane & 31, h = lane >> 5), the eight weights W[32 t + m][k(f, h, e)]."""
    from kvq_amd import kernels, _abi
    from kvq_amd.models.backbones.slowfast_model import pack_slow_bottleneck
    g = torch.Generator().manual_seed(5)
    w5 = torch.randn(64, 3, 1, 7, 7, generator=g)
    img = kernels.stem64_pack_weight(w5.permute(0, 2, 3, 4, 1).reshape(64, 147).contiguous(), torch.float16).float()
    assert tuple(img.shape) == (7, 64, 32)
    for kh, o, kw, c in ((0, 0, 0, 0), (3, 17, 6, 2), (6, 63, 4, 1)):
        assert img[kh, o, kw * 4 + c] == w5[o, c, 0, kh, kw].half().float()
    assert (img.reshape(7, 64, 8, 4)[:, :, 7, :] == 0).all() and (img.reshape(7, 64, 8, 4)[:, :, :, 3] == 0).all()
    wf = torch.randn(8, 3, 5, 7, 7, generator=g)
    imf = kernels.stem_mfma_pack_weight(wf.permute(2, 3, 4, 1, 0).reshape(735, 8).contiguous(), (5, 7, 7), 3, torch.float16).float()
    assert tuple(imf.shape) == (35, 16, 32) and (imf[:, 8:] == 0).all()
    for a, r, o, kw, c in ((0, 0, 0, 0, 0), (4, 6, 7, 6, 2), (2, 3, 5, 1, 1)):
        assert imf[a * 7 + r, o, kw * 4 + c] == wf[o, c, a, r, kw].half().float()
    wa, wb, wc = torch.randn(64, 256, generator=g).half(), torch.randn(64, 576, generator=g).half(), torch.randn(256, 64, generator=g).half()
    ba, bb, bc = torch.randn(64, generator=g), torch.randn(64, generator=g), torch.randn(256, generator=g)
    blob = pack_slow_bottleneck(wa, ba, wb, bb, wc, bc)
    assert blob.numel() == _abi.lib().kvq_slow_bottleneck_pack_bytes(256, 64, 256) == (32 + 72 + 32 + 2) * 1024
    assert _abi.lib().kvq_slow_bottleneck_pack_bytes(512, 128, 512) == 0
    frag = blob[: (32 + 72 + 32) * 1024].view(torch.float16).reshape(-1, 64, 8).float()          # [fragment][lane][e]

    def entry(base, ks, t, f, lane, e, acc_order=False):
        m, h = lane & 31, lane >> 5
        k = 16 * f + (8 * (e >> 2) + 4 * h + (e & 3) if acc_order else 8 * h + e)
        return frag[base + t * ks + f, lane, e], 32 * t + m, k
    for t, f, lane, e in ((0, 0, 0, 0), (1, 15, 63, 7), (1, 7, 37, 3)):
        v, row, k = entry(0, 16, t, f, lane, e)
        assert v == wa[row, k].float()
    for t, f, lane, e in ((0, 0, 1, 1), (1, 35, 40, 6)):
        v, row, k = entry(32, 36, t, f, lane, e)
        assert v == wb[row, k].float()
    for t, f, lane, e in ((0, 0, 2, 5), (7, 3, 62, 2)):
        v, row, k = entry(32 + 72, 4, t, f, lane, e, acc_order=True)
        assert v == wc[row, k].float()
    bias = blob[(32 + 72 + 32) * 1024:].view(torch.float32)
    assert torch.equal(bias[:64], ba) and torch.equal(bias[64:128], bb) and torch.equal(bias[128:384], bc) and (bias[384:] == 0).all()


def test_clip_assembly_vs_reference_golden(golden):
    """``clip_frame_indices`` against the frame indices the reference's own ``VideoDataset_NR_SlowFast_feature.__getitem__``
    (SlowFast_features.py:52-107) produced over a fake capture (tests/golden/make_golden.py::sec_sfclips): per-second 32-frame
    clips, last-frame padding, minimum 8 clips, undelivered frames, fps 0, and the too-short-video IndexError."""
    from kvq_amd.datasets.slowfast_clips import clip_frame_indices
    g = golden("sfclips.npz")
    assert len(g["cases"]) >= 14
    for name in g["cases"]:
        length, rate, readable = (int(v) for v in g[f"{name}/meta"])
        if f"{name}/error" in g:
            with pytest.raises(IndexError):
                clip_frame_indices(length, rate, readable)
            continue
        got = np.stack(clip_frame_indices(length, rate, readable))
        assert np.array_equal(got, g[f"{name}/idx"]), name
        assert got.shape[0] >= 8 and got.shape[1] == 32


def test_pil_transform_and_dataset_over_a_frame_stack(tmp_path):
    """The reference's frame transform (PIL Resize([r, r]) -> ToTensor -> Normalize(.45, .225), SlowFast_features.py:172-173)
    and the dataset over a ``.npy`` frame stack + ``.fps`` side file: clip count, padding, values."""
    import types
    from PIL import Image
    from kvq_amd.datasets.slowfast_clips import VideoDataset_NR_SlowFast_feature, pil_transform
    g = np.random.Generator(np.random.PCG64(5))
    frames = g.integers(0, 256, size=(70, 36, 48, 3), dtype=np.uint8)
    np.save(str(tmp_path / "a.mp4.npy"), frames)
    (tmp_path / "a.mp4.fps").write_text("29.97")
    (tmp_path / "v.csv").write_text("filename,score\na.mp4,3.5\n")
    tr = pil_transform(16)
    ref = (torch.from_numpy(np.array(Image.fromarray(frames[3]).resize((16, 16), Image.BILINEAR))).permute(2, 0, 1).float() / 255 - 0.45) / 0.225
    assert torch.equal(tr(frames[3]), ref)
    ds = VideoDataset_NR_SlowFast_feature(types.SimpleNamespace(resize=16, fps=None), None, str(tmp_path), str(tmp_path / "v.csv"))
    clips, name = ds[0]
    assert name == "a.mp4" and len(ds) == 1 and len(clips) == 8 and clips[0].shape == (32, 3, 16, 16)
    assert torch.equal(clips[0][5], tr(frames[5])) and torch.equal(clips[1][0], tr(frames[30]))
    assert torch.equal(clips[1][31], tr(frames[61])) and all(torch.equal(clips[k], clips[1]) for k in range(2, 8))


@pytest.mark.gpu
def test_hip_slowfast_full_clip_matches_oracle():
    """One 32 x 224 x 224 clip — the size BASELINE config 3 runs — HIP vs the CPU restatement (parity UNPINNED vs pytorchvideo)."""
    from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    x = torch.from_numpy(synth.synth_clip(31, 32, 224, 224, batch=1))
    m = slowfast()
    m.head_small_grid = "mean"          # reduced-size clip: final grid under the head's (8,7,7) kernel
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    m = m.cuda().eval()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        s_ref, f_ref = SF.slowfast_features(x, w)
        s, f = m(pack_pathway_output(x.cuda()))
    assert s.shape == s_ref.shape == (1, 2048, 1, 1, 1) and f.shape == f_ref.shape == (1, 256, 1, 1, 1)
    for got, ref in ((s, s_ref), (f, f_ref)):
        rel = ((got.cpu() - ref).norm() / ref.norm()).item()
        assert rel <= 5e-3, rel


@pytest.mark.gpu
def test_one_call_network_equals_layer_by_layer_sequencing():
    """``kvq_convnet_forward`` (the layer table handed to C once, one call per forward: frame select on the device, lateral
    concatenations written in place at a channel offset, slots recycled in the workspace) against the same kernels sequenced layer by
    layer from Python with torch.cat / index_select in between: the same values, bit for bit; and the plan's error paths."""
    import ctypes as C
    import kvq_amd.models.backbones.slowfast_model as M
    from kvq_amd import _abi
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    m = M.slowfast()
    m.head_small_grid = "mean"          # reduced-size clip: final grid under the head's (8,7,7) kernel
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    m = m.cuda().eval()
    x = torch.from_numpy(synth.synth_clip(77, 16, 96, 64, batch=3)).cuda()
    M.FUSE_FAST = M.FUSE_SLOW = M.STEM_POOL = False          # the layer-by-layer sequencing has one launch per conv / pool: compare like with like
    try:
        with torch.no_grad():
            s1, f1 = m(M.pack_pathway_output(x))
            s1b, f1b = m(M.pack_pathway_output(x))
            M.CONVNET = False
            try:
                s0, f0 = m(M.pack_pathway_output(x))
            finally:
                M.CONVNET = True
    finally:
        M.FUSE_FAST = M.FUSE_SLOW = M.STEM_POOL = True
    assert m.__dict__.get("_nets"), "the one-call path did not run"
    assert torch.equal(s1, s1b) and torch.equal(f1, f1b)
    assert torch.equal(s1, s0) and torch.equal(f1, f0)
    # the shipped plan (stems fused with their max-pools, fast blocks fused): same network, other accumulation order inside the stems
    for handle, *_ in m._nets.values():
        _abi.lib().kvq_convnet_destroy(handle)
    m._nets.clear()
    with torch.no_grad():
        s2, f2 = m(M.pack_pathway_output(x))
    for a, b_ in ((s2, s0), (f2, f0)):
        assert (a.float() - b_.float()).abs().max().item() <= 2e-2 * max(1.0, b_.float().abs().max().item())
    # a layer table whose shapes do not chain is rejected at plan creation, with a message
    t = (_abi.KvqNetTensor * 2)()
    t[0].B, t[0].D, t[0].H, t[0].W, t[0].C, t[0].kind = 1, 4, 8, 8, 16, _abi.NET_T_ACT16
    t[1].B, t[1].D, t[1].H, t[1].W, t[1].C, t[1].kind = 1, 4, 8, 8, 16, _abi.NET_T_ACT16
    o = (_abi.KvqNetOp * 1)()
    o[0].kind, o[0].src, o[0].dst, o[0].src2, o[0].dst32 = _abi.NET_POOL, 0, 1, -1, -1
    o[0].kernel3[:], o[0].stride3[:], o[0].pad3[:] = (1, 3, 3), (1, 2, 2), (0, 1, 1)          # -> 4 x 4 x 4, not 4 x 8 x 8
    h = C.c_void_p()
    rc = _abi.lib().kvq_convnet_create(o, 1, t, 2, 1, 0, _abi.DT_FP16, C.byref(h))
    assert rc != 0 and b"pool" in _abi.lib().kvq_last_error()


def _ref_fast_block(x, wa, ba, wb, bb, wc, bc, ws, bs, half, stride=1):
    """fp32 torch restatement of one fast-pathway residual block with the 16-bit rounding points of the HIP path"""
    import torch.nn.functional as F
    xf = x.float().permute(0, 4, 1, 2, 3)
    ci, cin, cout = wa.shape[0], x.shape[-1], wc.shape[0]
    a = F.relu(F.conv3d(xf, wa.float()[:, :3 * cin].reshape(ci, 3, 1, 1, cin).permute(0, 4, 1, 2, 3), ba, padding=(1, 0, 0))).to(half).float()
    b = F.relu(F.conv3d(a, wb.float()[:, :9 * ci].reshape(ci, 1, 3, 3, ci).permute(0, 4, 1, 2, 3), bb, stride=(1, stride, stride),
                        padding=(0, 1, 1))).to(half).float()
    c = F.conv3d(b, wc.float()[:, :ci].reshape(cout, ci, 1, 1, 1), bc)
    sc = F.conv3d(xf, ws.float()[:, :cin].reshape(cout, cin, 1, 1, 1), bs, stride=(1, stride, stride)) if ws is not None else xf
    return F.relu(c + sc).permute(0, 2, 3, 4, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,ci,cout,proj,stride", [(8, 8, 32, True, 1), (32, 8, 32, False, 1), (64, 16, 64, False, 1), (128, 32, 128, False, 1),
                                                     (32, 16, 64, True, 2), (64, 32, 128, True, 2)])
@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16])
def test_fast_bottleneck_kernel_vs_fp32_convs(cin, ci, cout, proj, stride, half):
    """``kvq_fast_bottleneck`` (conv_a 3x1x1 -> conv_b 1x3x3 -> conv_c 1x1x1 + shortcut, one launch) against torch conv3d in fp32
    with the same 16-bit rounding of the two inner activations: maps that are not multiples of the tile (odd sizes under stride 2),
    temporal and spatial borders, every built block (stride-2 first blocks with their strided projection included); unsupported
    blocks are refused."""
    from kvq_amd import kernels
    from kvq_amd.models.backbones.slowfast_model import pack_fast_bottleneck
    g = torch.Generator().manual_seed(cin * 1000 + ci)
    B, T, H, W = 2, 5, 17, 30
    x = torch.randn(B, T, H, W, cin, generator=g).to(half)

    def wgt(rows, k):
        kpad = -(-k // 32) * 32
        w = torch.zeros(rows, kpad)
        w[:, :k] = torch.randn(rows, k, generator=g) * (2.0 / k) ** 0.5
        return w.to(half)
    wa, wb, wc = wgt(ci, 3 * cin), wgt(ci, 9 * ci), wgt(cout, ci)
    ba, bb, bc = (torch.randn(n, generator=g) * 0.2 for n in (ci, ci, cout))
    ws, bs = (wgt(cout, cin), torch.randn(cout, generator=g) * 0.2) if proj else (None, None)
    ref = _ref_fast_block(x, wa, ba, wb, bb, wc, bc, ws, bs, half, stride)
    dev = lambda t: None if t is None else t.cuda()      # noqa: E731
    pack = pack_fast_bottleneck(dev(wa), dev(ba), dev(wb), dev(bb), dev(wc), dev(bc), cin, dev(ws), dev(bs), stride=stride)
    got = kernels.fast_bottleneck(x.cuda(), pack, ci, cout, proj, stride).float().cpu()
    assert got.shape == ref.shape
    tol = 2e-2 if half == torch.bfloat16 else 3e-3
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    assert (got - ref).norm().item() / ref.norm().item() < tol / 4
    with pytest.raises(AssertionError):
        kernels.fast_bottleneck(x.cuda(), pack, ci + 8, cout, proj, stride)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("dims,out_c", [((2, 3, 17, 30), 256), ((1, 2, 56, 56), 320), ((1, 1, 14, 14), 256)])
def test_slow_bottleneck_kernel_vs_fp32_convs(dims, out_c, half):
    """``kvq_slow_bottleneck`` (conv_a 1x1x1 -> conv_b 1x3x3 -> conv_c 1x1x1 + identity, one launch; 256 -> 64 -> 256) against torch
    conv3d in fp32 with the same 16-bit rounding of the two inner activations: maps that are not multiples of the 14 x 14 tile,
    spatial borders, an output row wider than the block's channels (the stage's last block writes into the lateral-concat tensor: the
    other channels stay untouched); other channel counts are refused."""
    import torch.nn.functional as F
    from kvq_amd import kernels
    from kvq_amd.models.backbones.slowfast_model import pack_slow_bottleneck
    cin, ci, cout = 256, 64, 256
    g = torch.Generator().manual_seed(sum(dims) + out_c)
    B, T, H, W = dims
    x = torch.randn(B, T, H, W, cin, generator=g).to(half)

    def wgt(rows, k):
        return (torch.randn(rows, k, generator=g) * (2.0 / k) ** 0.5).to(half)
    wa, wb, wc = wgt(ci, cin), wgt(ci, 9 * ci), wgt(cout, ci)
    ba, bb, bc = (torch.randn(n, generator=g) * 0.2 for n in (ci, ci, cout))
    xf = x.float().permute(0, 4, 1, 2, 3)
    a = F.relu(F.conv3d(xf, wa.float().reshape(ci, cin, 1, 1, 1), ba)).to(half).float()
    b = F.relu(F.conv3d(a, wb.float().reshape(ci, 1, 3, 3, ci).permute(0, 4, 1, 2, 3), bb, padding=(0, 1, 1))).to(half).float()
    ref = F.relu(F.conv3d(b, wc.float().reshape(cout, ci, 1, 1, 1), bc) + xf).permute(0, 2, 3, 4, 1)
    pack = pack_slow_bottleneck(wa.cuda(), ba.cuda(), wb.cuda(), bb.cuda(), wc.cuda(), bc.cuda())
    out = torch.full((B, T, H, W, out_c), 3.0, dtype=half, device="cuda")
    got = kernels.slow_bottleneck(x.cuda(), pack, ci, cout, out=out)
    assert (got[..., cout:] == 3.0).all()
    got = got[..., :cout].float().cpu()
    tol = 2e-2 if half == torch.bfloat16 else 3e-3
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    assert (got - ref).norm().item() / ref.norm().item() < tol / 4
    with pytest.raises(AssertionError):
        kernels.slow_bottleneck(x.cuda(), pack, ci + 8, cout)


@pytest.mark.gpu
def test_fused_fast_pathway_blocks_match_the_conv_by_conv_plan():
    """The network with the fast pathway's residual blocks and the slow pathway's res2 identity blocks fused (default) against the same plan with one launch per conv: the
    same rounding points, so the pooled features agree to accumulation-order noise; the fused plan is the one the oracle parity
    tests above exercise."""
    import kvq_amd.models.backbones.slowfast_model as M
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}
    x = torch.from_numpy(synth.synth_clip(91, 16, 96, 64, batch=2)).cuda()
    outs = []
    for fuse in (True, False):
        M.FUSE_FAST = M.FUSE_SLOW = fuse
        try:
            m = M.slowfast()
            m.head_small_grid = "mean"          # reduced-size clip: final grid under the head's (8,7,7) kernel
            m.load_state_dict(sd)
            m = m.cuda().eval()
            with torch.no_grad():
                outs.append(m.forward_clips(x))
            rows = m.profile_layers(x)
            assert any(r["kind"] == "bottleneck" for r in rows) == fuse
            assert any("0.res_blocks.1 (fused" in r["name"] for r in rows) == fuse          # a slow-pathway res2 identity block
        finally:
            M.FUSE_FAST = M.FUSE_SLOW = True
    for a, b in zip(*outs):
        assert (a - b).norm().item() / b.norm().item() < 1e-3


@pytest.mark.gpu
def test_two_lane_plan_equals_one_lane_and_profile_reads_every_op():
    """The fast pathway on the plan's second stream (KvqNetOp.lane = 1: fork / event-ordered / join inside one forward) against
    the same plan on the caller's stream alone: bit-identical features, also when forwards follow each other without a host
    synchronise and from a side stream; ``forward_clips`` (frame selection on the device) == ``forward(pack_pathway_output)``;
    ``kvq_convnet_profile`` returns one positive time per op and leaves the values unchanged."""
    import kvq_amd.models.backbones.slowfast_model as M
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}
    nets = []
    for lanes in (False, True):
        m = M.slowfast(two_lanes=lanes)
        m.head_small_grid = "mean"          # reduced-size clip: final grid under the head's (8,7,7) kernel
        m.load_state_dict(sd)
        nets.append(m.cuda().eval())
    xs = [torch.from_numpy(synth.synth_clip(70 + i, 16, 96, 64, batch=2)).cuda() for i in range(3)]
    with torch.no_grad():
        ref = [nets[0].forward_clips(x) for x in xs]
        got = [nets[1].forward_clips(x) for x in xs]                # back to back: the join of forward i orders forward i+1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got_side = [nets[1].forward_clips(x) for x in xs]
        torch.cuda.synchronize()
        packed = nets[1](M.pack_pathway_output(xs[0]))
        rows = nets[1].profile_layers(xs[0])
        after = nets[1].forward_clips(xs[0])
    for (s0, f0), (s1, f1), (s2, f2) in zip(ref, got, got_side):
        assert torch.equal(s0, s1) and torch.equal(f0, f1)
        assert torch.equal(s0, s2) and torch.equal(f0, f2)
    assert torch.equal(packed[0], ref[0][0]) and torch.equal(packed[1], ref[0][1])
    assert torch.equal(after[0], ref[0][0]) and torch.equal(after[1], ref[0][1])
    convs = [r for r in rows if r["kind"] == "conv"]
    fused = [r for r in rows if r["kind"] == "bottleneck"]
    n_proj = sum(1 for r in fused if r["name"].split(" ")[0].endswith("res_blocks.0"))
    assert all(r["ms"] > 0 for r in rows) and len(convs) + 3 * len(fused) + n_proj == len(M.conv_table()) - 2   # every conv but the two stems
    assert any(r["name"].endswith("multipathway_fusion") for r in convs)


@pytest.mark.gpu
def test_cli_extracts_features_from_a_video_tree(tmp_path):
    """``python SlowFast_features.py --video_root --video_csv --database --feature_save_folder`` (reference CLI, :200-217) over
    two ``.npy`` frame stacks: the on-disk layout the SimpleVQA dataset reads, clip count incl. the 8-clip minimum, and the
    saved features == the model called directly on the assembled clips."""
    import os
    import subprocess
    import sys
    import types
    from kvq_amd.datasets.slowfast_clips import VideoDataset_NR_SlowFast_feature, extract_video
    from kvq_amd.models.backbones.slowfast_model import slowfast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.random.Generator(np.random.PCG64(9))
    for name, n in (("a.mp4", 100), ("b.mp4", 45)):
        np.save(str(tmp_path / f"{name}.npy"), g.integers(0, 256, size=(n, 60, 80, 3), dtype=np.uint8))
    (tmp_path / "b.mp4.fps").write_text("15")
    (tmp_path / "v.csv").write_text("filename,score\na.mp4,1\nb.mp4,2\n")
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}, str(tmp_path / "w.pth"))
    r = subprocess.run([sys.executable, os.path.join(root, "SlowFast_features.py"), "--video_root", str(tmp_path), "--video_csv",
                        str(tmp_path / "v.csv"), "--database", "kvq", "--feature_save_folder", str(tmp_path / "feat"), "--resize", "64",
                        "--num_workers", "0", "--fps", "30", "--weights", str(tmp_path / "w.pth"), "--small_grid_mean"], capture_output=True,
                       text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    m = slowfast()
    m.head_small_grid = "mean"          # reduced-size clip: final grid under the head's (8,7,7) kernel
    m.load_state_dict(torch.load(str(tmp_path / "w.pth")))
    m = m.cuda().eval()
    ds = VideoDataset_NR_SlowFast_feature(types.SimpleNamespace(resize=64, fps=30.0), None, str(tmp_path), str(tmp_path / "v.csv"))
    for vi, (name, nclip) in enumerate((("a.mp4", 8), ("b.mp4", 8))):
        d = tmp_path / "feat" / "kvq" / name
        files = sorted(os.listdir(d))
        assert len(files) == 2 * nclip, files
        clips, vname = ds[vi]
        assert vname == name and len(clips) == nclip
        direct = extract_video(m, clips, "cuda")
        for i in range(nclip):
            slow = np.load(str(d / f"feature_{i}_slow_feature.npy"))
            fast = np.load(str(d / f"feature_{i}_fast_feature.npy"))
            assert slow.shape == (1, 2048, 1, 1, 1) and fast.shape == (1, 256, 1, 1, 1)
            assert np.allclose(slow, direct[i][0], rtol=0, atol=2e-3 * np.abs(direct[i][0]).max())
            assert np.allclose(fast, direct[i][1], rtol=0, atol=2e-3 * np.abs(direct[i][1]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 16, 64, 64), (2, 3, 8, 96, 64)])
def test_hip_slowfast_matches_oracle(shape):
    from kvq_amd.models.backbones.slowfast_model import pack_pathway_output, slowfast
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(sum(shape))).standard_normal(shape).astype(np.float32))
    m = slowfast()
    m.head_small_grid = "mean"          # reduced-size clip: final grid under the head's (8,7,7) kernel
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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,kd", [
    ((2, 3, 6, 30, 48), 5),        # Wo = 24: ragged second tile; Hp = 8: two row blocks, zero rows above and below
    ((1, 3, 3, 50, 252), 5),       # Wo = 126: eight column tiles (two per wave); Hp = 13: ragged last row block; T < kd
    ((1, 3, 2, 224, 224), 5),      # the clip geometry
    ((1, 3, 4, 21, 20), 1),        # odd H, one temporal tap
])
def test_conv_stem_pool_vs_torch(shape, kd, dtype):
    """kvq_conv_stem_pool (conv + bias + ReLU + 3x3/2 max-pool in one launch, straight from the fp32 clip) against F.conv3d of the
    ROUNDED operands -> ReLU -> max_pool3d, and within one 16-bit rounding of the three-launch path it replaces (pack +
    kvq_conv_stem_mfma + pool: same operands, fp32 accumulation in another order)."""
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(sum(shape) + kd))
    kernel, stride, pad = (kd, 7, 7), (1, 2, 2), (kd // 2, 3, 3)
    x = torch.from_numpy(g.standard_normal(shape).astype(np.float32))
    K = kd * 49 * 3
    w5 = torch.from_numpy((g.standard_normal((8, 3) + kernel) / np.sqrt(K)).astype(np.float32))
    bias = torch.from_numpy(g.standard_normal(8).astype(np.float32))
    w_kc = w5.permute(2, 3, 4, 1, 0).reshape(K, 8).contiguous()
    wp = kernels.stem_mfma_pack_weight(w_kc.cuda(), kernel, 3, dtype)
    out = kernels.conv_stem_pool(x.cuda(), wp, bias.cuda(), kd, True)
    xr, wr = x.to(dtype).double(), w5.to(dtype).double()
    ref = torch.relu(torch.nn.functional.conv3d(xr, wr, bias.double(), stride, pad))
    ref = torch.nn.functional.max_pool3d(ref, (1, 3, 3), (1, 2, 2), (0, 1, 1)).permute(0, 2, 3, 4, 1).float()
    assert tuple(out.shape) == tuple(ref.shape)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert (out.float().cpu() - ref).abs().max().item() <= ulp * max(1.0, ref.abs().max().item())
    stem = kernels.conv_stem_mfma(x.cuda(), wp, bias.cuda(), kernel, stride, pad, True)
    three = torch.nn.functional.max_pool3d(stem.float().permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2), (0, 1, 1)).permute(0, 2, 3, 4, 1)
    close = (out.float() - three).abs().max().item()
    assert close <= ulp * max(1.0, ref.abs().max().item()), close


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,t_index,coff", [
    ((2, 3, 8, 30, 48), [0, 2, 5, 7], 0),        # Wo = 24 (ragged second tile), Hp = 8
    ((1, 3, 4, 50, 220), [3, 0], 16),            # Wo = 110: seven tiles, the last ragged; Hp = 13: ragged last row block; channel slice
    ((1, 3, 8, 224, 224), [0, 7], 0),            # the clip geometry
    ((1, 3, 2, 21, 20), None, 0),                # odd H, every frame
])
def test_conv_stem64_pool_vs_torch(shape, t_index, coff, dtype):
    """kvq_conv_stem64_pool (frame selection + 1x7x7 conv + bias + ReLU + 3x3/2 max-pool in one launch from the fp32 clip) against
    index_select -> F.conv3d of the ROUNDED operands -> ReLU -> max_pool3d; a channel slice leaves the other channels alone."""
    from kvq_amd import kernels
    g = np.random.Generator(np.random.PCG64(sum(shape)))
    x = torch.from_numpy(g.standard_normal(shape).astype(np.float32))
    w5 = torch.from_numpy((g.standard_normal((64, 3, 1, 7, 7)) / np.sqrt(147)).astype(np.float32))
    bias = torch.from_numpy(g.standard_normal(64).astype(np.float32))
    w_ok = w5.permute(0, 2, 3, 4, 1).reshape(64, 147).contiguous()
    wimg = kernels.stem64_pack_weight(w_ok.cuda(), dtype)
    xs = x if t_index is None else x.index_select(2, torch.tensor(t_index))
    ref = torch.relu(torch.nn.functional.conv3d(xs.to(dtype).double(), w5.to(dtype).double(), bias.double(), (1, 2, 2), (0, 3, 3)))
    ref = torch.nn.functional.max_pool3d(ref, (1, 3, 3), (1, 2, 2), (0, 1, 1)).permute(0, 2, 3, 4, 1).float()
    out = None
    if coff:
        out = torch.full(tuple(ref.shape[:4]) + (96,), 7.0, dtype=dtype, device="cuda")
    got = kernels.conv_stem64_pool(x.cuda(), t_index, wimg, bias.cuda(), True, out=out, out_coff=coff)
    if coff:
        assert (got[..., :coff] == 7.0).all() and (got[..., coff + 64:] == 7.0).all()
        got = got[..., coff:coff + 64]
    assert tuple(got.shape) == tuple(ref.shape)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert (got.float().cpu() - ref).abs().max().item() <= ulp * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_head_pool_follows_the_reference_for_every_grid():
    """pytorchvideo's head is AvgPool3d((8,7,7)) / ((32,7,7)), stride 1, then AdaptiveAvgPool3d(1) (SlowFast_features.py:150-152).
    Equal grid (32 x 224 x 224): one global mean.  Larger grid (here 32 x 256 x 224 -> 8 x 8 x 7 / 32 x 8 x 7): the real pool runs —
    the mean of the overlapping window means, NOT the global mean — in the one-call plan and in the layer-by-layer path alike.
    Smaller grid: the reference raises; so does the module unless ``head_small_grid = "mean"`` (reduced-size tests).  And
    ``forward([slow, fast])`` re-selects the slow frames on the device only for a slow tensor that IS pack_pathway_output's
    selection of that fast tensor: any other slow tensor is consumed as given."""
    import kvq_amd.models.backbones.slowfast_model as M
    from kvq_amd import _abi
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    m = M.slowfast()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    m = m.cuda().eval()
    x = torch.from_numpy(synth.synth_clip(5, 32, 256, 224, batch=1))
    with torch.no_grad():
        s_ref, f_ref = SF.slowfast_features(x, w)
        s1, f1 = m.forward_clips(x.cuda())
        M.CONVNET = False
        try:
            s0, f0 = m(M.pack_pathway_output(x.cuda()))
        finally:
            M.CONVNET = True
    for got, ref in ((s1, s_ref), (f1, f_ref), (s0, s_ref), (f0, f_ref)):
        assert ((got.cpu() - ref).norm() / ref.norm()).item() <= 5e-3
    with torch.no_grad(), pytest.raises(_abi.KvqError, match="smaller than the reference"):
        m.forward_clips(torch.zeros(1, 3, 16, 96, 64, device="cuda"))
    # a slow tensor that is NOT the packed selection is used as given
    m.head_small_grid = "mean"
    xs = torch.from_numpy(synth.synth_clip(6, 16, 96, 64, batch=1)).cuda()
    packed = M.pack_pathway_output(xs)
    other = [packed[0].flip(2).contiguous(), packed[1]]
    with torch.no_grad():
        a_s, a_f = m(packed)
        b_s, b_f = m(other)
        M.CONVNET = False
        try:
            c_s, c_f = m(other)                 # the layer-by-layer sequencing never re-selects anything
        finally:
            M.CONVNET = True
    assert not torch.equal(a_s, b_s)            # the flipped slow frames went through the network
    assert torch.equal(b_s, c_s) and torch.equal(b_f, c_f)


@pytest.mark.gpu
def test_extractor_features_do_not_depend_on_the_batch():
    """extract_video stacks up to 8 unique clips per forward; with split-K on, whether a late convolution cuts K depends on the tile
    count, i.e. on the batch — the extractor switches it off (kvq_convnet_splitk), so a clip's features are bit-identical whether it
    runs alone or beside 7 others (the reference runs batch 1)."""
    from kvq_amd.datasets.slowfast_clips import extract_video
    from kvq_amd.models.backbones.slowfast_model import slowfast
    w = synth.synth_params(SF.param_shapes(), 3, "stress", prefix="sf.")
    m = slowfast()
    m.head_small_grid = "mean"
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    m = m.cuda().eval()
    g = np.random.Generator(np.random.PCG64(12))
    clips = [torch.from_numpy(g.standard_normal((32, 3, 64, 64)).astype(np.float32)) for _ in range(5)]
    together = extract_video(m, clips, "cuda", batch=8)
    alone = extract_video(m, clips, "cuda", batch=1)
    for (s8, f8), (s1, f1) in zip(together, alone):
        assert np.array_equal(s8, s1) and np.array_equal(f8, f1)
    # a batch past every row-count threshold the pools ever had (the (mean, std) head pools chose their kernel — and with it the
    # summation order — from rows x channels until round 4): 40 clips in one forward against the same clips alone
    many = clips * 8
    big = extract_video(m, many, "cuda", batch=40)
    for i, (s40, f40) in enumerate(big):
        assert np.array_equal(s40, alone[i % 5][0]) and np.array_equal(f40, alone[i % 5][1])
