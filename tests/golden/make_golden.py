"""Pin the oracle to the real reference and emit the committed golden fixtures.

Runs ONLY in the build container (needs /root/reference).  For every case it
  1. runs the imported reference module,
  2. runs the oracle restatement on the same seeded inputs and asserts agreement
     (bit-exact for index/integer work, <= 2e-5 abs for fp32),
  3. stores the *reference's* outputs (or hashes / strided samples of them) in
     ``tests/golden/*.npz`` — data only, no reference source.

Inputs and weights are never stored: they are re-drawn from ``kvq_amd.utils.synth``
PCG64 streams, which are platform-stable.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [section ...]
"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

import kvq_amd  # noqa: F401  (import shim)
from kvq_amd.utils import synth
from oracle import sampler_oracle as SO
from oracle import swin3d_oracle as O
from _ref_import import import_reference


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def samples(t, n=2048):
    """Strided subset + checksums of a tensor: small, but sensitive to any local error."""
    a = np.asarray(t, np.float32).reshape(-1)
    idx = np.unique(np.linspace(0, a.size - 1, min(n, a.size)).astype(np.int64))
    return dict(idx=idx, val=a[idx], sum=np.float64(a.astype(np.float64).sum()),
                asum=np.float64(np.abs(a.astype(np.float64)).sum()), shape=np.asarray(t.shape))


def put(d, prefix, s):
    for k, v in s.items():
        d[f"{prefix}/{k}"] = v


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays")


# ------------------------------------------------------------------------------------------
def sec_layout(ref):
    """rel-pos index, fragment gate and shift mask: integer work, bit-exact."""
    d = {}
    attn = ref.swin.WindowAttention3D(96, (8, 7, 7), 3, qkv_bias=True, frag_bias=True)
    rpi = attn.relative_position_index.numpy()
    assert np.array_equal(rpi, O.rel_pos_index((8, 7, 7)))
    d["rpi_877"] = rpi.astype(np.int16)
    attn4 = ref.swin.WindowAttention3D(96, (4, 4, 4), 3, qkv_bias=True)
    assert np.array_equal(attn4.relative_position_index.numpy(), O.rel_pos_index((4, 4, 4)))
    d["rpi_444"] = attn4.relative_position_index.numpy().astype(np.int16)
    cases = []
    # (D,H,W) token grids: the four Swin-T stages at 32x224x224, T=96 stage 1, config-5 grids
    # (64x256x256 -> 32,64,64 / 32,32,32 / 32,16,16 / 32,8,8), clamp / odd cases
    for dims in [(16, 56, 56), (16, 28, 28), (16, 14, 14), (16, 7, 7), (48, 28, 28), (32, 64, 64),
                 (32, 32, 32), (32, 16, 16), (32, 8, 8), (4, 20, 20), (4, 10, 10), (4, 5, 5),
                 (4, 3, 3), (8, 16, 16), (10, 9, 23)]:
        for window in [(8, 7, 7), (4, 4, 4)]:
            for shifted in (False, True):
                shift = tuple(w // 2 for w in window) if shifted else (0, 0, 0)
                lay = O.window_layout(*dims, window, shift)
                ws, ss = ref.swin.get_window_size(dims, window, shift)
                assert (ws, ss) == (lay["ws"], lay["ss"])
                Dp, Hp, Wp = lay["Dp"], lay["Hp"], lay["Wp"]
                gpi = ref.swin.global_position_index(Dp, Hp, Wp, fragments=(1,) + ws[1:],
                                                     window_size=ws, shift_size=ss, device="cpu")
                g_ref = gpi.abs().sum(-1).numpy()
                g = O.frag_gate(lay)
                assert np.array_equal(g_ref, g), (dims, window, shift)
                tag = "%d_%d_%d__%d%d%d__%d" % (dims + window + (int(shifted),))
                d[f"gate/{tag}/sha"] = np.frombuffer(bytes.fromhex(sha(g_ref.astype(np.int8))), np.uint8)
                d[f"gate/{tag}/max"] = np.int64(g_ref.max())
                m = O.shift_mask(lay)
                if any(s > 0 for s in ss):
                    m_ref = ref.swin.compute_mask(Dp, Hp, Wp, ws, ss, "cpu").numpy()
                    assert np.array_equal(m_ref, m), (dims, window, shift)
                    d[f"mask/{tag}/sha"] = np.frombuffer(bytes.fromhex(sha((m_ref != 0).astype(np.int8))), np.uint8)
                else:
                    assert m is None
                # window gather map == pad + roll + window_partition of an index tensor
                D, H, W = dims
                idx = torch.arange(D * H * W, dtype=torch.float32).reshape(1, D, H, W, 1) + 1
                idx = torch.nn.functional.pad(idx, (0, 0, 0, Wp - W, 0, Hp - H, 0, Dp - D))
                if any(s > 0 for s in ss):
                    idx = torch.roll(idx, shifts=tuple(-s for s in ss), dims=(1, 2, 3))
                src_ref = ref.swin.window_partition(idx, ws).reshape(-1).long().numpy() - 1
                assert np.array_equal(src_ref, lay["src"]), (dims, window, shift)
                d[f"src/{tag}/sha"] = np.frombuffer(bytes.fromhex(sha(src_ref.astype(np.int32))), np.uint8)
                cases.append(tag)
    d["cases"] = np.asarray(cases)
    save("layout.npz", d)


TRUNK_CASES = [
    # name, cfg name, scheme, wseed, clip seed, B, T, H, W
    ("t_grpb_stress_8x80", "SWIN_T_GRPB", "stress", 0, 11, 2, 8, 80, 80),
    ("t_grpb_stress_16x64", "SWIN_T_GRPB", "stress", 0, 12, 1, 16, 64, 64),
    ("t_plain_stress_16x96", "SWIN_T_PLAIN", "stress", 1, 13, 1, 16, 96, 96),
    ("t_grpb_stress_10x50x70", "SWIN_T_GRPB", "stress", 2, 14, 1, 10, 50, 70),
    ("t_grpb_stress_32x224", "SWIN_T_GRPB", "stress", 0, 15, 1, 32, 224, 224),
    ("t_grpb_init_32x224", "SWIN_T_GRPB", "init", 3, 16, 1, 32, 224, 224),
    # the other model.py keys (model.py:39-47): swin_small (depths 2/2/18/2) and swin_tiny_grpb_m (window 4,4,4)
    ("s_plain_stress_16x96", "SWIN_S_PLAIN", "stress", 4, 17, 1, 16, 96, 96),
    ("s_plain_init_16x96", "SWIN_S_PLAIN", "init", 7, 20, 1, 16, 96, 96),
    ("t_m444_stress_16x96", "SWIN_T_GRPB_M", "stress", 5, 18, 1, 16, 96, 96),
    ("t_m444_stress_12x72x104", "SWIN_T_GRPB_M", "stress", 6, 19, 2, 12, 72, 104),
]


def _ref_trunk(ref, cfg):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.swin.SwinTransformer3D(pretrained=None, use_checkpoint=False, embed_dim=cfg.embed_dim,
                                       depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                                       window_size=cfg.window, frag_biases=list(cfg.frag_biases))
    m.eval()
    return m


def sec_trunk(ref):
    d = {}
    names = []
    for name, cfgn, scheme, wseed, cseed, B, T, H, W in TRUNK_CASES:
        cfg = getattr(synth, cfgn)
        wts = synth.synth_swin_weights(cfg, wseed, scheme)
        hw = synth.synth_vqa_head_weights(cfg.num_features, 64, wseed, scheme)
        m = _ref_trunk(ref, cfg)
        missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in wts.items()}, strict=False)
        assert not missing.unexpected_keys and all("relative_position_index" in k for k in missing.missing_keys)
        head = ref.head.VQAHead(in_channels=cfg.num_features, hidden_channels=64).eval()
        head.load_state_dict({k: torch.from_numpy(v) for k, v in hw.items()})
        x = torch.from_numpy(synth.synth_clip(cseed, T, H, W, batch=B))
        with torch.no_grad():
            feat_ref = m({"technical": x})
            score_ref = head(feat_ref)
            feat, stages = O.swin3d_trunk(x, wts, cfg, return_stages=True)
            score = O.vqa_head(feat, hw)
        err = float((feat - feat_ref).abs().max())
        serr = float((score - score_ref).abs().max())
        print(f"{name}: feat {tuple(feat_ref.shape)} |oracle-ref| {err:.2e}  score {score_ref.flatten().tolist()} d {serr:.2e}")
        assert err <= 2e-5 and serr <= 1e-6
        put(d, f"{name}/feat", samples(feat_ref.numpy()))
        d[f"{name}/score"] = score_ref.numpy()
        d[f"{name}/meta"] = np.asarray([wseed, cseed, B, T, H, W])
        d[f"{name}/cfg"] = np.asarray(cfgn)
        d[f"{name}/scheme"] = np.asarray(scheme)
        names.append(name)
    d["cases"] = np.asarray(names)
    save("trunk.npz", d)


ADAPTIVE_CASES = [
    # forward(adaptive_window_size=True) (swin_backbone.py:1050-1055): window = (8,7,7) * clip // (32,224,224)
    ("t_grpb_stress_adaptive_16x160", "SWIN_T_GRPB", "stress", 0, 21, 1, 16, 160, 160),        # window (4,5,5), no padding
    ("t_grpb_stress_adaptive_24x128x176", "SWIN_T_GRPB", "stress", 2, 22, 2, 24, 128, 176),    # window (6,4,5), W padded 44 -> 45
]


def sec_adaptive(ref):
    """The reference trunk with adaptive_window_size=True — the branch no caller takes."""
    import contextlib
    import io
    d = {}
    names = []
    for name, cfgn, scheme, wseed, cseed, B, T, H, W in ADAPTIVE_CASES:
        cfg = getattr(synth, cfgn)
        wts = synth.synth_swin_weights(cfg, wseed, scheme)
        m = _ref_trunk(ref, cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in wts.items()}, strict=False)
        x = torch.from_numpy(synth.synth_clip(cseed, T, H, W, batch=B))
        aw = tuple((w * xs) // bs for w, xs, bs in zip(cfg.window, (T, H, W), (32, 224, 224)))
        with torch.no_grad():
            with contextlib.redirect_stdout(io.StringIO()):
                feat_ref = m({"technical": x}, adaptive_window_size=True)
                feat_plain = m({"technical": x})
            feat = O.swin3d_trunk(x, wts, cfg, adaptive_window=aw)
        err = float((feat - feat_ref).abs().max())
        print(f"{name}: window {aw} feat {tuple(feat_ref.shape)} |oracle-ref| {err:.2e}; |adaptive - plain| {float((feat_ref - feat_plain).abs().max()):.2e}")
        assert err <= 2e-5 and float((feat_ref - feat_plain).abs().max()) > 1e-2
        put(d, f"{name}/feat", samples(feat_ref.numpy()))
        d[f"{name}/meta"] = np.asarray([wseed, cseed, B, T, H, W])
        d[f"{name}/window"] = np.asarray(aw)
        d[f"{name}/cfg"] = np.asarray(cfgn)
        d[f"{name}/scheme"] = np.asarray(scheme)
        names.append(name)
    d["cases"] = np.asarray(names)
    save("adaptive.npz", d)


def sec_heads(ref):
    d = {}
    g = np.random.Generator(np.random.PCG64(77))
    feat = g.standard_normal((3, 768, 4, 7, 7)).astype(np.float32)
    hw = synth.synth_vqa_head_weights(768, 64, 5, "stress")
    head = ref.head.VQAHead(in_channels=768, hidden_channels=64).eval()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in hw.items()})
    with torch.no_grad():
        s_ref = head(torch.from_numpy(feat)).numpy()
    assert np.abs(O.vqa_head(torch.from_numpy(feat), hw).numpy() - s_ref).max() < 1e-6
    d["vqa/score"] = s_ref
    # the branches no config takes (head.py:61-62 pre_pool, :66-67 num_class > 1 with nn.Softmax()'s implicit dim)
    import warnings
    for tag, K, pool in (("pool", 1, True), ("k3", 3, False), ("k5pool", 5, True)):
        hk = synth.synth_vqa_head_weights(768, 64, 6, "stress", num_class=K)
        hd = ref.head.VQAHead(in_channels=768, hidden_channels=64, num_class=K, pre_pool=pool).eval()
        hd.load_state_dict({k: torch.from_numpy(v) for k, v in hk.items()})
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = hd(torch.from_numpy(feat)).numpy()
        assert sk.shape == (3, K)
        assert np.abs(O.vqa_head(torch.from_numpy(feat), hk, pre_pool=pool).numpy() - sk).max() < 1e-6
        d[f"vqa/{tag}/score"] = sk
    f2 = g.standard_normal((2, 8, 9472)).astype(np.float32)
    sw = synth.synth_simple_head_weights(9472, 128, 5, "stress")
    sh = ref.head.simpleVQAHead(9472, 128).eval()
    sh.load_state_dict({k: torch.from_numpy(v) for k, v in sw.items()})
    with torch.no_grad():
        s2 = sh(torch.from_numpy(f2)).numpy()
    assert np.abs(O.simple_vqa_head(torch.from_numpy(f2), sw).numpy() - s2).max() < 1e-5
    d["simple/score"] = s2
    save("heads.npz", d)


def sec_sampler(ref):
    import random as pyrandom
    d = {}
    fd = ref.fd
    # (a) fragment sampler, KSVQE-style 9x9 grid and the bench's 7x7 grid, aligned 8
    for tag, (T, H, W, Fh, Fw, fs, al, seed) in {
        "k9": (16, 400, 720, 9, 9, 32, 8, 101),
        "b7": (32, 270, 480, 7, 7, 32, 8, 102),
        "tight": (8, 224, 230, 7, 7, 32, 4, 103),     # cell == patch on H -> zero offsets branch
    }.items():
        g = np.random.Generator(np.random.PCG64(seed))
        video = g.integers(0, 256, size=(3, T, H, W)).astype(np.float32)
        torch.manual_seed(seed)
        out_ref = fd.get_spatial_fragments(torch.from_numpy(video), Fh, Fw, fs, fs, aligned=al).numpy()
        torch.manual_seed(seed)
        rh, rw = SO.draw_fragment_offsets(T, H, W, Fh, Fw, fs, fs, al)
        out = SO.spatial_fragments(video, rh, rw, Fh, Fw, fs, fs, al)
        assert np.array_equal(out, out_ref), tag
        d[f"frag/{tag}/meta"] = np.asarray([T, H, W, Fh, Fw, fs, al, seed])
        d[f"frag/{tag}/rnd_h"], d[f"frag/{tag}/rnd_w"] = rh, rw
        d[f"frag/{tag}/sha"] = np.frombuffer(bytes.fromhex(sha(out_ref.astype(np.uint8))), np.uint8)
        norm_ref = ((torch.from_numpy(out_ref).permute(1, 2, 3, 0) - torch.FloatTensor(synth.KVQ_MEAN))
                    / torch.FloatTensor(synth.KVQ_STD)).permute(3, 0, 1, 2).numpy()   # fusion_datasets.py:1017-1020
        assert np.array_equal(SO.normalize(out, synth.KVQ_MEAN, synth.KVQ_STD), norm_ref)
        put(d, f"frag/{tag}/norm", samples(norm_ref, 1024))
    # (a') sources smaller than the canvas: the bilinear-upsample fallback (fusion_datasets.py:43-50), uint8 and fp32 frames,
    # incl. flat regions (where (v / 255) * 255 truncates to v or v - 1 depending on the kernel's roundings)
    for tag, (T, H, W, Fh, Fw, fs, al, seed, u8) in {
        "up_u8": (8, 200, 300, 7, 7, 32, 4, 111, 1),
        "up_f32": (4, 180, 224, 7, 7, 32, 4, 112, 0),
        "up_k9": (4, 250, 270, 9, 9, 32, 2, 113, 1),
    }.items():
        g = np.random.Generator(np.random.PCG64(seed))
        video = g.integers(0, 256, size=(3, T, H, W)).astype(np.uint8)
        video[:, :, :40, :48] = 100
        video[:, :, 50:90] = 37
        video = video if u8 else video.astype(np.float32)
        torch.manual_seed(seed)
        out_ref = fd.get_spatial_fragments(torch.from_numpy(video), Fh, Fw, fs, fs, aligned=al).numpy()
        torch.manual_seed(seed)
        rh, rw = SO.draw_fragment_offsets(T, H, W, Fh, Fw, fs, fs, al)
        out = SO.spatial_fragments(video, rh, rw, Fh, Fw, fs, fs, al)
        assert out.shape == out_ref.shape and np.array_equal(out.astype(np.float32), out_ref), tag
        d[f"frag/{tag}/meta"] = np.asarray([T, H, W, Fh, Fw, fs, al, seed, u8])
        d[f"frag/{tag}/sha"] = np.frombuffer(bytes.fromhex(sha(out_ref.astype(np.float32))), np.uint8)
    # (b) temporal sampler: KSVQE val call (32, 3, 4) x1 clip and SimpleVQA (1, 8, 10, 1)
    for tag, (n, fs_t, ft, iv, nc, seed) in {"ksvqe": (300, 32, 3, 4, 1, 7), "simple": (300, 1, 8, 10, 1, 8),
                                               "short": (90, 32, 3, 4, 1, 9), "clips3": (500, 32, 1, 2, 3, 10)}.items():
        np.random.seed(seed)
        pyrandom.seed(seed)
        ref_idx = fd.UnifiedFrameSampler(fs_t, ft, iv, nc)(n)
        np.random.seed(seed)
        rnd = SO.draw_frame_offsets(n, fs_t, ft, iv, nc)
        idx = SO.frame_indices(n, fs_t, ft, iv, rnd)
        assert np.array_equal(idx, ref_idx) and idx.dtype == ref_idx.dtype, tag
        d[f"frames/{tag}/meta"] = np.asarray([n, fs_t, ft, iv, nc, seed])
        d[f"frames/{tag}/rnd"] = rnd
        d[f"frames/{tag}/idx"] = ref_idx
    # (c) harness math: rescale + SRCC/PLCC/KRCC/RMSE on a fixed 900-vector (trainer.py:287-292)
    g = np.random.Generator(np.random.PCG64(900))
    labels = g.uniform(1, 5, 900)
    preds = 0.3 * labels + g.standard_normal(900) * 0.2 - 1.0
    tr = ref.trainer
    p = tr.Trainer.rescale(None, list(preds), list(labels))
    s, pl, k = tr.spearmanr(labels, p)[0], tr.pearsonr(labels, p)[0], tr.kendallr(labels, p)[0]
    r = np.sqrt(((labels - p) ** 2).mean())
    mine = SO.quality_metrics(preds, labels)
    assert np.allclose(mine, (s, pl, k, r), rtol=0, atol=1e-12)
    d["metrics/srcc_plcc_krcc_rmse"] = np.asarray([s, pl, k, r])
    d["metrics/rescaled_head"] = np.asarray(p[:8])
    # clip reshape (trainer.py:192-201)
    x = torch.arange(2 * 3 * 12 * 2 * 2, dtype=torch.float32).reshape(2, 3, 12, 2, 2)
    b, c, t, h, w = x.shape
    xr = x.reshape(b, c, 3, t // 3, h, w).permute(0, 2, 1, 3, 4, 5).reshape(b * 3, c, t // 3, h, w)
    assert np.array_equal(SO.split_clips(x.numpy(), 3), xr.numpy())
    save("sampler.npz", d)


def sec_resnet(ref):
    """SimpleVQA spatial branch (2D ResNet-50 + avg/std pooling + feature concat) and the whole
    VQA_Network(simpleVQA) score on seeded weights."""
    from oracle import resnet_oracle as RO
    d = {}
    g = np.random.Generator(np.random.PCG64(55))
    for name, (B, T, H, W) in {"r50_2x96": (1, 2, 96, 96), "r50_b2_3x64x80": (2, 3, 64, 80)}.items():
        wts = synth.synth_resnet50_weights(4, "stress")
        hw = synth.synth_simple_head_weights(9472, 128, 4, "stress")
        m = ref.simple.ResNet(ref.simple.Bottleneck, [3, 4, 6, 3]).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in wts.items()})
        head = ref.head.simpleVQAHead(9472, 128).eval()
        head.load_state_dict({k: torch.from_numpy(v) for k, v in hw.items()})
        frames = torch.from_numpy(g.standard_normal((B, 3, T, H, W)).astype(np.float32))
        feat3d = torch.from_numpy(g.standard_normal((B, T, 2304)).astype(np.float32))
        with torch.no_grad():
            f_ref = m({"simpleVQA": frames, "feat": feat3d})
            s_ref = head(f_ref)
            f = RO.simplevqa_features(frames, feat3d, wts)
            s = O.simple_vqa_head(f, hw)
        err = float((f - f_ref).abs().max() / f_ref.abs().max())
        print(f"{name}: feat {tuple(f_ref.shape)} rel |oracle-ref| {err:.2e} score {s_ref.flatten().tolist()} d {float((s - s_ref).abs().max()):.2e}")
        assert err <= 1e-5 and float((s - s_ref).abs().max()) <= 1e-4
        d[f"{name}/feat"] = f_ref.numpy()
        d[f"{name}/score"] = s_ref.numpy()
        d[f"{name}/meta"] = np.asarray([B, T, H, W])
    save("resnet.npz", d)


def sec_taps(ref):
    """multi=True / layer>-1 feature taps of SwinTransformer3D.forward (swin_backbone.py:1060-1078)."""
    import contextlib
    import io
    import torch.nn.functional as F
    d = {}
    name, cfgn, scheme, wseed, cseed, B, T, H, W = TRUNK_CASES[0]
    cfg = getattr(synth, cfgn)
    wts = synth.synth_swin_weights(cfg, wseed, scheme)
    m = _ref_trunk(ref, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in wts.items()}, strict=False)
    x = torch.from_numpy(synth.synth_clip(cseed, T, H, W, batch=B))
    with contextlib.redirect_stdout(io.StringIO()):
        multi_ref = m({"technical": x}, multi=True)
        layers_ref = [m({"technical": x}, layer=i) for i in range(5)]
    feat, stages = O.swin3d_trunk(x, wts, cfg, return_stages=True)
    st_cf = [s.permute(0, 4, 1, 2, 3) for s in stages]
    multi = torch.cat([F.interpolate(s, size=feat.shape[2:], mode="trilinear") for s in st_cf[:-1]], 1)
    e1 = float((multi - multi_ref).abs().max())
    e2 = max(float((a - b).abs().max()) for a, b in zip(st_cf, layers_ref))
    print(f"taps {name}: multi {tuple(multi_ref.shape)} |oracle-ref| {e1:.2e}; layers {e2:.2e}")
    assert e1 <= 2e-5 and e2 <= 2e-5
    put(d, "multi", samples(multi_ref.numpy(), 4096))
    for i, t in enumerate(layers_ref):
        put(d, f"layer{i}", samples(t.contiguous().numpy(), 2048))
    d["case"] = np.asarray(name)
    save("taps.npz", d)


CLIP_CASES = [("clip_112", 3, 112, 112, 21), ("clip_224", 2, 224, 224, 22), ("clip_96x128", 2, 96, 128, 23)]


def sec_clip(ref):
    """KSVQE's CLIP_tool: the reference extractor over the vendored CLIP ViT-B/16 with synthetic weights."""
    import contextlib
    import importlib
    import io
    from oracle import clip_oracle as CO
    with contextlib.redirect_stdout(io.StringIO()):
        CB = importlib.import_module("models.backbones.CLIP_backbone")
        CM = importlib.import_module("models.backbones.clip.model")
    wts = synth.synth_clip_visual_weights(7)
    vis = CM.VisionTransformer(input_resolution=224, patch_size=16, width=768, layers=12, heads=12, output_dim=512)
    ext = CB.CLIP_extractor_addadapter_cls(visual=vis, CLIP_location=8, cls_use=True).eval()
    missing = ext.load_state_dict({k: torch.from_numpy(v) for k, v in wts.items()}, strict=True)
    # the reference's blocks run under torch.utils.checkpoint by default (model.py:206): same numbers in eval / no_grad
    d = {}
    for name, B, H, W, seed in CLIP_CASES:
        g = np.random.Generator(np.random.PCG64(seed))
        x = torch.from_numpy(g.standard_normal((B, 3, H, W)).astype(np.float32))
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            ra, rc, rp = ext(x)
            oa, oc, op = CO.clip_visual_extractor(x, wts)
        e = [float((a - b).abs().max()) for a, b in ((ra, oa), (rc, oc), (rp, op))]
        print(f"{name}: cls_attn {tuple(ra.shape)} cls {tuple(rc.shape)} pat {tuple(rp.shape)}  |oracle-ref| {e[0]:.2e} {e[1]:.2e} {e[2]:.2e}")
        assert e[0] <= 2e-5 and e[1] <= 2e-4 and e[2] <= 2e-4
        put(d, f"{name}/cls_attn", samples(ra.numpy()))
        put(d, f"{name}/cls_token", samples(rc.numpy()))
        put(d, f"{name}/pat_token", samples(rp.numpy(), 4096))
        d[f"{name}/meta"] = np.asarray([B, H, W, seed])
    d["cases"] = np.asarray([c[0] for c in CLIP_CASES])
    save("clip.npz", d)


def cdm_inputs(seed=31, n=2, t=16, hw=49, nk=49, dim=768):
    g = np.random.Generator(np.random.PCG64(seed))
    r = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32))     # noqa: E731
    return dict(Q=r(n * t, hw, dim), K=r(n * t, nk, dim), xs=r(n * hw, t, dim), sem_x=r(n * t, dim, 7, 7), sem_in=r(n * t, dim, 7, 7),
                dist_x=r(n, dim, t, 7, 7) * 1.5 + 0.3, dist_in=r(n, t * hw, dim))


def sec_cdm(ref):
    """KSVQE's CDM modules: the reference classes with synthetic weights on seeded inputs."""
    import contextlib
    import importlib
    import io
    from oracle import ksvqe_oracle as KO
    with contextlib.redirect_stdout(io.StringIO()):
        K = importlib.import_module("models.backbones.KSVQE_model")
    w = {m: {k: torch.from_numpy(v) for k, v in sd.items()} for m, sd in synth.synth_cdm_weights(11).items()}
    x = cdm_inputs()
    mods = dict(cross=K.crossattention1(768, 12), self=K.Attention(768, 12), sem=K.Semantic_Transformation2(768),
                dist=K.Dist_Transformation3(768))
    for k, m in mods.items():
        m.load_state_dict(w[k], strict=True)
        m.eval()
    with torch.no_grad():
        ref_out = dict(cross=mods["cross"](x["Q"], x["K"])[0], cross_A=mods["cross"](x["Q"], x["K"])[1], self=mods["self"](x["xs"]),
                       sem=mods["sem"](x["sem_x"], x["sem_in"]), dist=mods["dist"](x["dist_x"], x["dist_in"]))
        o, a = KO.cross_attention(x["Q"], x["K"], w["cross"], 12)
        mine = dict(cross=o, cross_A=a, self=KO.self_attention(x["xs"], w["self"], 12),
                    sem=KO.semantic_transformation2(x["sem_x"], x["sem_in"], w["sem"]),
                    dist=KO.dist_transformation3(x["dist_x"], x["dist_in"], w["dist"]))
    d = {}
    for k in ref_out:
        e = float((ref_out[k] - mine[k]).abs().max())
        print(f"cdm {k}: {tuple(ref_out[k].shape)} |oracle-ref| {e:.2e}")
        assert e <= 2e-5
        put(d, k, samples(ref_out[k].contiguous().numpy(), 4096))
    save("cdm.npz", d)


def qrs_inputs(seed=41, b=2, t=16, n_key=4, hw=288):
    g = np.random.Generator(np.random.PCG64(seed))
    return (torch.from_numpy(g.standard_normal((b, 3, t, hw, hw)).astype(np.float32)),
            torch.from_numpy(g.uniform(-1, 1, (b, n_key, 49)).astype(np.float32)))


def sec_qrs(ref):
    """KSVQE's key-frame selection and QRS eval path: the reference's own functions on seeded inputs."""
    import contextlib
    import importlib
    import io
    import types
    from oracle import ksvqe_oracle as KO
    with contextlib.redirect_stdout(io.StringIO()):
        K = importlib.import_module("models.backbones.KSVQE_model")
        P = importlib.import_module("models.backbones.patchnet")
    x, score = qrs_inputs()
    fake = types.SimpleNamespace()
    with torch.no_grad():
        gid_ref, key_ref = K.KSVQE.obtain_keyframes(fake, x[:, :, :, :16, :16].contiguous())
        gid, key = KO.obtain_keyframes(x[:, :, :, :16, :16].contiguous())
        assert torch.equal(gid, gid_ref) and torch.equal(key, key_ref)
        net = P.RegionNet_CLIP(k=49, anchor_size=32, stride=1, num_samples=1, sample_type="topkpertubation").eval()
        out_ref = net(x, score, 0.5, gid_ref)
        out, idx = KO.qrs_select(x, score, gid_ref)
    e = float((out - out_ref).abs().max())
    print(f"qrs: patches {tuple(out_ref.shape)} |oracle-ref| {e:.2e}; regions {idx.tolist()}; group ids {gid_ref[0].int().tolist()}")
    assert e == 0.0
    d = {"gid": gid_ref.numpy(), "idx": idx.numpy().astype(np.int32)}
    put(d, "patches", samples(out_ref.numpy(), 8192))
    put(d, "key", samples(key_ref.numpy(), 1024))
    save("qrs.npz", d)


def sec_contrique(ref):
    """KSVQE's CONTRIQUE branch.  torchvision is absent here, so the reference's ``CONTRIQUE_model`` is built
    around a torchvision-ORDERED shell of the reference's OWN ResNet-50 blocks (simpleVQA_model.py Bottleneck: the same
    v1.5 structure as torchvision.models.resnet50) — the patching, normalisation and projector are the reference's code,
    the encoder's block arithmetic is the reference's SimpleVQA restatement of it."""
    import contextlib
    import importlib
    import io
    from oracle import ksvqe_oracle as KO
    with contextlib.redirect_stdout(io.StringIO()):
        K = importlib.import_module("models.backbones.KSVQE_model")
        r = ref.simple.resnet50(pretrained=False)
    shell = torch.nn.Module()
    for name in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool"):
        shell.add_module(name, getattr(r, name))
    shell.add_module("fc", torch.nn.Linear(2048, 10))
    m = K.CONTRIQUE_model(shell, 2048).eval()
    wts = synth.synth_contrique_weights(13)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in wts.items()}, strict=True)
    g = np.random.Generator(np.random.PCG64(51))
    x = torch.from_numpy(g.standard_normal((1, 3, 3, 64, 96)).astype(np.float32))
    with torch.no_grad():
        z_ref = m(x)
        z = KO.contrique(x, wts)
    e = float((z - z_ref).abs().max())
    print(f"contrique: {tuple(z_ref.shape)} |oracle-ref| {e:.2e}  |z| max {float(z_ref.abs().max()):.3f}")
    assert e <= 2e-4
    d = {"z": z_ref.numpy()}
    save("contrique.npz", d)


def sec_ksvqe(ref):
    """The whole KSVQE.forward of the REFERENCE with synthetic weights.  Its constructor fetches CLIP and a
    CONTRIQUE checkpoint from absolute paths and needs torchvision (KSVQE_model.py:1068-1074, :1608): those three hooks are
    replaced here by the reference's own classes (vendored ViT, CONTRIQUE_model over the torchvision-ordered shell of the
    reference's ResNet-50 blocks) — the forward itself is untouched reference code."""
    import contextlib
    import importlib
    import io
    from oracle import ksvqe_oracle as KO
    with contextlib.redirect_stdout(io.StringIO()):
        K = importlib.import_module("models.backbones.KSVQE_model")
        CB = importlib.import_module("models.backbones.CLIP_backbone")
        CM = importlib.import_module("models.backbones.clip.model")

    def shell(*a, **k):
        with contextlib.redirect_stdout(io.StringIO()):
            r = ref.simple.resnet50(pretrained=False)
        sh = torch.nn.Module()
        for name in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool"):
            sh.add_module(name, getattr(r, name))
        sh.add_module("fc", torch.nn.Linear(2048, 10))
        return sh

    K.get_network = shell
    K.build_CLIPmodel_basedadapter_cls = lambda CLIP_location=None, cls_use=None, **k: CB.CLIP_extractor_addadapter_cls(
        visual=CM.VisionTransformer(224, 16, 768, 12, 12, 512), CLIP_location=CLIP_location, cls_use=cls_use)
    wts = synth.synth_ksvqe_weights(3)
    real_load = torch.load
    torch.load = lambda path, *a, **k: ({kk[len("distortion_tool."):]: torch.from_numpy(v) for kk, v in wts.items()
                                         if kk.startswith("distortion_tool.")} if "CONTRIQUE" in str(path) else real_load(path, *a, **k))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            m = K.KSVQE(pretrained=None, num_samples=1, sample_type="topkpertubation", CLIP_location=8, cls_use=True, tuning_stage=2,
                        use_checkpoint=False)
    finally:
        torch.load = real_load
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in wts.items()}, strict=False)
    assert not missing.unexpected_keys and all("relative_position_index" in k for k in missing.missing_keys), missing
    m.eval()
    inp = {k: torch.from_numpy(v) for k, v in synth.synth_ksvqe_inputs(4, b=2).items()}
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        f_ref, l_ref = m(inp)
        f, l = KO.ksvqe_forward(inp, wts, synth.SWIN_T_GRPB)
    e = float((f - f_ref).abs().max())
    print(f"ksvqe: feat {tuple(f_ref.shape)} |oracle-ref| {e:.2e} (max |f| {float(f_ref.abs().max()):.2f}); loss {float(l_ref):.6f} vs {float(l):.6f}")
    assert e <= 5e-4 and abs(float(l) - float(l_ref)) <= 1e-4
    d = {"loss": np.float64(l_ref)}
    put(d, "feat", samples(f_ref.numpy(), 8192))
    # the feature taps no caller asks for (:1489-1498): multi = resized concat of feats[:-1]; layer = feats[layer]
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        taps_ref = {"multi": m(inp, multi=True), "layer0": m(inp, layer=0), "layer2": m(inp, layer=2), "layer4": m(inp, layer=4)}
        taps_orc = {"multi": KO.ksvqe_forward(inp, wts, synth.SWIN_T_GRPB, multi=True)}
        for k in (0, 2, 4):
            taps_orc[f"layer{k}"] = KO.ksvqe_forward(inp, wts, synth.SWIN_T_GRPB, layer=k)
    for k, v in taps_ref.items():
        e = float((taps_orc[k] - v).abs().max())
        print(f"ksvqe {k}: {tuple(v.shape)} |oracle-ref| {e:.2e} (max {float(v.abs().max()):.2f})")
        assert taps_orc[k].shape == v.shape and e <= 5e-4 * max(1.0, float(v.abs().max()))
        put(d, k, samples(v.numpy(), 4096))
    save("ksvqe.npz", d)


def sec_ckpt(ref):
    """Checkpoint formats: what the REFERENCE's inflate_weights / load_swin leave in the trunk's state
    dict for synthetic 2D / Video-Swin checkpoints (kvq_amd.utils.synth), and the build's loaders on the same files."""
    import contextlib
    import io
    import tempfile
    from kvq_amd.models.backbones.swin_backbone import SwinTransformer3D as Mine
    d = {}
    cfg = synth.SWIN_T_GRPB
    tmp = tempfile.mkdtemp(prefix="kvq_ckpt_")
    cases = [("inflate_w7", "2d", 7), ("inflate_w12", "2d", 12), ("load_swin", "3d", 0)]
    for name, kind, w2d in cases:
        path = os.path.join(tmp, name + ".pth")
        if kind == "2d":
            body = {"model": {k: torch.from_numpy(v) for k, v in synth.synth_swin2d_checkpoint(cfg, 5, w2d).items()}}
        else:
            body = {"state_dict": {k: torch.from_numpy(v) for k, v in synth.synth_swin3d_checkpoint(cfg, 6).items()}}
        torch.save(body, path)
        m = _ref_trunk(ref, cfg)
        mine = Mine()
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            if kind == "2d":
                m.pretrained = path
                m.inflate_weights()
                mine.pretrained = path
                mine.inflate_weights()
            else:
                m.load_swin(path)
                mine.load_swin(path)
        sr, sm = m.state_dict(), mine.state_dict()
        keys = [k for k in sr if "relative_position_index" not in k and (
            "position_bias_table" in k or k.startswith("patch_embed") or k.startswith("norm") or "blocks.1.attn.qkv" in k)]
        worst = 0.0
        for k in keys:
            worst = max(worst, float((sr[k] - sm[k]).abs().max()))
            if kind == "3d" and "fragment_position_bias_table" not in k and k != "norm.weight" and not k.startswith("patch_embed.norm"):
                assert (sr[k] - sm[k]).abs().max() == 0
        print(f"{name}: {len(keys)} tensors, |mine - ref| max {worst:.2e}")
        assert worst <= 1e-6
        for k in keys:
            put(d, f"{name}/{k}", samples(sr[k].numpy(), 256))
        d[f"{name}/keys"] = np.asarray(keys)
    d["cases"] = np.asarray([c[0] for c in cases])
    save("ckpt.npz", d)


SFCLIP_CASES = [  # (frames, fps reported by the container, frames the decoder delivers)
    (300, 30.0, 300), (250, 29.97, 250), (100, 30.0, 100), (70, 30.0, 70), (61, 30.0, 61), (50, 30.0, 50), (45, 25.0, 45),
    (20, 10.0, 20), (40, 0.0, 40), (96, 23.976, 96), (200, 60.0, 200), (300, 30.0, 270), (64, 30.0, 40), (10, 30.0, 10)]


def sec_sfclips(ref):
    """Clip assembly of the motion-feature extractor: the reference's OWN ``VideoDataset_NR_SlowFast_feature.__getitem__``
    (SlowFast_features.py:52-107) run over a fake cv2.VideoCapture whose frame i carries i in its pixels; stored: the frame
    index of every slot of every clip."""
    import importlib
    import types
    from kvq_amd.datasets.slowfast_clips import clip_frame_indices
    state = {}

    class FakeCapture:
        def __init__(self, *a):
            self.pos = 0

        def open(self, filename):
            self.pos = 0

        def get(self, prop):
            return {7: float(state["length"]), 5: float(state["fps"])}[prop]

        def read(self):
            i = self.pos
            self.pos += 1
            if i >= state["readable"]:
                return False, None
            f = np.zeros((2, 2, 3), np.uint8)
            f[..., 0], f[..., 1] = i % 251, i // 251          # "BGR"; cvtColor below reverses the channels
            return True, f

        def release(self):
            pass

    cv2 = sys.modules["cv2"]
    cv2.VideoCapture, cv2.CAP_PROP_FRAME_COUNT, cv2.CAP_PROP_FPS, cv2.COLOR_BGR2RGB = FakeCapture, 7, 5, 4
    cv2.cvtColor = lambda frame, code: frame[..., ::-1].copy()
    hub = types.ModuleType("pytorchvideo.models.hub")
    hub.slowfast_r50 = None
    for n in ("pytorchvideo", "pytorchvideo.models"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["pytorchvideo.models.hub"] = hub
    sf = importlib.import_module("SlowFast_features")
    csv_path = os.path.join(os.getcwd(), "one.csv")
    with open(csv_path, "w") as f:
        f.write("filename,score\nfake.mp4,1\n")
    d, names = {}, []
    to_t = lambda img: torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float()      # noqa: E731  (RGB after cvtColor)
    for length, fps, readable in SFCLIP_CASES:
        state.update(length=length, fps=fps, readable=readable)
        ds = sf.VideoDataset_NR_SlowFast_feature(types.SimpleNamespace(resize=2), to_t, os.getcwd(), csv_path)
        name = f"L{length}_f{fps:g}_r{readable}"
        try:
            clips, vname = ds[0]
            assert vname == "fake.mp4"
            idx = np.stack([(c[:, 2, 0, 0] + 251 * c[:, 1, 0, 0]).numpy().astype(np.int64) for c in clips])     # RGB: R = B of the fake
            mine = np.stack(clip_frame_indices(length, int(round(fps)), readable))
            assert np.array_equal(idx, mine), (name, idx, mine)
            d[f"{name}/idx"] = idx.astype(np.int32)
        except IndexError as e:
            try:
                clip_frame_indices(length, int(round(fps)), readable)
                raise AssertionError(f"{name}: the reference raised IndexError, the restatement did not")
            except IndexError:
                pass
            d[f"{name}/error"] = np.asarray(f"IndexError: {e}")
        d[f"{name}/meta"] = np.asarray([length, int(round(fps)), readable])
        names.append(name)
        print(name, "clips:", d.get(f"{name}/idx", np.zeros((0, 0))).shape[0], str(d.get(f"{name}/error", "")))
    d["cases"] = np.asarray(names)
    save("sfclips.npz", d)


SECTIONS = {"sfclips": sec_sfclips, "ksvqe": sec_ksvqe, "contrique": sec_contrique, "qrs": sec_qrs, "cdm": sec_cdm, "clip": sec_clip, "taps": sec_taps, "ckpt": sec_ckpt, "resnet": sec_resnet, "layout": sec_layout, "trunk": sec_trunk, "heads": sec_heads, "sampler": sec_sampler, "adaptive": sec_adaptive}


def main():
    want = sys.argv[1:] or list(SECTIONS)
    ref = import_reference()
    torch.set_grad_enabled(False)
    for s in want:
        SECTIONS[s](ref)


if __name__ == "__main__":
    main()
