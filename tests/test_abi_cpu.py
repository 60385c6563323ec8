"""CPU: the C-ABI library loads and exports every symbol include/kvq_hip.h declares (no compute
calls without a GPU); host-side module logic (state_dict surface, config dispatch, error paths)."""
import os
import re

import pytest
import torch

import kvq_amd  # noqa: F401
from kvq_amd import _abi, _build
from kvq_amd.models import VQA_Network
from kvq_amd.utils import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    _build.build()
    header = open(os.path.join(ROOT, "include", "kvq_hip.h")).read()
    declared = set(re.findall(r"\b(kvq_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 17
    handle = _abi.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in kvq_hip.h but not exported"
    assert declared == set(_abi.SYMBOLS), declared ^ set(_abi.SYMBOLS)
    assert handle.kvq_abi_version() == _abi.ABI_VERSION == 31


def test_struct_layouts_match_header_sizes():
    import ctypes as C
    assert C.sizeof(_abi.KvqSwinCfg) == 4 * (3 + 1 + 1 + 1 + 4 + 4 + 3 + 1 + 4 + 3)
    assert C.sizeof(_abi.KvqSwinBlockW) == 18 * 8
    assert C.sizeof(_abi.KvqBlockTailArgs) == 152          # ABI 31: + x_f16
    assert C.sizeof(_abi.KvqSwinWeights) == 5 * 8 + 8 + 3 * 4 * 8 + 2 * 8
    assert C.sizeof(_abi.KvqPatchMergeArgs) == 104         # ABI 31: + x_f16, out_f16
    assert C.sizeof(_abi.KvqPatchEmbedArgs) == 136         # ABI 31: + out_f16
    assert C.sizeof(_abi.KvqFragmentSource) == 3 * 16 * 8 + 8 + 10 * 4 + 2 * 16 + 8
    assert C.sizeof(_abi.KvqAttnDenseArgs) == 104
    assert C.sizeof(_abi.KvqGemmArgs) == 144
    assert C.sizeof(_abi.KvqConvArgs) == 160
    assert _abi.dtype_code('bf16') == 0 and _abi.dtype_code(torch.float16) == 1


def test_fragment_source_eligibility_is_host_logic():
    """kvq_patch_embed_fragments_supported decides on the host whether the embedding launch may read through the sampler
    (include/kvq_hip.h): uint8 frames, 4 x 4 patches inside the mini-patches, canvas = grid x mini-patch, source >= canvas,
    whole aligned frame groups, one clip per 32 tokens, a sane channel stride."""
    handle = _abi.lib()
    f = _abi.KvqFragmentSource()
    f.n_clips, f.src_is_u8, f.Hs, f.Ws, f.Fh, f.Fw, f.fs_h, f.fs_w, f.aligned = 4, 1, 540, 960, 7, 7, 32, 32, 8
    ok = lambda *a: handle.kvq_patch_embed_fragments_supported(f, *a)          # noqa: E731  (B, in_chans, pd, T, H, W)
    assert ok(4, 3, 2, 32, 224, 224) == 1
    assert ok(3, 3, 2, 32, 224, 224) == 0 and ok(17, 3, 2, 32, 224, 224) == 0       # batch != clips; more clips than the struct holds
    assert ok(4, 3, 2, 32, 224, 256) == 0                                            # canvas != grid x mini-patch
    assert ok(4, 3, 2, 30, 224, 224) == 0                                            # T % aligned
    assert ok(4, 5, 2, 32, 224, 224) == 0                                            # mean / std hold 4 channels
    f.chan_stride = 32 * 540 * 960 - 1
    assert ok(4, 3, 2, 32, 224, 224) == 0                                            # planes would overlap
    f.chan_stride = 256 * 540 * 960
    assert ok(4, 3, 2, 32, 224, 224) == 1                                            # clips = runs of frames of a 256-frame video
    f.src_is_u8 = 0
    assert ok(4, 3, 2, 32, 224, 224) == 0
    f.src_is_u8, f.fs_h, f.Fh = 1, 14, 16
    assert ok(4, 3, 2, 32, 224, 224) == 0                                            # patch rows would span two mini-patches
    f.fs_h, f.Fh, f.Hs = 32, 7, 200
    assert ok(4, 3, 2, 32, 224, 224) == 0                                            # source smaller than the canvas
    f.Hs, f.Fh, f.Fw = 540, 1, 1
    assert ok(4, 3, 2, 8, 32, 32) == 1                                               # 4 x 8 x 8 = 256 tokens per clip
    f.fs_h = f.fs_w = 12
    f.aligned = 2
    assert ok(4, 3, 2, 2, 12, 12) == 0                                               # 9 tokens per clip: a wave of 32 would straddle clips
    assert ok(4, 3, 2, 64, 12, 12) == 1                                              # 32 x 9 tokens
    assert handle.kvq_patch_embed_fragments_supported(None, 4, 3, 2, 32, 224, 224) == 0


def test_error_paths_without_gpu():
    handle = _abi.lib()
    rc = handle.kvq_gemm_bf16(None, None)
    assert rc == -1 and b"NULL" in handle.kvq_last_error()
    import ctypes as C
    cfg = _abi.KvqSwinCfg()
    out = C.c_void_p()
    rc = handle.kvq_swin3d_plan_create(C.byref(cfg), 1, 32, 224, 224, _abi.DT_FP16, C.byref(out))
    assert rc == -3          # num_stages == 0 -> unsupported, reported not crashed
    with pytest.raises(_abi.KvqError):
        _abi.check(rc, "plan")


def test_state_dict_surface_matches_reference_keys():
    net = VQA_Network({"model": {"args": {"swin_tiny_grpb": {"head": {"in_channels": 768, "hidden_channels": 64}}}}})
    sd = net.state_dict()
    for k, shp in synth.swin_param_shapes(synth.SWIN_T_GRPB).items():
        assert tuple(sd["swin_tiny_grpb_backbone." + k].shape) == shp, k
    for k, shp in synth.vqa_head_param_shapes().items():
        assert tuple(sd["swin_tiny_grpb_head." + k].shape) == shp, k
    assert sd["swin_tiny_grpb_backbone.layers.0.blocks.0.attn.relative_position_index"].shape == (392, 392)
    assert "swin_tiny_grpb_backbone.layers.3.blocks.0.attn.fragment_position_bias_table" not in sd
    n = sum(p.numel() for p in net.swin_tiny_grpb_backbone.parameters())
    assert abs(n / 1e6 - 28.08) < 0.01       # 28.08 M parameters
    # DataParallel-prefixed checkpoints (trainer.py:62-74) strip to these names
    net.load_state_dict({k: v for k, v in sd.items()})


def test_model_keys_and_errors():
    tiny = VQA_Network({"model": {"args": {"swin_tiny": {"backbone": {}, "head": {}}}}})
    assert tiny.key_names == ["swin_tiny"] and not tiny.swin_tiny_backbone.frag_biases[0]
    m = VQA_Network({"model": {"args": {"swin_tiny_grpb_m": {"head": {}}}}})
    assert m.swin_tiny_grpb_m_backbone.window_size == (4, 4, 4)
    with pytest.raises(NotImplementedError):
        VQA_Network({"model": {"args": {"unknown_key": {}}}})
    with pytest.raises(NotImplementedError, match="conv_tiny"):
        VQA_Network({"model": {"args": {"conv_tiny": {}}}})
    with pytest.raises(_abi.KvqError, match="no CPU path"):
        tiny(inputs={"technical": torch.zeros(1, 3, 8, 64, 64)})


def test_relative_position_index_buffer_matches_golden(golden):
    from kvq_amd.models.backbones.swin_backbone import _rel_pos_index
    g = golden("layout.npz")
    assert torch.equal(_rel_pos_index((8, 7, 7)), torch.from_numpy(g["rpi_877"].astype("int64")))
