"""ctypes binding of include/kvq_hip.h (libkvq_hip.so).

There is NO fallback: if the library is missing and cannot be built, or a call
fails, this raises.  The product path never routes through PyTorch eager or the
CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import _build

MAX_STAGES = 4
K_NAMES = ["im2col", "layernorm", "gemm_qkv", "attn", "gemm_proj", "gemm_fc1", "gemm_fc2", "gemm_merge",
           "gemm_embed", "tail", "embed", "merge"]
K_COUNT = len(K_NAMES)

EPI_BIAS_BF16, EPI_GELU_BF16, EPI_QKV_BF16, EPI_RESID_F32, EPI_STORE_F32, EPI_RELU_BF16, EPI_QGELU_BF16 = range(7)
DT_BF16, DT_FP16 = 0, 1
ABI_VERSION = 31


def dtype_code(dt) -> int:
    """torch dtype / name -> KvqDtype."""
    import torch
    is_int = isinstance(dt, int) and not isinstance(dt, bool)
    if (is_int and dt == DT_FP16) or (not is_int and dt in (torch.float16, "fp16", "float16")):
        return DT_FP16
    if (is_int and dt == DT_BF16) or (not is_int and dt in (torch.bfloat16, "bf16", "bfloat16")):
        return DT_BF16
    raise ValueError(f"16-bit operand dtype must be fp16 or bf16, got {dt!r}")


def torch_dtype(code: int):
    import torch
    return torch.float16 if code == DT_FP16 else torch.bfloat16

p_void = C.c_void_p


class KvqSwinCfg(C.Structure):
    _fields_ = [("patch", C.c_int32 * 3), ("in_chans", C.c_int32), ("embed_dim", C.c_int32),
                ("num_stages", C.c_int32), ("depths", C.c_int32 * MAX_STAGES),
                ("num_heads", C.c_int32 * MAX_STAGES), ("window", C.c_int32 * 3), ("mlp_ratio", C.c_int32),
                ("frag_bias", C.c_int32 * MAX_STAGES), ("adaptive_window", C.c_int32 * 3)]


class KvqSwinBlockW(C.Structure):
    _fields_ = [(n, p_void) for n in ("norm1_w", "norm1_b", "rpb_table", "fpb_table", "bias_pack", "qkv_w", "qkv_b", "proj_w",
                                      "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "tail_pack", "qkv_pack", "bias_dense")]


class KvqSwinMergeW(C.Structure):
    _fields_ = [(n, p_void) for n in ("norm_w", "norm_b", "red_w", "merge_pack")]


class KvqSwinWeights(C.Structure):
    _fields_ = [("embed_w", p_void), ("embed_b", p_void), ("embed_ln_w", p_void), ("embed_ln_b", p_void),
                ("embed_pack", p_void), ("blocks", C.POINTER(KvqSwinBlockW)), ("merges", KvqSwinMergeW * (MAX_STAGES - 1)),
                ("norm_w", p_void), ("norm_b", p_void)]


class KvqAttnDenseArgs(C.Structure):
    _fields_ = [("qkv", p_void), ("bias_dense", p_void), ("n_types", C.c_int32), ("BW", C.c_int32), ("nW", C.c_int32), ("N", C.c_int32),
                ("num_heads", C.c_int32), ("dtype", C.c_int32), ("out", p_void), ("tile_skip", p_void), ("dsplit_from", C.c_int32),
                ("x_ln", p_void), ("w_qkv", p_void), ("b_qkv", p_void), ("q_scale", C.c_float), ("pad_mask", p_void)]


class KvqGemmArgs(C.Structure):
    _fields_ = [("A", p_void), ("W", p_void), ("bias", p_void), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("epilogue", C.c_int32), ("out_bf16", p_void), ("out_f32", p_void),
                ("num_heads", C.c_int32), ("q_scale", C.c_float), ("scatter_map", p_void),
                ("map_rows", C.c_int32), ("out_rows", C.c_int32), ("dtype", C.c_int32), ("resid_bf16", p_void), ("resid_f32", p_void),
                ("splitk_ws", p_void), ("splitk_ws_bytes", C.c_size_t), ("ldc", C.c_int32), ("col_off", C.c_int32),
                ("a_gather", p_void), ("a_rows", C.c_int32), ("a_phys_rows", C.c_int32)]


class KvqConvArgs(C.Structure):
    _fields_ = [("x", p_void), ("W", p_void), ("bias", p_void), ("taps", p_void), ("dims5", C.c_int32 * 5),
                ("kernel3", C.c_int32 * 3), ("stride3", C.c_int32 * 3), ("pad3", C.c_int32 * 3), ("Kpad", C.c_int32),
                ("N", C.c_int32), ("epilogue", C.c_int32), ("dtype", C.c_int32), ("out_bf16", p_void), ("out_f32", p_void),
                ("resid_bf16", p_void), ("resid_f32", p_void), ("splitk_ws", p_void), ("splitk_ws_bytes", C.c_size_t),
                ("ldc", C.c_int32), ("col_off", C.c_int32)]


class KvqNetTensor(C.Structure):
    _fields_ = [("B", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("kind", C.c_int32)]


class KvqNetOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("src", C.c_int32), ("src2", C.c_int32), ("dst", C.c_int32), ("dst32", C.c_int32),
                ("kernel3", C.c_int32 * 3),
                ("stride3", C.c_int32 * 3), ("pad3", C.c_int32 * 3), ("cout", C.c_int32), ("kpad", C.c_int32), ("relu", C.c_int32),
                ("is_max", C.c_int32), ("dst_coff", C.c_int32), ("per_frame", C.c_int32), ("mean_off", C.c_int32),
                ("std_off", C.c_int32), ("out_stride", C.c_int64), ("w", p_void), ("bias", p_void), ("t_index", p_void),
                ("n_index", C.c_int32), ("lane", C.c_int32)]


NET_CONV, NET_POOL, NET_STEM8, NET_STEM_MFMA, NET_MEAN_STD, NET_SELECT_T, NET_BOTTLENECK, NET_STEM_POOL, NET_STEM64_POOL, NET_BOTTLENECK_S = range(10)
NET_T_ACT16, NET_T_F32_PLANAR, NET_T_ACT32 = 0, 1, 2


class KvqBlockTailArgs(C.Structure):
    _fields_ = [("attn", p_void), ("x", p_void), ("scatter_map", p_void), ("map_rows", C.c_int32),
                ("out_rows", C.c_int32), ("M", C.c_int32), ("C", C.c_int32), ("hidden", C.c_int32), ("pack", p_void),
                ("next_norm_w", p_void), ("next_norm_b", p_void), ("next_dst", p_void), ("next_ln", p_void),
                ("next_rows", C.c_int32), ("eps", C.c_float), ("dtype", C.c_int32), ("attn_gather", p_void),
                ("next_qkv_pack", p_void), ("next_qkv_b", p_void), ("qkv_out", p_void), ("q_scale", C.c_float), ("num_heads", C.c_int32),
                ("x_f16", C.c_int32)]


FRAG_MAX_CLIPS = 16


class KvqFragmentSource(C.Structure):
    _fields_ = [("video", p_void * FRAG_MAX_CLIPS), ("hoff", p_void * FRAG_MAX_CLIPS), ("woff", p_void * FRAG_MAX_CLIPS),
                ("chan_stride", C.c_int64), ("n_clips", C.c_int32), ("src_is_u8", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("Fh", C.c_int32),
                ("Fw", C.c_int32), ("fs_h", C.c_int32), ("fs_w", C.c_int32), ("aligned", C.c_int32), ("normalise", C.c_int32),
                ("mean", C.c_float * 4), ("std", C.c_float * 4), ("indirect", p_void)]


class KvqPatchEmbedArgs(C.Structure):
    _fields_ = [("x", p_void), ("B", C.c_int32), ("in_chans", C.c_int32), ("T", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("pd", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32), ("embed_dim", C.c_int32),
                ("pack", p_void), ("has_norm", C.c_int32), ("out", p_void), ("next_norm_w", p_void),
                ("next_norm_b", p_void), ("next_dst", p_void), ("next_ln", p_void), ("next_rows", C.c_int32),
                ("eps", C.c_float), ("dtype", C.c_int32), ("frag", C.POINTER(KvqFragmentSource)), ("out_f16", C.c_int32)]


class KvqPatchMergeArgs(C.Structure):
    _fields_ = [("x", p_void), ("merge_map", p_void), ("B", C.c_int32), ("L", C.c_int32), ("Ln", C.c_int32), ("C", C.c_int32),
                ("pack", p_void), ("out", p_void), ("next_norm_w", p_void), ("next_norm_b", p_void), ("next_dst", p_void),
                ("next_ln", p_void), ("next_rows", C.c_int32), ("eps", C.c_float), ("dtype", C.c_int32), ("x_f16", C.c_int32), ("out_f16", C.c_int32)]


class KvqProfRecord(C.Structure):
    _fields_ = [("kind", C.c_int32), ("variant", C.c_int32), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


# every symbol include/kvq_hip.h declares: name -> (restype, argtypes)
i32, i64, f32, sz = C.c_int32, C.c_int64, C.c_float, C.c_size_t
SYMBOLS = {
    "kvq_abi_version": (i32, []),
    "kvq_last_error": (C.c_char_p, []),
    "kvq_device_name": (i32, [C.c_char_p, i32]),
    "kvq_swin3d_plan_create": (i32, [C.POINTER(KvqSwinCfg), i32, i32, i32, i32, i32, C.POINTER(p_void)]),
    "kvq_swin3d_plan_destroy": (None, [p_void]),
    "kvq_swin3d_workspace_bytes": (sz, [p_void]),
    "kvq_swin3d_out_dims": (i32, [p_void, C.POINTER(i32 * 4)]),
    "kvq_vit_embed_ln": (i32, [p_void, p_void, p_void, p_void, p_void, i32, i32, i32, f32, p_void, p_void]),
    "kvq_mha_small": (i32, [p_void, i32, i32, i32, i32, i32, p_void, p_void]),
    "kvq_mha_cross": (i32, [p_void, i64, p_void, i64, p_void, i64, i32, i32, i32, i32, i32, f32, i32, p_void, p_void]),
    "kvq_cls_gather": (i32, [p_void, i32, i32, i32, i32, p_void, p_void]),
    "kvq_cls_mix": (i32, [p_void, p_void, i32, i32, i32, f32, i32, p_void]),
    "kvq_cosine_cls": (i32, [p_void, i32, i32, i32, p_void, p_void]),
    "kvq_convert": (i32, [p_void, p_void, i64, i32, i32, p_void]),
    "kvq_sem_modulate": (i32, [p_void, p_void, p_void, f32, p_void, f32, i32, i32, p_void, p_void]),
    "kvq_dist_modulate": (i32, [p_void, p_void, p_void, i32, i32, i32, i32, p_void, p_void]),
    "kvq_qrs_top_region": (i32, [p_void, i32, i32, i32, i32, i32, i32, p_void, p_void]),
    "kvq_crop_regions": (i32, [p_void, p_void, i32, i32, i32, i32, i32, i32, i32, i32, p_void, p_void]),
    "kvq_l2_normalize_rows": (i32, [p_void, i32, i32, i32, p_void, p_void]),
    "kvq_axpby": (i32, [p_void, p_void, f32, f32, p_void, i64, p_void]),
    "kvq_conv_implicit": (i32, [C.POINTER(KvqConvArgs), p_void]),
    "kvq_convnet_create": (i32, [C.POINTER(KvqNetOp), i32, C.POINTER(KvqNetTensor), i32, i32, i32, i32, C.POINTER(p_void)]),
    "kvq_convnet_destroy": (None, [p_void]),
    "kvq_convnet_splitk": (i32, [p_void, i32]),
    "kvq_convnet_workspace_bytes": (sz, [p_void]),
    "kvq_convnet_forward": (i32, [p_void, C.POINTER(p_void), C.POINTER(p_void), p_void, sz, p_void]),
    "kvq_qkv_fill_pad": (i32, [p_void, p_void, p_void, i32, i32, i32, i32, C.c_float, i32, p_void]),
    "kvq_fast_bottleneck_pack_bytes": (sz, [i32, i32, i32, i32, i32]),
    "kvq_fast_bottleneck": (i32, [p_void, C.POINTER(i32), i32, i32, i32, i32, i32, p_void, i32, p_void, p_void]),
    "kvq_convnet_profile": (i32, [p_void, i32]),
    "kvq_convnet_profile_read": (i32, [p_void, C.POINTER(C.c_float), i32, C.POINTER(i32)]),
    "kvq_swin3d_forward_stages": (i32, [p_void, p_void, p_void, i32, i32, p_void, p_void, p_void, sz, p_void]),
    "kvq_swin3d_set_taps": (i32, [p_void, C.POINTER(p_void)]),
    "kvq_swin3d_tap_dims": (i32, [p_void, i32, C.POINTER(i32 * 4)]),
    "kvq_resize_trilinear_cl": (i32, [p_void, i32, i32, i32, i32, i32, p_void, i32, i32, i32, i32, i32, p_void]),
    "kvq_swin3d_forward": (i32, [p_void, C.POINTER(KvqSwinWeights), p_void, p_void, p_void, sz, p_void]),
    "kvq_swin3d_forward_fragments": (i32, [p_void, C.POINTER(KvqSwinWeights), C.POINTER(KvqFragmentSource), p_void, p_void, sz,
                                           p_void]),
    "kvq_swin3d_profile": (i32, [p_void, i32]),
    "kvq_swin3d_profile_read": (i32, [p_void, C.POINTER(KvqProfRecord), i32, C.POINTER(i32)]),
    "kvq_layernorm_rows": (i32, [p_void, p_void, i32, i32, i32, i32, i32, p_void, p_void, f32, p_void, i32, p_void,
                                 p_void]),
    "kvq_gemm_bf16": (i32, [C.POINTER(KvqGemmArgs), p_void]),
    "kvq_gemm_splitk_factor": (i32, [i32, i32, i32]),
    "kvq_gemm_splitk_bytes": (sz, [i32, i32, i32]),
    "kvq_debug_gemm_trace": (i32, [p_void, i32]),
    "kvq_gemm_tile_mode": (i32, [i32]),
    "kvq_patch_embed_supported": (i32, [i32] * 8),
    "kvq_patch_embed_fragments_supported": (i32, [C.POINTER(KvqFragmentSource), i32, i32, i32, i32, i32, i32]),
    "kvq_patch_embed_pack_bytes": (sz, [i32, i32]),
    "kvq_patch_embed_pack": (i32, [p_void, p_void, p_void, p_void, i32, i32, p_void, p_void]),
    "kvq_patch_embed": (i32, [C.POINTER(KvqPatchEmbedArgs), p_void]),
    "kvq_patch_merge_supported": (i32, [i32]),
    "kvq_patch_merge_pack_bytes": (sz, [i32]),
    "kvq_patch_merge_pack": (i32, [p_void, p_void, p_void, i32, i32, p_void, p_void]),
    "kvq_patch_merge": (i32, [C.POINTER(KvqPatchMergeArgs), p_void]),
    "kvq_block_tail_supported": (i32, [i32, i32]),
    "kvq_block_tail_pack_bytes": (sz, [i32, i32]),
    "kvq_block_tail_pack": (i32, [p_void, p_void, p_void, p_void, p_void, p_void, p_void, p_void, i32, i32, p_void,
                                  p_void]),
    "kvq_block_tail_qkv_pack_bytes": (sz, [i32, i32]),
    "kvq_block_tail_qkv_pack": (i32, [p_void, i32, i32, p_void, p_void]),
    "kvq_block_tail": (i32, [C.POINTER(KvqBlockTailArgs), p_void]),
    "kvq_window_attention": (i32, [p_void, p_void, p_void, p_void, p_void, i32, i32, i32, i32, i32, i32, i32, i32, p_void,
                                   p_void]),
    "kvq_attn_bias32_bytes": (sz, [i32, i32, i32]),
    "kvq_attn_bias32_build": (i32, [p_void, p_void, p_void, i32, i32, i32, i32, i32, i32, p_void, p_void, p_void]),
    "kvq_window_attention32": (i32, [C.POINTER(KvqAttnDenseArgs), p_void]),
    "kvq_swin3d_bias_dense_bytes": (sz, [p_void, i32]),
    "kvq_swin3d_bias_dense_build": (i32, [p_void, i32, p_void, p_void, p_void, p_void, p_void]),
    "kvq_patch_im2col": (i32, [p_void, i32, i32, i32, i32, i32, i32, i32, i32, i32, p_void, p_void]),
    "kvq_vqa_head": (i32, [p_void, i32, i32, i32, i64, i64, i64, p_void, p_void, p_void, i32, p_void, p_void, p_void,
                           p_void, p_void]),
    "kvq_vqa_head_classes": (i32, [p_void, i32, i32, i32, i64, i64, i64, p_void, p_void, i32, p_void, p_void, i32, i32, p_void,
                                   p_void, p_void]),
    "kvq_simple_vqa_head": (i32, [p_void, i32, i32, i32, p_void, p_void, i32, p_void, p_void, p_void, p_void,
                                  p_void]),
    "kvq_resize_bilinear": (i32, [p_void, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(f32),
                                  C.POINTER(f32), p_void, p_void]),
    "kvq_upsample_frames_out_dims": (i32, [i32, i32, C.c_double, C.POINTER(i32 * 2)]),
    "kvq_upsample_frames": (i32, [p_void, i32, i32, i32, i32, i32, C.c_double, p_void, p_void]),
    "kvq_im2col_nd": (i32, [p_void, i32, i32, C.POINTER(i64 * 5), C.POINTER(i32 * 5), C.POINTER(i32 * 3),
                            C.POINTER(i32 * 3), C.POINTER(i32 * 3), i32, p_void, p_void]),
    "kvq_pack_channels_last8": (i32, [p_void, C.POINTER(i32 * 5), C.POINTER(i64 * 5), i32, p_void, p_void]),
    "kvq_conv_stem_direct": (i32, [p_void, C.POINTER(i32 * 5), p_void, p_void, i32, C.POINTER(i32 * 3), C.POINTER(i32 * 3),
                                   C.POINTER(i32 * 3), i32, i32, p_void, p_void]),
    "kvq_pack_clip_cl4": (i32, [p_void, C.POINTER(i32 * 5), i32, i32, p_void, p_void]),
    "kvq_conv_stem_pool": (i32, [p_void, C.POINTER(i32 * 5), p_void, p_void, i32, i32, i32, p_void, p_void]),
    "kvq_conv_stem64_pool": (i32, [p_void, C.POINTER(i32 * 5), p_void, i32, p_void, p_void, i32, i32, p_void, i32, i32, p_void]),
    "kvq_slow_bottleneck_pack_bytes": (C.c_size_t, [i32, i32, i32]),
    "kvq_slow_bottleneck": (i32, [p_void, C.POINTER(i32 * 4), i32, i32, i32, p_void, i32, p_void, i32, p_void]),
    "kvq_conv_stem_mfma": (i32, [p_void, C.POINTER(i32 * 4), p_void, p_void, C.POINTER(i32 * 3), C.POINTER(i32 * 3), C.POINTER(i32 * 3),
                                 i32, i32, p_void, p_void]),
    "kvq_pool_nd": (i32, [p_void, i32, C.POINTER(i32 * 5), C.POINTER(i32 * 3), C.POINTER(i32 * 3), C.POINTER(i32 * 3),
                          i32, p_void, p_void]),
    "kvq_pool_nd_strided": (i32, [p_void, i32, C.POINTER(i32 * 5), C.POINTER(i32 * 3), C.POINTER(i32 * 3), C.POINTER(i32 * 3),
                                  i32, p_void, i32, i32, p_void]),
    "kvq_mean_std_pool": (i32, [p_void, i32, i32, i32, i32, p_void, i64, i32, i32, p_void]),
    "kvq_fragment_gather_batch": (i32, [C.POINTER(KvqFragmentSource), i32, i32, p_void, p_void]),
    "kvq_fragment_gather": (i32, [p_void, i32, i32, i32, i32, i32, p_void, p_void, i32, i32, i32, i32, i32,
                                  C.POINTER(f32), C.POINTER(f32), p_void, p_void]),
}

_lib: Optional[C.CDLL] = None
_lib_lock = __import__("threading").Lock()      # lib() is entered from the prefetch thread too


class KvqError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (building first if missing/stale) libkvq_hip.so.  Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        return _load()


def _load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own libamdhip64; import it FIRST so that this process has exactly one HIP
    # runtime (ours resolves to the already-loaded soname).  Loading the system runtime first leaves
    # two runtimes in the process, and streams / device pointers are not interchangeable between them.
    import torch  # noqa: F401
    path = _build.LIB
    if _build.is_stale():
        try:
            path = _build.build()
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(_build.LIB):
                raise KvqError(f"libkvq_hip.so is missing and could not be built: {e}") from e
            path = _build.LIB   # stale but present (e.g. no hipcc on this box): use it
    handle = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if handle.kvq_abi_version() != ABI_VERSION:
        raise KvqError("libkvq_hip.so ABI version mismatch")
    _lib = handle
    return handle


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().kvq_last_error().decode("utf-8", "replace")
        # the reference raises AssertionError for this one (fusion_datasets.py:60)
        if "Please provide match vclip and align index" in msg:
            raise AssertionError(msg)
        raise KvqError(f"{what} failed (status {rc}): {msg}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def stream_of(t) -> int:
    """The launch stream for work on tensor ``t``: the CURRENT stream of the CURRENT device, which must be the tensor's
    device — a pointer of another GPU handed to this device's stream would be a cross-device launch with no ordering
    against the copies that produced it (rank r of a multi-GPU job must keep its data on cuda:r)."""
    import torch
    cur = torch.cuda.current_device()
    if not t.is_cuda or t.device.index != cur:
        raise KvqError(f"tensor lives on {t.device} but the current HIP device is cuda:{cur}: "
                       "torch.cuda.set_device() to the tensor's device (or move the tensor) before calling the kernels")
    return torch.cuda.current_stream().cuda_stream
