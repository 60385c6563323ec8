"""Thin per-kernel wrappers over the C-ABI (torch tensors in, torch tensors out).

Used by the parity tests (each kernel against the oracle) and by the modules.  PyTorch is
only the owner of device memory and the provider of the current HIP stream here.  The 16-bit
MFMA operand type (fp16 | bf16) is taken from the tensors' dtype.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _abi
from ._abi import check, current_stream, dtype_code, lib, ptr, stream_of

HALF_TYPES = (torch.float16, torch.bfloat16)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _abi.KvqError("kvq_amd kernels need tensors on a HIP device (no CPU fallback exists)")


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(lib().kvq_device_name(buf, 256), "kvq_device_name")
    return buf.value.decode()


def layernorm_rows(x: torch.Tensor, gamma, beta, *, index_map: Optional[torch.Tensor] = None, nparts=1,
                   n_batch=1, rows_out: Optional[int] = None, out_dtype=torch.float16, eps=1e-5):
    """x fp32 [n_batch*rows_in, Cin]; index_map int32 [rows_out, nparts] (or None = identity).
    out_dtype: torch.float16 / torch.bfloat16 (MFMA operand) or torch.float32."""
    _need_gpu(x, gamma, beta, index_map)
    assert x.dtype == torch.float32 and x.is_contiguous()
    rows_in = x.shape[0] // n_batch
    Cin = x.shape[1]
    rows_out = rows_in if rows_out is None else rows_out
    out = torch.empty(n_batch * rows_out, nparts * Cin, dtype=out_dtype, device=x.device)
    half = out_dtype in HALF_TYPES
    check(lib().kvq_layernorm_rows(ptr(x), ptr(index_map), nparts, n_batch, rows_in, rows_out, Cin, ptr(gamma),
                                   ptr(beta), eps, ptr(out) if half else None,
                                   dtype_code(out_dtype) if half else 0, None if half else ptr(out),
                                   current_stream()), "kvq_layernorm_rows")
    return out


def _splitk_scratch(a, M: int, N: int, K: int, device):
    """Split-K scratch for a GEMM / implicit-conv launch (long K, few output tiles): the library says how much it would use;
    the buffer comes from torch's stream-ordered allocator (reused only by later work of this stream), so it may be dropped
    as soon as the launch is enqueued.  Returned so that the caller keeps it alive across the call."""
    nb = lib().kvq_gemm_splitk_bytes(M, N, K)
    if not nb:
        return None
    ws = torch.empty(nb, dtype=torch.uint8, device=device)
    a.splitk_ws, a.splitk_ws_bytes = ptr(ws), nb
    return ws


def gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], epilogue: int, *, out=None,
         num_heads=0, q_scale=1.0, scatter_map=None, map_rows=0, out_rows=0, a_gather=None, a_rows=0, rows=None):
    """A [M,K], W [N,K], both fp16 or both bf16.  Returns the output tensor (allocated unless given).
    ``a_gather`` (+ ``a_rows``, ``rows`` = GEMM rows): row m reads A row (m // a_rows) * (A rows per batch) + a_gather[m % a_rows].
    QKV epilogue with ``scatter_map``: GEMM row m lands in row (m // map_rows) * out_rows + scatter_map[m % map_rows] of a buffer
    with out_rows rows per batch element (the other rows are ``qkv_fill_pad``'s)."""
    _need_gpu(A, W, bias, out, scatter_map, a_gather)
    assert A.dtype in HALF_TYPES and W.dtype == A.dtype and A.is_contiguous() and W.is_contiguous()
    M, K = A.shape
    a_phys = 0
    if a_gather is not None:
        a_phys = A.shape[0] // (rows // a_rows)
        M = rows
    N = W.shape[0]
    if out is None:
        if epilogue == _abi.EPI_QKV_BF16:
            out = torch.empty(3, num_heads, M if scatter_map is None else M // map_rows * out_rows, 32, dtype=A.dtype, device=A.device)
        elif epilogue in (_abi.EPI_BIAS_BF16, _abi.EPI_GELU_BF16, _abi.EPI_QGELU_BF16):
            out = torch.empty(M, N, dtype=A.dtype, device=A.device)
        elif epilogue == _abi.EPI_STORE_F32:
            out = torch.empty(M, N, dtype=torch.float32, device=A.device)
        else:
            raise ValueError("RESID epilogue accumulates into an existing tensor: pass out=")
    a = _abi.KvqGemmArgs()
    a.A, a.W, a.bias, a.M, a.N, a.K, a.epilogue = ptr(A), ptr(W), ptr(bias), M, N, K, epilogue
    if out.dtype in HALF_TYPES:
        assert out.dtype == A.dtype
        a.out_bf16 = ptr(out)
    else:
        a.out_f32 = ptr(out)
    a.num_heads, a.q_scale = num_heads, q_scale
    a.scatter_map, a.map_rows, a.out_rows = ptr(scatter_map), map_rows, out_rows
    a.a_gather, a.a_rows, a.a_phys_rows = ptr(a_gather), a_rows, a_phys
    a.dtype = dtype_code(A.dtype)
    _ws = _splitk_scratch(a, a.M, a.N, a.K, A.device) if (a.epilogue != _abi.EPI_QKV_BF16 and a_gather is None) else None   # noqa: F841
    check(lib().kvq_gemm_bf16(C.byref(a), current_stream()), "kvq_gemm_bf16")
    return out


def gemm_tile_mode(mode: int) -> int:
    """-1: main loop by shape (default), 0: never the 256 x 256 eight-phase tile, 1: whenever eligible.  Returns the previous mode."""
    return lib().kvq_gemm_tile_mode(mode)


def qkv_fill_pad(qkv: torch.Tensor, qkv_bias: torch.Tensor, pad_rows: torch.Tensor, n_batch: int, q_scale: float):
    """q | k | v = bias (q scaled) in the padding rows ``pad_rows`` (int32, window order) of every batch element of the head-major
    buffer qkv [3, nH, n_batch * rows, 32] — what the reference computes for rows padded after norm1."""
    _need_gpu(qkv, qkv_bias, pad_rows)
    assert qkv.dtype in HALF_TYPES and qkv.is_contiguous() and pad_rows.dtype == torch.int32 and qkv_bias.dtype == torch.float32
    check(lib().kvq_qkv_fill_pad(ptr(qkv), ptr(qkv_bias), ptr(pad_rows), pad_rows.numel(), n_batch, qkv.shape[2] // n_batch, qkv.shape[1],
                                 q_scale, dtype_code(qkv.dtype), stream_of(qkv)), "kvq_qkv_fill_pad")
    return qkv


def window_attention(qkv: torch.Tensor, tok: torch.Tensor, rpb: torch.Tensor, fpb: Optional[torch.Tensor],
                     center: int, nW: int, N: int, use_mask: bool, bias_pack: Optional[torch.Tensor] = None):
    """qkv fp16|bf16 [3,nH,BW*N,32] (q pre-scaled), tok int32 [nW*N,2]; returns [BW*N, nH*32]."""
    _need_gpu(qkv, tok, rpb, fpb)
    assert qkv.dtype in HALF_TYPES and qkv.is_contiguous()
    nH = qkv.shape[1]
    BW = qkv.shape[2] // N
    out = torch.empty(BW * N, nH * 32, dtype=qkv.dtype, device=qkv.device)
    check(lib().kvq_window_attention(ptr(qkv), ptr(tok), ptr(rpb), ptr(fpb), ptr(bias_pack), rpb.shape[0], center, BW, nW, N, nH,
                                     int(use_mask), dtype_code(qkv.dtype), ptr(out), current_stream()),
          "kvq_window_attention")
    return out


LOG2E = 1.4426950408889634


def attn_bias32(tok: torch.Tensor, rpb: torch.Tensor, fpb: Optional[torch.Tensor], center: int, nW: int, N: int, use_mask: bool):
    """The attention bias image of one block for ``window_attention32`` (include/kvq_hip.h); ``nW`` = the number of window TYPES
    to build (the first nW windows' descriptors of ``tok``)."""
    _need_gpu(tok, rpb, fpb)
    nH = rpb.shape[1]
    out = torch.empty(lib().kvq_attn_bias32_bytes(nW, N, nH), dtype=torch.uint8, device=rpb.device)
    big = torch.zeros(1, dtype=torch.float32, device=rpb.device)
    check(lib().kvq_attn_bias32_build(ptr(tok), ptr(rpb), ptr(fpb), rpb.shape[0], center, nW, N, nH, int(use_mask),
                                      ptr(out), ptr(big), current_stream()), "kvq_attn_bias32_build")
    out.max_abs_bias = big          # device scalar: largest un-masked |bias| (fp16 storage rounds by 2^-11 of it)
    return out


def window_attention32(qkv: torch.Tensor, image: torch.Tensor, nW: int, N: int, n_types: Optional[int] = None,
                       tile_skip: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, dsplit_from: int = -1,
                       x_ln: Optional[torch.Tensor] = None, w_qkv: Optional[torch.Tensor] = None, b_qkv: Optional[torch.Tensor] = None,
                       q_scale: float = 1.0, pad_mask: Optional[torch.Tensor] = None):
    """qkv fp16|bf16 [3,nH,BW*N,32] with q scaled by head_dim^-0.5 * log2(e) + the pre-built bias image of ``attn_bias32``; returns
    [BW*N, nH*32].  ``n_types`` (default nW): window w uses bias w % n_types.  ``tile_skip`` int32 [nW]: bit t = rows 16t..16t+15 of the
    window are padding only (a 32-row q-block is passed over when both its tiles are; such rows keep what ``out`` held).
    ``dsplit_from`` >= 0: windows >= it are depth-split (shifted (8,7,7) blocks).  ``x_ln`` [BW*N, C] + ``w_qkv`` [3C, C] + ``b_qkv`` [3C]:
    the launch computes q | k | v itself (``qkv`` = a [1|3, nH, BW*N, 32] buffer whose first third receives q; ``q_scale`` =
    head_dim^-0.5 * log2(e)).  ``pad_mask`` int32 [nW, 13] (+ ``b_qkv``): bit r of window w = row r is a padding row — its k | v are
    written by the kernel (= the bias), its q taken as zero; those rows of ``qkv`` are not read."""
    _need_gpu(qkv, image, tile_skip, out, pad_mask)
    assert qkv.dtype in HALF_TYPES and qkv.is_contiguous()
    nH = qkv.shape[1]
    BW = qkv.shape[2] // N
    if out is None:
        out = torch.empty(BW * N, nH * 32, dtype=qkv.dtype, device=qkv.device)
    a = _abi.KvqAttnDenseArgs()
    a.qkv, a.bias_dense, a.n_types, a.BW, a.nW, a.N, a.num_heads = ptr(qkv), ptr(image), nW if n_types is None else n_types, BW, nW, N, nH
    a.dtype, a.out, a.tile_skip, a.dsplit_from = dtype_code(qkv.dtype), ptr(out), ptr(tile_skip), dsplit_from
    if x_ln is not None:      # fused qkv projection: ``qkv`` only lends its q third as scratch ([nH, BW*N, 32] is enough)
        _need_gpu(x_ln, w_qkv, b_qkv)
        assert x_ln.dtype == qkv.dtype == w_qkv.dtype and x_ln.is_contiguous() and w_qkv.is_contiguous() and b_qkv.dtype == torch.float32
        a.x_ln, a.w_qkv, a.b_qkv, a.q_scale = ptr(x_ln), ptr(w_qkv), ptr(b_qkv), q_scale
    if pad_mask is not None:
        _need_gpu(b_qkv)
        assert pad_mask.dtype == torch.int32 and pad_mask.is_contiguous() and (b_qkv is None or b_qkv.dtype == torch.float32)
        a.pad_mask, a.b_qkv = ptr(pad_mask), ptr(b_qkv)
    check(lib().kvq_window_attention32(C.byref(a), current_stream()), "kvq_window_attention32")
    return out


def patch_im2col(x: torch.Tensor, patch: Sequence[int], out_dtype=torch.float16):
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, Cin, T, H, W = x.shape
    pd, ph, pw = patch
    D, Hh, Ww = -(-T // pd), -(-H // ph), -(-W // pw)
    out = torch.empty(B * D * Hh * Ww, Cin * pd * ph * pw, dtype=out_dtype, device=x.device)
    check(lib().kvq_patch_im2col(ptr(x), B, Cin, T, H, W, pd, ph, pw, dtype_code(out_dtype), ptr(out),
                                 current_stream()), "kvq_patch_im2col")
    return out


def vqa_head(feat: torch.Tensor, w1, b1, w2, b2, w1t=None):
    """feat fp32 (B,C,D,H,W) with ANY strides over a dense token grid -> score fp32 [B,1].
    ``w1`` [hidden,C] (what the fp32-MFMA kernel reads: hidden == 64, channels-last features) and / or ``w1t`` [C,hidden]
    (the VALU kernel's layout; made here from ``w1`` when missing)."""
    _need_gpu(feat, w1, b1, w2, b2, w1t)
    assert feat.dtype == torch.float32 and feat.dim() == 5
    B, Cc, D, H, W = feat.shape
    L = D * H * W
    sb, sc, sd, sh, sw = feat.stride()
    if not (sh == W * sw and sd == H * sh):          # tokens must be addressable with one stride
        feat = feat.contiguous()
        sb, sc, sd, sh, sw = feat.stride()
    if w1 is not None:
        w1 = w1.reshape(w1.shape[0], -1).contiguous()
    if w1t is None:
        w1t = w1.t().contiguous()               # [C][hidden]: what the VALU kernel streams (kvq_hip.h)
    hidden = w1t.shape[1]
    scratch = torch.empty(B * L, dtype=torch.float32, device=feat.device)
    score = torch.empty(B, dtype=torch.float32, device=feat.device)
    check(lib().kvq_vqa_head(ptr(feat), B, L, Cc, sb, sw, sc, ptr(w1t), ptr(w1), ptr(b1), hidden, ptr(w2), ptr(b2),
                             ptr(scratch), ptr(score), current_stream()), "kvq_vqa_head")
    return score.reshape(B, 1)


def vqa_head_classes(feat: torch.Tensor, w1, b1, w2, b2, pre_pool=False):
    """VQAHead's pre_pool / num_class > 1 branches (head.py:60-68): feat fp32 (B,C,D,H,W) with any strides over a dense token grid,
    ``w1`` [hidden,C], ``w2`` [num_class,hidden], ``b2`` [num_class] -> fp32 [B,num_class] (softmax over the classes per token when
    num_class > 1, then the mean over the tokens; ``pre_pool``: the token grid is averaged first)."""
    _need_gpu(feat, w1, b1, w2, b2)
    assert feat.dtype == torch.float32 and feat.dim() == 5
    B, Cc, D, H, W = feat.shape
    L = D * H * W
    sb, sc, sd, sh, sw = feat.stride()
    if not (sh == W * sw and sd == H * sh):
        feat = feat.contiguous()
        sb, sc, sd, sh, sw = feat.stride()
    w1t = w1.reshape(w1.shape[0], -1).t().contiguous()
    hidden = w1t.shape[1]
    w2 = w2.reshape(w2.shape[0], -1).contiguous()
    K = w2.shape[0]
    assert w2.shape[1] == hidden and b2.numel() == K and w1t.shape[0] == Cc
    scratch = torch.empty(B * Cc + B * K if pre_pool else B * L * K, dtype=torch.float32, device=feat.device)
    score = torch.empty(B, K, dtype=torch.float32, device=feat.device)
    check(lib().kvq_vqa_head_classes(ptr(feat), B, L, Cc, sb, sw, sc, ptr(w1t), ptr(b1), hidden, ptr(w2), ptr(b2), K,
                                     1 if pre_pool else 0, ptr(scratch), ptr(score), current_stream()), "kvq_vqa_head_classes")
    return score


def simple_vqa_head(feat: torch.Tensor, w1, b1, w2, b2):
    _need_gpu(feat, w1, b1, w2, b2)
    assert feat.dtype == torch.float32
    feat = feat.contiguous()
    B, T, Cin = feat.shape
    scratch = torch.empty(B * T, dtype=torch.float32, device=feat.device)
    score = torch.empty(B, dtype=torch.float32, device=feat.device)
    check(lib().kvq_simple_vqa_head(ptr(feat), B, T, Cin, ptr(w1), ptr(b1), w1.shape[0], ptr(w2), ptr(b2),
                                    ptr(scratch), ptr(score), current_stream()), "kvq_simple_vqa_head")
    return score.reshape(B, 1)


def fragment_gather(video: torch.Tensor, hoff: torch.Tensor, woff: torch.Tensor, fragments_h, fragments_w,
                    fsize_h, fsize_w, aligned, mean=None, std=None, out=None):
    """video uint8/fp32 (C,T,H,W) on device; hoff/woff int32 (Fh,Fw,T//aligned) ABSOLUTE patch origins.
    ``out``: optional fp32 (C,T,Fh*fs,Fw*fs) destination (e.g. one clip of a batch tensor) — no allocation, no copy."""
    _need_gpu(video, hoff, woff)
    assert video.dtype in (torch.uint8, torch.float32) and video.is_contiguous()
    Cc, T, H, W = video.shape
    shape = (Cc, T, fragments_h * fsize_h, fragments_w * fsize_w)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=video.device)
    else:
        assert tuple(out.shape) == shape and out.dtype == torch.float32 and out.is_contiguous() and out.device == video.device
    m = (C.c_float * Cc)(*mean) if mean is not None else None
    s = (C.c_float * Cc)(*std) if std is not None else None
    hoff, woff = hoff.contiguous(), woff.contiguous()
    check(lib().kvq_fragment_gather(ptr(video), int(video.dtype == torch.uint8), Cc, T, H, W, ptr(hoff), ptr(woff),
                                    fragments_h, fragments_w, fsize_h, fsize_w, aligned, m, s, ptr(out),
                                    stream_of(video)), "kvq_fragment_gather")
    return out


class FragmentSource:
    """A batch of technical-branch clips that is still (decoded uint8 frames, sampler draws): the arguments ``fragment_gather``
    would get, clip by clip.  ``SwinTransformer3D.forward`` reads its patch-embedding operand straight out of it (K1 fused
    into the embedding launch: the fp32 clip is neither written nor read back); ``materialise()`` is the two-step form, for
    consumers that want the tensor.  Bit-identical either way.

    videos: uint8 (C,T,Hs,Ws) device tensors of one shape — contiguous, or runs of frames of a longer video (frames
    contiguous, one common channel stride: what ``split_clips`` makes); hoffs / woffs: int32 (Fh,Fw,T//aligned) device tensors
    of ABSOLUTE patch origins (as ``fragment_gather``)."""

    def __init__(self, videos, hoffs, woffs, fragments_h, fragments_w, fsize_h, fsize_w, aligned, mean=None, std=None):
        videos, hoffs, woffs = list(videos), [h.contiguous() for h in hoffs], [w.contiguous() for w in woffs]
        assert len(videos) == len(hoffs) == len(woffs) and 0 < len(videos)
        _need_gpu(*videos, *hoffs, *woffs)
        v0 = videos[0]
        Cc, T, Hs, Ws = v0.shape
        nt = T // aligned
        for v, h, w in zip(videos, hoffs, woffs):
            assert v.shape == v0.shape and v.dtype == v0.dtype and v.device == v0.device
            assert v.stride()[1:] == (Hs * Ws, Ws, 1) and v.stride(0) == v0.stride(0) >= T * Hs * Ws, "frames must be contiguous"
            assert h.dtype == w.dtype == torch.int32 and tuple(h.shape) == tuple(w.shape) == (fragments_h, fragments_w, nt)
        assert (mean is None) == (std is None)
        self.videos, self.hoffs, self.woffs = videos, hoffs, woffs
        self.geometry = (fragments_h, fragments_w, fsize_h, fsize_w, aligned)
        self.mean, self.std = mean, std
        self.device, self.is_cuda, self.dtype = v0.device, True, torch.float32
        self.shape = (len(videos), Cc, T, fragments_h * fsize_h, fragments_w * fsize_w)
        self._c = None

    @staticmethod
    def cat(sources):
        """the batch of several sources (same geometry / normalisation), in order"""
        s0 = sources[0]
        assert all(s.geometry == s0.geometry and s.mean == s0.mean and s.std == s0.std for s in sources)
        return FragmentSource([v for s in sources for v in s.videos], [h for s in sources for h in s.hoffs],
                              [w for s in sources for w in s.woffs], *s0.geometry, mean=s0.mean, std=s0.std)

    def split_clips(self, num_clips):
        """every T-frame entry as ``num_clips`` clips of T/num_clips consecutive frames — the harness's clip reshape
        (trainer.py:306-319: (b,c,nc*t,h,w) -> (b*nc,c,t,h,w)) without moving a byte: the clips are views of the frames."""
        if num_clips == 1:
            return self
        _, _, T, _, _ = self.shape
        aligned = self.geometry[4]
        assert T % num_clips == 0 and (T // num_clips) % aligned == 0, "a clip must hold whole aligned frame groups"
        t, nt = T // num_clips, T // num_clips // aligned
        vs, hs, ws = [], [], []
        for v, h, w in zip(self.videos, self.hoffs, self.woffs):
            for k in range(num_clips):
                vs.append(v[:, k * t:(k + 1) * t])
                hs.append(h[:, :, k * nt:(k + 1) * nt])
                ws.append(w[:, :, k * nt:(k + 1) * nt])
        return FragmentSource(vs, hs, ws, *self.geometry, mean=self.mean, std=self.std)

    def record_stream(self, stream):
        for t in self.videos + self.hoffs + self.woffs:
            t.record_stream(stream)

    def c_struct(self, any_dtype=False):
        """KvqFragmentSource, or None when the batch has more clips than the struct holds / the frames are not uint8 (the fused
        read; ``any_dtype``: fp32 frames too — the batched gather takes them)."""
        v0 = self.videos[0]
        if len(self.videos) > _abi.FRAG_MAX_CLIPS or (v0.dtype != torch.uint8 and not any_dtype):
            return None
        if self._c is not None:
            return self._c
        f = _abi.KvqFragmentSource()
        for i, (v, h, w) in enumerate(zip(self.videos, self.hoffs, self.woffs)):
            f.video[i], f.hoff[i], f.woff[i] = ptr(v), ptr(h), ptr(w)
        f.chan_stride = v0.stride(0)
        f.n_clips, f.src_is_u8, f.Hs, f.Ws = len(self.videos), int(v0.dtype == torch.uint8), v0.shape[2], v0.shape[3]
        f.Fh, f.Fw, f.fs_h, f.fs_w, f.aligned = self.geometry
        f.normalise = int(self.mean is not None)
        if self.mean is not None:
            for c in range(v0.shape[0]):
                f.mean[c], f.std[c] = self.mean[c], self.std[c]
        self._c = f
        return f

    def pointer_table(self):
        """the batch's device pointers as the table ``KvqFragmentSource.indirect`` names: int64 [3 * 16] = video | hoff | woff,
        on the device (made once; what a ``FragmentSlot`` is loaded from)"""
        if getattr(self, "_table", None) is None:
            n = _abi.FRAG_MAX_CLIPS
            assert len(self.videos) <= n
            host = torch.zeros(3 * n, dtype=torch.int64)
            for i, (v, h, w) in enumerate(zip(self.videos, self.hoffs, self.woffs)):
                host[i], host[n + i], host[2 * n + i] = ptr(v), ptr(h), ptr(w)
            self._table = host.to(self.device)
        return self._table

    def materialise(self, out=None):
        """the fp32 (B,C,T,H,W) batch: ``kvq_fragment_gather_batch`` — one launch for up to 16 clips (views included), else one
        ``fragment_gather`` per clip"""
        if out is None:
            out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        assert tuple(out.shape) == tuple(self.shape) and out.dtype == torch.float32 and out.is_contiguous()
        f = self.c_struct(any_dtype=True)
        if f is not None:
            check(lib().kvq_fragment_gather_batch(C.byref(f), self.shape[1], self.shape[2], ptr(out), current_stream()),
                  "kvq_fragment_gather_batch")
            return out
        for b, (v, h, w) in enumerate(zip(self.videos, self.hoffs, self.woffs)):
            fragment_gather(v.contiguous(), h, w, *self.geometry, mean=self.mean, std=self.std, out=out[b])
        return out


class FragmentSlot(FragmentSource):
    """A ``FragmentSource`` whose per-video addresses live in a 384-byte DEVICE table instead of the launch parameters
    (``KvqFragmentSource.indirect``): a forward recorded into a hipGraph through a slot serves every later batch of the same
    geometry — ``load(source)`` rewrites the table on the current stream (one small device-to-device copy) in front of the replay.
    Same kernels, same arithmetic: scores are bit-identical to the forward on the source itself."""

    def __init__(self, source: FragmentSource):
        if not (type(source) is FragmentSource and source.c_struct() is not None):
            raise ValueError("FragmentSlot: uint8 frames, at most 16 clips")
        self.geometry, self.mean, self.std = source.geometry, source.mean, source.std
        self.device, self.is_cuda, self.dtype, self.shape = source.device, True, source.dtype, source.shape
        self._frame_shape, self._frame_stride = tuple(source.videos[0].shape), source.videos[0].stride(0)
        self.table = torch.zeros(3 * _abi.FRAG_MAX_CLIPS, dtype=torch.int64, device=self.device)
        self._c = None
        self.load(source)

    def load(self, source: FragmentSource):
        """point the slot at ``source`` (same geometry, frame shape and normalisation), in stream order"""
        if not (source.geometry == self.geometry and source.shape == self.shape and source.mean == self.mean and source.std == self.std
                and tuple(source.videos[0].shape) == self._frame_shape and source.videos[0].stride(0) == self._frame_stride
                and source.videos[0].dtype == torch.uint8):
            # a real exception (not an assert: python -O must not replay a graph recorded with other constants)
            raise ValueError("FragmentSlot.load: a slot serves one sampler geometry, frame shape / stride, uint8 frames and one "
                             f"normalisation; got geometry {source.geometry} frames {tuple(source.videos[0].shape)} {source.videos[0].dtype}")
        self.table.copy_(source.pointer_table(), non_blocking=True)
        self.current = source                      # keeps the frames alive while the table names them
        self.videos, self.hoffs, self.woffs = source.videos, source.hoffs, source.woffs

    def release(self, stream=None):
        """drop the slot's reference to the loaded source once the launch that reads it has been ENQUEUED on ``stream``: the caching
        allocator keeps the frames' memory until that stream passes this point (record_stream), the Python objects go now"""
        cur = getattr(self, "current", None)
        if cur is None:
            return
        if stream is not None:
            cur.record_stream(stream)
            tab = getattr(cur, "_table", None)
            if tab is not None:
                tab.record_stream(stream)
        self.current = None
        self.videos = self.hoffs = self.woffs = None

    def c_struct(self, any_dtype=False):
        if self._c is None:
            f = _abi.KvqFragmentSource()
            f.chan_stride = self._frame_stride
            f.n_clips, f.src_is_u8, f.Hs, f.Ws = self.shape[0], 1, self._frame_shape[2], self._frame_shape[3]
            f.Fh, f.Fw, f.fs_h, f.fs_w, f.aligned = self.geometry
            f.normalise = int(self.mean is not None)
            if self.mean is not None:
                for c in range(self._frame_shape[0]):
                    f.mean[c], f.std[c] = self.mean[c], self.std[c]
            f.indirect = ptr(self.table)
            self._c = f
        return self._c

    def materialise(self, out=None):
        # the two-step form reads the loaded source's own addresses: inside a recording that would freeze THIS video's pointers
        # into the graph, so a forward that cannot take the fused read fails its capture (LaneGraphs then runs it eagerly)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FragmentSlot.materialise() inside a hipGraph capture: this forward does not read the batch through the sampler")
        if getattr(self, "current", None) is None:
            raise RuntimeError("FragmentSlot.materialise(): no source loaded (load() one first)")
        return self.current.materialise(out)

    def split_clips(self, num_clips):
        assert num_clips == 1, "split the source, then load it"
        return self


# ------------------------------------------------------------------------------------------------
# convolution front-ends (channels-last 16-bit activations (B,D,H,W,C))
# ------------------------------------------------------------------------------------------------
def _i32x(vals):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def conv_out_dims(dims, kernel, stride, pad):
    return tuple((d + 2 * p - k) // s + 1 for d, k, s, p in zip(dims, kernel, stride, pad))


def im2col_nd(x: torch.Tensor, dims5, strides5, kernel, stride, pad, out_dtype, k_pad=None):
    """x: device tensor (fp32 network input or 16-bit activation) addressed through ELEMENT strides5 =
    (b,c,d,h,w) over dims5 = (B,C,D,H,W).  Returns ([B*Do*Ho*Wo, Kpad] 16-bit, (Do,Ho,Wo))."""
    _need_gpu(x)
    B, Cc, D, H, W = dims5
    Do, Ho, Wo = conv_out_dims((D, H, W), kernel, stride, pad)
    K = kernel[0] * kernel[1] * kernel[2] * Cc
    k_pad = -(-K // 32) * 32 if k_pad is None else k_pad
    out = torch.empty(B * Do * Ho * Wo, k_pad, dtype=out_dtype, device=x.device)
    st = (C.c_int64 * 5)(*[int(s) for s in strides5])
    check(lib().kvq_im2col_nd(ptr(x), int(x.dtype == torch.float32), dtype_code(out_dtype), C.byref(st),
                              C.byref(_i32x(dims5)), C.byref(_i32x(kernel)), C.byref(_i32x(stride)),
                              C.byref(_i32x(pad)), k_pad, ptr(out), current_stream()), "kvq_im2col_nd")
    return out, (Do, Ho, Wo)


def pack_channels_last8(x: torch.Tensor, dims5, strides5, out_dtype):
    """fp32 frames addressed through ELEMENT strides5 = (b,t,c,h,w) over dims5 = (B,T,C,H,W), C <= 8 -> 16-bit channels-last
    (B*T, H, W, 8) with the channels >= C zero (the implicit-GEMM operand of a stem conv)."""
    _need_gpu(x)
    assert x.dtype == torch.float32
    B, T, Cc, H, W = dims5
    out = torch.empty(B * T, H, W, 8, dtype=out_dtype, device=x.device)
    st = (C.c_int64 * 5)(*[int(s) for s in strides5])
    check(lib().kvq_pack_channels_last8(ptr(x), C.byref(_i32x(dims5)), C.byref(st), dtype_code(out_dtype), ptr(out),
                                        current_stream()), "kvq_pack_channels_last8")
    return out


def pool_nd(x: torch.Tensor, kernel, stride, pad, is_max: bool):
    """x (B,D,H,W,C) channels-last 16-bit, contiguous."""
    _need_gpu(x)
    assert x.dtype in HALF_TYPES and x.is_contiguous() and x.dim() == 5
    B, D, H, W, Cc = x.shape
    Do, Ho, Wo = conv_out_dims((D, H, W), kernel, stride, pad)
    out = torch.empty(B, Do, Ho, Wo, Cc, dtype=x.dtype, device=x.device)
    check(lib().kvq_pool_nd(ptr(x), dtype_code(x.dtype), C.byref(_i32x((B, Cc, D, H, W))), C.byref(_i32x(kernel)),
                            C.byref(_i32x(stride)), C.byref(_i32x(pad)), int(is_max), ptr(out), current_stream()),
          "kvq_pool_nd")
    return out


def slow_bottleneck(x: torch.Tensor, pack: torch.Tensor, ci: int, cout: int, out=None):
    """One identity residual block of SlowFast's slow pathway (res2) in one launch (csrc/slowneck.hip).  x (B,T,H,W,cin) channels-last
    16-bit; ``pack`` from ``models.backbones.slowfast_model.pack_slow_bottleneck``; ``out`` (B,T,H,W,C >= cout) receives channels 0..cout-1."""
    _need_gpu(x, pack)
    B, T, H, W, cin = x.shape
    assert x.is_contiguous() and x.dtype in HALF_TYPES
    assert pack.numel() == lib().kvq_slow_bottleneck_pack_bytes(cin, ci, cout) > 0, "block not built / wrong image size"
    if out is None:
        out = torch.empty(B, T, H, W, cout, dtype=x.dtype, device=x.device)
    assert tuple(out.shape[:4]) == (B, T, H, W) and out.is_contiguous() and out.dtype == x.dtype
    check(lib().kvq_slow_bottleneck(ptr(x), _i32x((B, T, H, W)), cin, ci, cout, ptr(pack), dtype_code(x.dtype), ptr(out), out.shape[4],
                                    stream_of(x)), "kvq_slow_bottleneck")
    return out


def fast_bottleneck(x: torch.Tensor, pack: torch.Tensor, ci: int, cout: int, projection: bool, stride: int = 1):
    """One residual block of SlowFast's fast pathway in one launch (csrc/bottleneck.hip).  x (B,T,H,W,cin) channels-last 16-bit,
    ``pack`` the uint8 image described in include/kvq_hip.h -> (B,T,ceil(H/stride),ceil(W/stride),cout)."""
    _need_gpu(x, pack)
    assert x.dtype in HALF_TYPES and x.is_contiguous() and x.dim() == 5 and pack.dtype == torch.uint8 and pack.is_contiguous()
    B, T, H, W, cin = x.shape
    assert pack.numel() == lib().kvq_fast_bottleneck_pack_bytes(cin, ci, cout, int(projection), stride) > 0, "block not built / wrong image size"
    out = torch.empty(B, T, -(-H // stride), -(-W // stride), cout, dtype=x.dtype, device=x.device)
    check(lib().kvq_fast_bottleneck(ptr(x), _i32x((B, T, H, W)), cin, ci, cout, int(projection), stride, ptr(pack), dtype_code(x.dtype),
                                    ptr(out), stream_of(x)), "kvq_fast_bottleneck")
    return out


def mean_std_pool(x: torch.Tensor, out: torch.Tensor, mean_off: int, std_off: int):
    """x 16-bit [rows, HW, C]; writes fp32 mean/unbiased-std into out[row, mean_off:+C] / out[row, std_off:+C]."""
    _need_gpu(x, out)
    assert x.dtype in HALF_TYPES and x.is_contiguous() and out.dtype == torch.float32 and out.stride(-1) == 1
    rows, HW, Cc = x.shape
    check(lib().kvq_mean_std_pool(ptr(x), dtype_code(x.dtype), rows, HW, Cc, ptr(out), out.stride(0), mean_off, std_off,
                                  current_stream()), "kvq_mean_std_pool")
    return out


def conv_gemm(A: torch.Tensor, W: torch.Tensor, bias, relu: bool, resid=None, resid_f32=None, want_f32=False):
    """out[M,N] = (relu)(A @ W^T + bias (+ resid)), 16-bit operands.  ``resid`` 16-bit / ``resid_f32`` fp32
    identity branch (ReLU epilogue only).  Returns the 16-bit output, or (16-bit, fp32 copy) when
    ``want_f32`` — the fp32 copy keeps a residual stream un-rounded between blocks."""
    _need_gpu(A, W, bias, resid, resid_f32)
    M, N = A.shape[0], W.shape[0]
    out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    out32 = torch.empty(M, N, dtype=torch.float32, device=A.device) if want_f32 else None
    a = _abi.KvqGemmArgs()
    a.A, a.W, a.bias, a.M, a.N, a.K = ptr(A), ptr(W), ptr(bias), M, N, A.shape[1]
    a.epilogue = _abi.EPI_RELU_BF16 if relu else _abi.EPI_BIAS_BF16
    assert relu or (resid is None and resid_f32 is None and not want_f32), "identity add / fp32 copy: ReLU epilogue only"
    a.out_bf16, a.out_f32, a.resid_bf16, a.resid_f32 = ptr(out), ptr(out32), ptr(resid), ptr(resid_f32)
    a.dtype = dtype_code(A.dtype)
    _ws = _splitk_scratch(a, a.M, a.N, a.K, A.device) if a.epilogue != _abi.EPI_QKV_BF16 else None   # noqa: F841
    check(lib().kvq_gemm_bf16(C.byref(a), current_stream()), "kvq_gemm_bf16")
    return (out, out32) if want_f32 else out


def vit_embed_ln(tok: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, ln_w, ln_b, B: int, eps=1e-5):
    """tok fp32 [B*G, D] (patch embeddings), cls [D], pos [1+G, D] -> LN(concat(cls, tok) + pos): fp32 (B, 1+G, D)."""
    _need_gpu(tok, cls, pos, ln_w, ln_b)
    G, D = tok.shape[0] // B, tok.shape[1]
    assert tok.dtype == torch.float32 and tok.is_contiguous() and pos.shape == (G + 1, D) and pos.is_contiguous()
    out = torch.empty(B, G + 1, D, dtype=torch.float32, device=tok.device)
    check(lib().kvq_vit_embed_ln(ptr(tok), ptr(cls), ptr(pos), ptr(ln_w), ptr(ln_b), B, G, D, eps, ptr(out), current_stream()),
          "kvq_vit_embed_ln")
    return out


def mha_small(qkv: torch.Tensor, B: int, L: int, heads: int):
    """qkv 16-bit [B*L, 3*D] (in_proj output, rows [q|k|v]) -> softmax(q k^T / sqrt(hd)) v, heads concatenated: [B*L, D]."""
    _need_gpu(qkv)
    assert qkv.dtype in HALF_TYPES and qkv.is_contiguous() and qkv.shape[0] == B * L and qkv.shape[1] % (3 * heads) == 0
    D = qkv.shape[1] // 3
    out = torch.empty(B * L, D, dtype=qkv.dtype, device=qkv.device)
    check(lib().kvq_mha_small(ptr(qkv), B, L, heads, D // heads, dtype_code(qkv.dtype), ptr(out), current_stream()), "kvq_mha_small")
    return out


def mha_cross(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, heads: int, scale: float):
    """q 16-bit [B*Lq, >= heads*64] / k, v [B*Lk, ...] (row-strided 2-D views allowed, last dim contiguous) ->
    softmax(scale q k^T) v per head: [B*Lq, heads*64]."""
    _need_gpu(q, k, v)
    for t in (q, k, v):
        assert t.dtype in HALF_TYPES and t.dim() == 2 and t.stride(1) == 1
    Lq, Lk, D = q.shape[0] // B, k.shape[0] // B, heads * 64
    out = torch.empty(B * Lq, D, dtype=q.dtype, device=q.device)
    check(lib().kvq_mha_cross(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), B, Lq, Lk, heads, 64,
                              scale, dtype_code(q.dtype), ptr(out), current_stream()), "kvq_mha_cross")
    return out


def to_half(x: torch.Tensor, out_dtype):
    """fp32 activation -> 16-bit MFMA operand (same shape)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib().kvq_convert(ptr(x), ptr(out), x.numel(), 1, dtype_code(out_dtype), current_stream()), "kvq_convert")
    return out


def to_float(x: torch.Tensor):
    """16-bit activation -> fp32 (same shape)."""
    _need_gpu(x)
    assert x.dtype in HALF_TYPES and x.is_contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().kvq_convert(ptr(x), ptr(out), x.numel(), 0, dtype_code(x.dtype), current_stream()), "kvq_convert")
    return out


def sem_modulate(x: torch.Tensor, inp: torch.Tensor, w_gama, b_gama: float, w_beta, b_beta: float):
    """x, inp fp32 [M, C] token rows -> sigmoid(<w_gama, x_m> + b_gama) * inp_m + (<w_beta, x_m> + b_beta)."""
    _need_gpu(x, inp, w_gama, w_beta)
    assert x.dtype == torch.float32 and inp.dtype == torch.float32 and x.is_contiguous() and inp.is_contiguous() and x.shape == inp.shape
    out = torch.empty_like(inp)
    check(lib().kvq_sem_modulate(ptr(x), ptr(inp), ptr(w_gama), float(b_gama), ptr(w_beta), float(b_beta), x.shape[0], x.shape[1],
                                 ptr(out), current_stream()), "kvq_sem_modulate")
    return out


def dist_modulate(inp: torch.Tensor, gamma_logit: torch.Tensor, beta: torch.Tensor):
    """inp fp32 (B, rows, C); gamma_logit / beta 16-bit [B, C] -> sigmoid(gamma_logit)[:, None] * inp + beta[:, None]."""
    _need_gpu(inp, gamma_logit, beta)
    assert inp.dtype == torch.float32 and inp.is_contiguous() and gamma_logit.dtype in HALF_TYPES and beta.dtype == gamma_logit.dtype
    B, rows, Cc = inp.shape
    out = torch.empty_like(inp)
    check(lib().kvq_dist_modulate(ptr(inp), ptr(gamma_logit), ptr(beta), B, rows, Cc, dtype_code(beta.dtype), ptr(out),
                                  current_stream()), "kvq_dist_modulate")
    return out


def qrs_top_region(score: torch.Tensor, gh: int, gw: int, kh: int, kw: int):
    """score fp32 (BK, gs, gs) -> int32 [BK]: index of the kh x kw window of the (gh, gw)-upsampled map with the largest mean."""
    _need_gpu(score)
    assert score.dtype == torch.float32 and score.is_contiguous() and score.dim() == 3 and score.shape[1] == score.shape[2]
    idx = torch.empty(score.shape[0], dtype=torch.int32, device=score.device)
    check(lib().kvq_qrs_top_region(ptr(score), score.shape[0], score.shape[1], gh, gw, kh, kw, ptr(idx), current_stream()),
          "kvq_qrs_top_region")
    return idx


def crop_regions(x: torch.Tensor, region: torch.Tensor, anchor: int, kh: int, kw: int):
    """x fp32 (B, C, T, H, W), region int32 [B*T] -> (B, C, T, kh*anchor, kw*anchor): every frame's selected window."""
    _need_gpu(x, region)
    assert x.dtype == torch.float32 and x.is_contiguous() and region.dtype == torch.int32 and region.is_contiguous()
    B, Cc, T, H, W = x.shape
    out = torch.empty(B, Cc, T, kh * anchor, kw * anchor, dtype=torch.float32, device=x.device)
    check(lib().kvq_crop_regions(ptr(x), ptr(region), B, Cc, T, H, W, anchor, kh, kw, ptr(out), current_stream()), "kvq_crop_regions")
    return out


def l2_normalize_rows(x: torch.Tensor, out_dtype):
    """F.normalize(x, dim=1) of fp32 [M, D] -> 16-bit [M, D]."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib().kvq_l2_normalize_rows(ptr(x), x.shape[0], x.shape[1], dtype_code(out_dtype), ptr(out), current_stream()),
          "kvq_l2_normalize_rows")
    return out


def axpby(x: torch.Tensor, y: torch.Tensor, a: float, b: float):
    """a * x + b * y, fp32, same shape."""
    _need_gpu(x, y)
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
    out = torch.empty_like(x)
    check(lib().kvq_axpby(ptr(x), ptr(y), float(a), float(b), ptr(out), x.numel(), current_stream()), "kvq_axpby")
    return out


def cls_gather(x: torch.Tensor, out_dtype):
    """x fp32 (B, L, D) -> x[:, 0] as 16-bit [B, D]."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    B, L, D = x.shape
    out = torch.empty(B, D, dtype=out_dtype, device=x.device)
    check(lib().kvq_cls_gather(ptr(x), B, L, D, dtype_code(out_dtype), ptr(out), current_stream()), "kvq_cls_gather")
    return out


def cls_mix(x: torch.Tensor, a: torch.Tensor, ratio: float = 0.5):
    """In place: x[:, 0] = ratio * a + (1 - ratio) * x[:, 0]; x fp32 (B, L, D), a 16-bit [B, D]."""
    _need_gpu(x, a)
    assert x.dtype == torch.float32 and x.is_contiguous() and a.dtype in HALF_TYPES and a.is_contiguous()
    B, L, D = x.shape
    check(lib().kvq_cls_mix(ptr(x), ptr(a), B, L, D, ratio, dtype_code(a.dtype), current_stream()), "kvq_cls_mix")
    return x


def cosine_cls(x: torch.Tensor):
    """cosine similarity of x[:, 0] with x[:, 1:] along D: fp32 (B, L-1)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    B, L, D = x.shape
    out = torch.empty(B, L - 1, dtype=torch.float32, device=x.device)
    check(lib().kvq_cosine_cls(ptr(x), B, L, D, ptr(out), current_stream()), "kvq_cosine_cls")
    return out


_TAPS = {}


def live_taps(size, kernel, stride, pad):
    """Per axis, the kernel offsets that touch the image for at least one output position (the others only ever read
    padding): 3x3 / pad 1 on a 1x1 map keeps the centre tap, 3x3 / stride 2 / pad 1 on 2x2 keeps offsets {1, 2}."""
    out = []
    for n, k, s, p in zip(size, kernel, stride, pad):
        no = (n + 2 * p - k) // s + 1
        out.append([a for a in range(k) if any(0 <= o * s - p + a < n for o in range(no))])
    return out


def prune_conv_weight(W: torch.Tensor, kernel, Cc: int, live):
    """Columns of a (kd,kh,kw,c)-ordered [N][Kpad] conv weight that belong to the ``live`` taps, zero-padded to a multiple of 32."""
    kd, kh, kw = kernel
    n = W.shape[0]
    w = W[:, :kd * kh * kw * Cc].reshape(n, kd, kh, kw, Cc)
    idx = [torch.as_tensor(l, device=W.device) for l in live]
    w = w.index_select(1, idx[0]).index_select(2, idx[1]).index_select(3, idx[2]).reshape(n, -1)
    k = w.shape[1]
    out = torch.zeros(n, (k + 31) // 32 * 32, dtype=W.dtype, device=W.device)
    out[:, :k] = w
    return out


def conv_taps(kernel, Cc: int, H: int, W: int, k_pad: int, device, live=None):
    """Tap table of ``kvq_conv_implicit``: int32 [k_pad/8][4], one row per 8-channel chunk of the (kd,kh,kw,c)-ordered
    K axis: {kd, kh, kw, ((kd*H + kh)*W + kw)*C + c0}; -1 in the last column marks the zero padding of K.  ``live``:
    per-axis lists of the kernel offsets kept (``live_taps``), in which case K only spans those taps."""
    key = (tuple(kernel), Cc, H, W, k_pad, str(device), None if live is None else tuple(map(tuple, live)))
    t = _TAPS.get(key)
    if t is None:
        import numpy as np
        kd, kh, kw = kernel
        la, lb, lc = live if live is not None else (range(kd), range(kh), range(kw))
        rows = np.full((k_pad // 8, 4), -1, np.int32)
        q = 0
        for a in la:
            for b in lb:
                for c in lc:
                    for c0 in range(0, Cc, 8):
                        rows[q] = (a, b, c, ((a * H + b) * W + c) * Cc + c0)
                        q += 1
        rows[q:, :3] = 0
        t = _TAPS[key] = torch.from_numpy(rows).to(device)
    return t


TAP_TABLE = False      # test hook: True = always hand kvq_conv_implicit a tap table (tap walk vs table: bit-identical)


def conv_implicit(x: torch.Tensor, W: torch.Tensor, bias, kernel, stride, pad, relu: bool, resid=None, resid_f32=None,
                  want_f32=False, store_f32=False, live=None):
    """Conv (+ folded BN, + identity, + ReLU) on a channels-last 16-bit activation x (B, D, H, W, C), C % 8 == 0, without a
    patch matrix: W [N][Kpad] with the (kd,kh,kw,c) column order of ``im2col_nd``.  Returns the (B, Do, Ho, Wo, N) 16-bit
    output, or (output, fp32 copy [M][N]) when ``want_f32``; ``store_f32``: only the fp32 [M][N] result (no ReLU).
    ``live`` (``live_taps``): W holds the columns of those taps only (``prune_conv_weight``)."""
    _need_gpu(x, W, bias, resid, resid_f32)
    assert x.dtype in HALF_TYPES and x.is_contiguous() and x.dim() == 5 and W.dtype == x.dtype and W.is_contiguous()
    B, D, H, Wd, Cc = x.shape
    if Cc % 8:
        raise _abi.KvqError(f"kvq_conv_implicit: channels-last input needs C % 8 == 0 (got C={Cc})")
    Do, Ho, Wo = conv_out_dims((D, H, Wd), kernel, stride, pad)
    N, k_pad = W.shape
    M = B * Do * Ho * Wo
    out = None if store_f32 else torch.empty(M, N, dtype=x.dtype, device=x.device)
    out32 = torch.empty(M, N, dtype=torch.float32, device=x.device) if (want_f32 or store_f32) else None
    assert relu or (resid is None and resid_f32 is None and not want_f32), "identity add / fp32 copy: ReLU epilogue only"
    a = _abi.KvqConvArgs()
    # a tap table only where it is needed: pruned taps (the table DEFINES K) or C % 32 != 0; otherwise the kernel walks the full
    # tap set with wave-uniform counters (no table loads between the slice DMAs)
    table = conv_taps(kernel, Cc, H, Wd, k_pad, x.device, live) if (live is not None or Cc % 32 or TAP_TABLE) else None
    a.x, a.W, a.bias, a.taps = ptr(x), ptr(W), ptr(bias), ptr(table)
    a.dims5[:] = (B, Cc, D, H, Wd)
    a.kernel3[:], a.stride3[:], a.pad3[:] = tuple(kernel), tuple(stride), tuple(pad)
    a.Kpad, a.N = k_pad, N
    a.epilogue = _abi.EPI_STORE_F32 if store_f32 else (_abi.EPI_RELU_BF16 if relu else _abi.EPI_BIAS_BF16)
    a.dtype = dtype_code(x.dtype)
    a.out_bf16, a.out_f32, a.resid_bf16, a.resid_f32 = ptr(out), ptr(out32), ptr(resid), ptr(resid_f32)
    _ws = _splitk_scratch(a, x.shape[0] * Do * Ho * Wo, N, k_pad, x.device)   # noqa: F841
    check(lib().kvq_conv_implicit(C.byref(a), current_stream()), "kvq_conv_implicit")
    if store_f32:
        return out32
    out = out.reshape(B, Do, Ho, Wo, N)
    return (out, out32) if want_f32 else out


def resize_bilinear(video: torch.Tensor, rh: int, rw: int, crop=None, mean=None, std=None, round_u8=None):
    """video u8|fp32 (C,T,H,W) -> bilinear resize to (rh,rw) [-> crop (cy,cx,oh,ow)] [-> (v-mean)/std], fp32."""
    _need_gpu(video)
    assert video.dtype in (torch.uint8, torch.float32) and video.is_contiguous()
    Cc, T, H, W = video.shape
    cy, cx, oh, ow = crop if crop is not None else (0, 0, rh, rw)
    out = torch.empty(Cc, T, oh, ow, dtype=torch.float32, device=video.device)
    m = (C.c_float * Cc)(*mean) if mean is not None else None
    s = (C.c_float * Cc)(*std) if std is not None else None
    rnd = int(video.dtype == torch.uint8) if round_u8 is None else int(round_u8)
    check(lib().kvq_resize_bilinear(ptr(video), int(video.dtype == torch.uint8), Cc, T, H, W, rh, rw, cy, cx, oh, ow,
                                    rnd, m, s, ptr(out), stream_of(video)), "kvq_resize_bilinear")
    return out


def upsample_frames(video: torch.Tensor, scale_factor: float):
    """get_spatial_fragments' small-source fallback (fusion_datasets.py:43-50): F.interpolate(video / 255.0, scale_factor,
    mode="bilinear") * 255.0 cast back to the frame type, ATen-CPU-exact.  video u8|fp32 (C,T,H,W) -> same type, upsampled."""
    _need_gpu(video)
    assert video.dtype in (torch.uint8, torch.float32) and video.is_contiguous()
    Cc, T, H, W = video.shape
    od = (C.c_int32 * 2)()
    check(lib().kvq_upsample_frames_out_dims(H, W, float(scale_factor), od), "kvq_upsample_frames_out_dims")
    out = torch.empty(Cc, T, od[0], od[1], dtype=video.dtype, device=video.device)
    check(lib().kvq_upsample_frames(ptr(video), int(video.dtype == torch.uint8), Cc, T, H, W, float(scale_factor), ptr(out),
                                    stream_of(video)), "kvq_upsample_frames")
    return out


def block_tail_pack(proj_w, proj_b, norm2_w, norm2_b, fc1_w, fc1_b, fc2_w, fc2_b):
    """Weight image of the fused proj+norm2+Mlp launch (include/kvq_hip.h: kvq_block_tail_pack)."""
    _need_gpu(proj_w, proj_b, norm2_w, norm2_b, fc1_w, fc1_b, fc2_w, fc2_b)
    assert proj_w.dtype in HALF_TYPES and fc1_w.dtype == proj_w.dtype and fc2_w.dtype == proj_w.dtype
    Cc, hidden = proj_w.shape[0], fc1_w.shape[0]
    nbytes = lib().kvq_block_tail_pack_bytes(Cc, hidden)
    if not nbytes:
        raise _abi.KvqError(f"kvq_block_tail: unsupported width C={Cc}, hidden={hidden}")
    pack = torch.empty(nbytes, dtype=torch.uint8, device=proj_w.device)
    check(lib().kvq_block_tail_pack(ptr(proj_w.contiguous()), ptr(proj_b), ptr(norm2_w), ptr(norm2_b),
                                    ptr(fc1_w.contiguous()), ptr(fc1_b), ptr(fc2_w.contiguous()), ptr(fc2_b), Cc, hidden,
                                    ptr(pack), current_stream()), "kvq_block_tail_pack")
    return pack


def block_tail_qkv_pack(qkv_w: torch.Tensor, hidden: int):
    """Image of a block's qkv weight (16-bit [3C][C]) that the PREVIOUS block's fused tail streams to emit q | k | v
    (include/kvq_hip.h: kvq_block_tail_qkv_pack); None when this width's tail cannot."""
    _need_gpu(qkv_w)
    assert qkv_w.dtype in HALF_TYPES and qkv_w.is_contiguous()
    Cc = qkv_w.shape[1]
    nbytes = lib().kvq_block_tail_qkv_pack_bytes(Cc, hidden)
    if not nbytes:
        return None
    pack = torch.empty(nbytes, dtype=torch.uint8, device=qkv_w.device)
    check(lib().kvq_block_tail_qkv_pack(ptr(qkv_w), Cc, hidden, ptr(pack), current_stream()), "kvq_block_tail_qkv_pack")
    return pack


def block_tail(attn: torch.Tensor, x: torch.Tensor, pack: torch.Tensor, hidden: int, *, scatter_map=None, map_rows=0,
               out_rows=0, next_norm=None, next_dst=None, next_rows=0, eps=1e-5, attn_gather=None, next_qkv=None):
    """x (fp32 [n_batch*out_rows, C], in place — or fp16 at C <= 192: the round-6 residual stream of stages 0-1, ``x_f16``)
    += proj(attn) scattered; x += Mlp(norm2(x)).
    ``next_norm=(gamma, beta)`` + ``next_dst`` additionally returns norm1_next(x) in the next window order — or, with
    ``next_qkv=(qkv_pack, qkv_bias, q_scale)``, the next block's q | k | v, head-major [3][C/32][n_batch*next_rows][32]."""
    _need_gpu(attn, x, pack, scatter_map, next_dst)
    assert attn.dtype in HALF_TYPES and attn.is_contiguous() and x.dtype in (torch.float32, torch.float16) and x.is_contiguous()
    M, Cc = attn.shape
    a = _abi.KvqBlockTailArgs()
    a.x_f16 = int(x.dtype == torch.float16)
    a.attn, a.x, a.scatter_map, a.attn_gather = ptr(attn), ptr(x), ptr(scatter_map), ptr(attn_gather)
    a.map_rows, a.out_rows = (map_rows, out_rows) if (scatter_map is not None or attn_gather is not None) else (M, M)
    a.M, a.C, a.hidden, a.pack, a.eps, a.dtype = M, Cc, hidden, ptr(pack), eps, dtype_code(attn.dtype)
    nxt = None
    if next_norm is not None:
        n_batch = x.shape[0] // a.out_rows
        a.next_norm_w, a.next_norm_b, a.next_dst, a.next_rows = ptr(next_norm[0]), ptr(next_norm[1]), ptr(next_dst), next_rows
        if next_qkv is not None:
            nxt = torch.empty(3, Cc // 32, n_batch * next_rows, 32, dtype=attn.dtype, device=x.device)
            a.next_qkv_pack, a.next_qkv_b, a.qkv_out, a.q_scale, a.num_heads = ptr(next_qkv[0]), ptr(next_qkv[1]), ptr(nxt), float(next_qkv[2]), Cc // 32
        else:
            nxt = torch.empty(n_batch * next_rows, Cc, dtype=attn.dtype, device=x.device)
            a.next_ln = ptr(nxt)
    check(lib().kvq_block_tail(C.byref(a), current_stream()), "kvq_block_tail")
    return nxt


def patch_embed(x, w: torch.Tensor, bias, ln_w, ln_b, patch, *, next_norm=None, next_dst=None, next_rows=0,
                eps=1e-5, out_f16=False):
    """PatchEmbed3D as one launch: x fp32 (B,Cin,T,H,W) or a ``FragmentSource``, w 16-bit [E][Cin*pd*ph*pw] -> fp32
    [B*D0*H0*W0, E] (+ the first block's norm1 rows when ``next_norm=(gamma, beta)`` / ``next_dst`` are given)."""
    frag = None
    if isinstance(x, FragmentSource):
        frag = x.c_struct()
        if frag is None:
            raise _abi.KvqError("kvq_patch_embed: this FragmentSource has no fused read (materialise() it)")
        _need_gpu(w, bias, ln_w, ln_b, next_dst)
    else:
        _need_gpu(x, w, bias, ln_w, ln_b, next_dst)
        assert x.dtype == torch.float32 and x.is_contiguous()
    assert w.dtype in HALF_TYPES and w.is_contiguous()
    B, Cin, T, H, W = x.shape
    pd, ph, pw = patch
    Ed, K = w.shape
    if not lib().kvq_patch_embed_supported(Cin, pd, ph, pw, Ed, T, H, W):
        raise _abi.KvqError("kvq_patch_embed: unsupported shape (use patch_im2col + gemm + layernorm_rows)")
    pack = torch.empty(lib().kvq_patch_embed_pack_bytes(Ed, K), dtype=torch.uint8, device=x.device)
    check(lib().kvq_patch_embed_pack(ptr(w), ptr(bias), ptr(ln_w), ptr(ln_b), Ed, K, ptr(pack), current_stream()),
          "kvq_patch_embed_pack")
    L0 = (T // pd) * (H // ph) * (W // pw)
    out = torch.empty(B * L0, Ed, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)      # out_f16: the fp16 residual stream
    a = _abi.KvqPatchEmbedArgs()
    a.out_f16 = int(out_f16)
    a.x, a.B, a.in_chans, a.T, a.H, a.W, a.pd, a.ph, a.pw, a.embed_dim = (None if frag else ptr(x)), B, Cin, T, H, W, pd, ph, pw, Ed
    if frag is not None:
        a.frag = C.pointer(frag)
    a.pack, a.has_norm, a.out, a.eps, a.dtype = ptr(pack), int(ln_w is not None), ptr(out), eps, dtype_code(w.dtype)
    nxt = None
    if next_norm is not None:
        nxt = torch.empty(B * next_rows, Ed, dtype=w.dtype, device=x.device)
        a.next_norm_w, a.next_norm_b, a.next_dst, a.next_ln, a.next_rows = (ptr(next_norm[0]), ptr(next_norm[1]),
                                                                            ptr(next_dst), ptr(nxt), next_rows)
    check(lib().kvq_patch_embed(C.byref(a), current_stream()), "kvq_patch_embed")
    return out, nxt


def patch_merge(x: torch.Tensor, merge_map: torch.Tensor, n_batch: int, red_w: torch.Tensor, norm_w, norm_b, out_dtype=torch.float16, *,
                next_norm=None, next_dst=None, next_rows=0, eps=1e-5, out_f16=False):
    """PatchMerging as one launch (C = 96 / 128 / 192): x fp32 (or fp16: the round-6 residual stream; ``out_f16``: the merged one too) [n_batch*L, C], merge_map int32 [Ln, 4] (-1 = zero padding), red_w fp32 [2C, 4C]
    -> fp32 [n_batch*Ln, 2C] (+ the next block's norm1 rows, ``out_dtype``, when ``next_norm=(gamma, beta)`` / ``next_dst`` are given)."""
    _need_gpu(x, merge_map, red_w, norm_w, norm_b, next_dst)
    assert x.dtype in (torch.float32, torch.float16) and x.is_contiguous() and merge_map.dtype == torch.int32 and merge_map.is_contiguous()
    assert red_w.dtype == torch.float32 and red_w.is_contiguous()
    Cc = x.shape[1]
    L, Ln = x.shape[0] // n_batch, merge_map.shape[0]
    nb = lib().kvq_patch_merge_pack_bytes(Cc)
    if not nb:
        raise _abi.KvqError("kvq_patch_merge: unsupported width (use layernorm_rows + gemm)")
    pack = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(lib().kvq_patch_merge_pack(ptr(red_w), ptr(norm_w), ptr(norm_b), Cc, dtype_code(out_dtype), ptr(pack), current_stream()),
          "kvq_patch_merge_pack")
    out = torch.empty(n_batch * Ln, 2 * Cc, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)
    a = _abi.KvqPatchMergeArgs()
    a.x_f16, a.out_f16 = int(x.dtype == torch.float16), int(out_f16)
    a.x, a.merge_map, a.B, a.L, a.Ln, a.C, a.pack, a.out = ptr(x), ptr(merge_map), n_batch, L, Ln, Cc, ptr(pack), ptr(out)
    a.eps, a.dtype = eps, dtype_code(out_dtype)
    nxt = None
    if next_norm is not None:
        nxt = torch.empty(n_batch * next_rows, 2 * Cc, dtype=out_dtype, device=x.device)
        a.next_norm_w, a.next_norm_b, a.next_dst, a.next_ln, a.next_rows = (ptr(next_norm[0]), ptr(next_norm[1]), ptr(next_dst),
                                                                            ptr(nxt), next_rows)
    check(lib().kvq_patch_merge(C.byref(a), current_stream()), "kvq_patch_merge")
    return out, nxt


def stem_mfma_pack_weight(w_kc: torch.Tensor, kernel, cin: int, out_dtype):
    """[K][Cout] fp32 stem weight (K ordered kd,kh,kw,c; Cout <= 8, kw = 7, cin <= 4) -> the 16-bit [kd*kh][16][32] image of
    ``kvq_conv_stem_mfma``: k = tap * 4 + c, zero rows / taps / channels as padding."""
    kd, kh, kw = kernel
    cout = w_kc.shape[1]
    assert kw == 7 and cin <= 4 and cout <= 8 and w_kc.shape[0] == kd * kh * kw * cin
    w = w_kc.reshape(kd * kh, kw, cin, cout).permute(0, 3, 1, 2)                 # [slice][o][tap][c]
    img = torch.zeros(kd * kh, 16, 8, 4, dtype=torch.float32, device=w_kc.device)
    img[:, :cout, :kw, :cin] = w
    if out_dtype == torch.float16:
        img = img.clamp(-65504.0, 65504.0)
    return img.reshape(kd * kh, 16, 32).to(out_dtype).contiguous()


def conv_stem_mfma(x: torch.Tensor, wpack: torch.Tensor, bias8: torch.Tensor, kernel, stride, pad, relu: bool):
    """Conv3d(C <= 4 -> 8, (kd, kh, 7), stride (sd, sh, 2), padding (pd, ph, 3)) + bias [+ ReLU] on the matrix cores: x fp32
    (B,C,T,H,W) contiguous -> 16-bit channels-last (B,Do,Ho,Wo,8).  ``wpack`` from ``stem_mfma_pack_weight``."""
    _need_gpu(x, wpack, bias8)
    assert x.dtype == torch.float32 and x.is_contiguous() and wpack.dtype in HALF_TYPES and bias8.numel() == 8
    B, Cin, T, H, W = x.shape
    x4 = torch.empty(B, T, H, W + 8, 4, dtype=wpack.dtype, device=x.device)
    check(lib().kvq_pack_clip_cl4(ptr(x), C.byref((C.c_int32 * 5)(B, Cin, T, H, W)), 4, dtype_code(wpack.dtype), ptr(x4),
                                  current_stream()), "kvq_pack_clip_cl4")
    do, ho, wo = conv_out_dims((T, H, W), kernel, stride, pad)
    out = torch.empty(B, do, ho, wo, 8, dtype=wpack.dtype, device=x.device)
    check(lib().kvq_conv_stem_mfma(ptr(x4), C.byref((C.c_int32 * 4)(B, T, H, W)), ptr(wpack), ptr(bias8), C.byref(_i32x(kernel)),
                                   C.byref(_i32x(stride)), C.byref(_i32x(pad)), int(relu), dtype_code(wpack.dtype), ptr(out),
                                   current_stream()), "kvq_conv_stem_mfma")
    return out


def conv_stem_pool(x: torch.Tensor, wpack: torch.Tensor, bias8: torch.Tensor, kd: int, relu: bool = True):
    """SlowFast's fast-pathway stem in one launch (``kvq_conv_stem_pool``): Conv3d(3 -> 8, (kd,7,7), stride (1,2,2), padding
    (kd//2,3,3)) + bias [+ ReLU] + MaxPool3d((1,3,3), (1,2,2), (0,1,1)); x fp32 (B,3,T,H,W) -> 16-bit channels-last (B,T,Hp,Wp,8)."""
    _need_gpu(x, wpack, bias8)
    assert x.dtype == torch.float32 and x.is_contiguous() and wpack.dtype in HALF_TYPES and bias8.numel() == 8
    B, Cin, T, H, W = x.shape
    hp, wp_ = ((H - 1) // 2 + 1 - 1) // 2 + 1, ((W - 1) // 2 + 1 - 1) // 2 + 1
    out = torch.empty(B, T, hp, wp_, 8, dtype=wpack.dtype, device=x.device)
    check(lib().kvq_conv_stem_pool(ptr(x), C.byref((C.c_int32 * 5)(B, Cin, T, H, W)), ptr(wpack), ptr(bias8), int(kd), int(relu),
                                   dtype_code(wpack.dtype), ptr(out), current_stream()), "kvq_conv_stem_pool")
    return out


def stem64_pack_weight(w_ok: torch.Tensor, out_dtype):
    """[64][>= 147] stem weight (K ordered kh,kw,c over a 1x7x7 kernel, 3 input channels) -> the 16-bit [7][64][32] image of
    ``kvq_conv_stem64_pool``: entry [kh][o][kw * 4 + c], tap 7 and channel 3 zero."""
    assert w_ok.shape[0] == 64 and w_ok.shape[1] >= 147
    img = torch.zeros(7, 64, 8, 4, dtype=torch.float32, device=w_ok.device)
    img[:, :, :7, :3] = w_ok[:, :147].float().reshape(64, 7, 7, 3).permute(1, 0, 2, 3)
    if out_dtype == torch.float16:
        img = img.clamp(-65504.0, 65504.0)
    return img.reshape(7, 64, 32).to(out_dtype).contiguous()


def conv_stem64_pool(x: torch.Tensor, t_index, wimg: torch.Tensor, bias64: torch.Tensor, relu: bool = True, out=None, out_coff: int = 0):
    """SlowFast's slow-pathway stem in one launch (``kvq_conv_stem64_pool``): frames ``t_index`` of the fp32 clip x (B,3,T,H,W) ->
    Conv3d(3 -> 64, (1,7,7), (1,2,2), (0,3,3)) + bias [+ ReLU] + MaxPool3d((1,3,3), (1,2,2), (0,1,1)); 16-bit channels-last
    (B,F,Hp,Wp,64), or channels out_coff .. out_coff+63 of ``out`` (B,F,Hp,Wp,C)."""
    _need_gpu(x, wimg, bias64)
    assert x.dtype == torch.float32 and x.is_contiguous() and wimg.dtype in HALF_TYPES and bias64.numel() == 64
    B, Cin, T, H, W = x.shape
    ti = None if t_index is None else torch.as_tensor(t_index, dtype=torch.int32).to(x.device)
    F_ = T if ti is None else ti.numel()
    hp, wp_ = ((H - 1) // 2 + 1 - 1) // 2 + 1, ((W - 1) // 2 + 1 - 1) // 2 + 1
    if out is None:
        out = torch.empty(B, F_, hp, wp_, 64, dtype=wimg.dtype, device=x.device)
    assert tuple(out.shape[:4]) == (B, F_, hp, wp_) and out.is_contiguous() and out.dtype == wimg.dtype
    check(lib().kvq_conv_stem64_pool(ptr(x), C.byref((C.c_int32 * 5)(B, Cin, T, H, W)), ptr(ti) if ti is not None else None, F_, ptr(wimg),
                                     ptr(bias64), int(relu), dtype_code(wimg.dtype), ptr(out), out.shape[4], out_coff, current_stream()),
          "kvq_conv_stem64_pool")
    return out


def conv_stem_direct(x: torch.Tensor, w_kc: torch.Tensor, bias: torch.Tensor, kernel, stride, pad, relu: bool, out_dtype):
    """Direct Conv3d for few output channels: x fp32 (B,C,D,H,W), w_kc fp32 [K][Cout] (K ordered kd,kh,kw,c), -> 16-bit
    channels-last (B,Do,Ho,Wo,Cout)."""
    _need_gpu(x, w_kc, bias)
    assert x.dtype == torch.float32 and x.is_contiguous() and w_kc.dtype == torch.float32 and w_kc.is_contiguous()
    B, Cin, D, H, W = x.shape
    cout = w_kc.shape[1]
    do, ho, wo = conv_out_dims((D, H, W), kernel, stride, pad)
    out = torch.empty(B, do, ho, wo, cout, dtype=out_dtype, device=x.device)
    check(lib().kvq_conv_stem_direct(ptr(x), C.byref((C.c_int32 * 5)(B, Cin, D, H, W)), ptr(w_kc), ptr(bias), cout,
                                     C.byref(_i32x(kernel)), C.byref(_i32x(stride)), C.byref(_i32x(pad)), int(relu),
                                     dtype_code(out_dtype), ptr(out), current_stream()), "kvq_conv_stem_direct")
    return out
