"""Host-side mirror of the reference's ``models/backbones/simpleVQA_model.py`` ``ResNet`` (:128-264):
2D ResNet-50 on the video frames with (avg, std) pooling after layer2/3/4, concatenated with the
pre-extracted SlowFast features -> (B, T, 9472).  Same module tree / state_dict keys as the reference
(torchvision layout incl. BatchNorm buffers and the unused ``quality`` regressor);
forward = im2col + MFMA GEMM (BatchNorm folded, ReLU / identity add in the epilogue) + pooling kernels
of libkvq_hip.so on channels-last 16-bit activations.  No PyTorch compute path."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import _abi, kernels
from .swin_backbone import _Affine


IMPLICIT_CONV = True     # 0: materialised im2col + GEMM
CONVNET = True                  # 0: the layer-by-layer Python sequencing


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, stride=1, pad=0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")     # simpleVQA_model.py:171
        self.kernel, self.stride, self.pad = k, stride, pad


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1, self.bn1 = _Conv(inplanes, planes, 1), _BN(planes)
        self.conv2, self.bn2 = _Conv(planes, planes, 3, stride, 1), _BN(planes)
        self.conv3, self.bn3 = _Conv(planes, planes * 4, 1), _BN(planes * 4)
        self.downsample = (nn.Sequential(_Conv(inplanes, planes * 4, 1, stride), _BN(planes * 4))
                           if downsample else None)
        self.stride = stride


def _fold(conv: _Conv, bn: _BN, half, device):
    """BatchNorm (eval) folded into the conv: w' = w*g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps);
    weight reordered to the im2col column order (kh,kw,c), zero padded to a multiple of 32, 16-bit."""
    w = conv.weight.detach().to(device=device, dtype=torch.float32)
    scale = bn.weight.detach().to(device, torch.float32) / torch.sqrt(bn.running_var.to(device, torch.float32) + bn.eps)
    bias = bn.bias.detach().to(device, torch.float32) - bn.running_mean.to(device, torch.float32) * scale
    w = (w * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    K = w.shape[1]
    kpad = -(-K // 32) * 32
    if kpad != K:
        w = torch.nn.functional.pad(w, (0, kpad - K))
    if half == torch.float16:
        w = w.clamp(-65504.0, 65504.0)
    return w.to(half).contiguous(), bias.contiguous()


class ResNet(nn.Module):
    def __init__(self, block=Bottleneck, layers=(3, 4, 6, 3), operand_dtype=None, **kwargs):
        super().__init__()
        self.operand_dtype = _abi.dtype_code(operand_dtype or os.environ.get("KVQ_OPERAND_DTYPE", "fp16"))
        self.inplanes = 64
        self.conv1, self.bn1 = _Conv(3, 64, 7, 2, 3), _BN(64)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        # unused by forward but part of the reference's state_dict (simpleVQA_model.py:167)
        self.quality = nn.Sequential(_Affine((128, 4096 + 2048 + 1024 + 2048 + 256), (128,)), _Affine((1, 128), (1,)))
        self._wcache = None
        # features() only: carry the stream between bottlenecks in 16 bits (CONTRIQUE_model switches it on; forward() — the SimpleVQA
        # path pinned to the reference by |dscore| <= 1e-3 — always keeps the fp32 stream)
        self.residual16 = False

    def _make_layer(self, planes, blocks, stride):
        layers = [Bottleneck(self.inplanes, planes, stride, downsample=(stride != 1 or self.inplanes != planes * 4))]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    # ---- weights ---------------------------------------------------------------------------------
    def _weights(self, device):
        sig = (self.operand_dtype,) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) +
                                            list(self.buffers()))
        if self._wcache is not None and self._wcache[0] == sig:
            return self._wcache[1]
        half = _abi.torch_dtype(self.operand_dtype)
        w = {"stem": _fold(self.conv1, self.bn1, half, device)}
        # the same stem weight for the channel-padded implicit conv: (kh,kw,c<3) columns spread to (kh,kw,8), K 392 -> 416
        ws = w["stem"][0][:, :147].reshape(64, 49, 3)
        w8 = torch.zeros(64, 416, dtype=ws.dtype, device=device)
        w8[:, :392].view(64, 49, 8)[:, :, :3] = ws
        w["stem8"] = w8
        self.__dict__["_pruned"] = {}
        for li, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4), 1):
            for bi, blk in enumerate(layer):
                k = f"l{li}.{bi}."
                w[k + "1"] = _fold(blk.conv1, blk.bn1, half, device)
                w[k + "2"] = _fold(blk.conv2, blk.bn2, half, device)
                w[k + "3"] = _fold(blk.conv3, blk.bn3, half, device)
                if blk.downsample is not None:
                    w[k + "d"] = _fold(blk.downsample[0], blk.downsample[1], half, device)
        self._wcache = (sig, w)
        return w

    # ---- conv on channels-last 16-bit (N,H,W,C) ---------------------------------------------------
    @staticmethod
    def _cols(x, k, stride, pad, kpad):
        n, h, w_, c = x.shape
        if k == 1 and stride == 1:
            return x.reshape(n * h * w_, c), (h, w_)
        a, (_, ho, wo) = kernels.im2col_nd(x, (n, c, 1, h, w_), (h * w_ * c, 1, 0, w_ * c, c), (1, k, k),
                                           (1, stride, stride), (0, pad, pad), x.dtype, kpad)
        return a, (ho, wo)

    def _conv_relu(self, x, wb, k, stride, pad):
        n, h, w_, c = x.shape
        if IMPLICIT_CONV and not (k == 1 and stride == 1) and c % 8 == 0 and x.is_contiguous():
            # implicit GEMM: the A tiles are fetched from the activation itself (no patch matrix).  On tiny maps (CONTRIQUE's
            # 32x32 patches reach 2x2 and 1x1) most taps of a padded 3x3 only ever read zeros: drop them from K.
            live, wk = None, wb[0]
            if k > 1 and min(h, w_) < k:
                live = kernels.live_taps((1, h, w_), (1, k, k), (1, stride, stride), (0, pad, pad))
                if len(live[1]) * len(live[2]) < k * k:
                    cache = self.__dict__.setdefault("_pruned", {})
                    ck = (wb[0].data_ptr(), wb[0]._version, h, w_, stride)
                    if ck not in cache:
                        cache[ck] = kernels.prune_conv_weight(wb[0], (1, k, k), c, live)
                    wk = cache[ck]
                else:
                    live = None
            y = kernels.conv_implicit(x.reshape(n, 1, h, w_, c), wk, wb[1], (1, k, k), (1, stride, stride), (0, pad, pad), True,
                                      live=live)
            return y.reshape(n, y.shape[2], y.shape[3], wb[0].shape[0])
        a, (ho, wo) = self._cols(x, k, stride, pad, wb[0].shape[1])
        return kernels.conv_gemm(a, wb[0], wb[1], True).reshape(x.shape[0], ho, wo, wb[0].shape[0])

    def _bottleneck(self, x16, x32, w, key, blk):
        """x16: 16-bit activation feeding the convs; x32: the same tensor un-rounded (fp32) for the identity
        path — the residual stream stays fp32 between blocks, like the Swin trunk's."""
        n = x16.shape[0]
        out = self._conv_relu(x16, w[key + "1"], 1, 1, 0)
        out = self._conv_relu(out, w[key + "2"], 3, blk.stride, 1)
        r16 = self.residual16
        if blk.downsample is None:
            identity = (x16 if r16 else x32).reshape(-1, x16.shape[-1])
        else:                                                   # 1x1/stride conv + BN, no ReLU, kept in fp32 (16-bit with residual16)
            wd, bd = w[key + "d"]
            nb, hb, wb_, cb = x16.shape
            if IMPLICIT_CONV and blk.stride != 1 and cb % 8 == 0 and x16.is_contiguous():
                identity = kernels.conv_implicit(x16.reshape(nb, 1, hb, wb_, cb), wd, bd, (1, 1, 1), (1, blk.stride, blk.stride),
                                                 (0, 0, 0), False, store_f32=not r16)
                identity = identity.reshape(-1, identity.shape[-1])
            else:
                a, _ = self._cols(x16, 1, blk.stride, 0, wd.shape[1])
                identity = kernels.gemm(a, wd, bd, _abi.EPI_BIAS_BF16 if r16 else _abi.EPI_STORE_F32)
        ho, wo = out.shape[1], out.shape[2]
        w3, b3 = w[key + "3"]
        if r16:
            # 16-bit residual stream (as slowfast_model's default): per output element the launch reads 2 B and writes 2 B instead of
            # reading 4 and writing 4 + 2 — these 1x1 convs are HBM-bound; one more 16-bit rounding per block
            y16 = kernels.conv_gemm(out.reshape(n * ho * wo, -1), w3, b3, True, resid=identity)
            return y16.reshape(n, ho, wo, -1), None
        y16, y32 = kernels.conv_gemm(out.reshape(n * ho * wo, -1), w3, b3, True, resid_f32=identity, want_f32=True)
        return y16.reshape(n, ho, wo, -1), y32.reshape(n, ho, wo, -1)        # relu(bn3(conv3) + identity)

    # ---- the whole network as ONE C call (csrc/convnet.hip) ---------------------------------------------------------------
    def _net(self, b, T, h1, w1, feat_dim, device):
        """kvq_convnet plan of forward(): stem over the 8-channel packed frames, max-pool, the 16 bottlenecks with their
        un-rounded fp32 residual stream (identity branch / projection shortcut / fp32 copy of every block output), and the
        avgpool + global_std_pool2d of layers 2-4 written straight into the (frames, 7168 + feat) feature rows."""
        import ctypes as C
        w = self._weights(device)
        key = (b, T, h1, w1, feat_dim, str(device), self.operand_dtype, _abi.current_stream(), id(w))
        hit = self.__dict__.setdefault("_nets", {}).get(key)
        if hit is not None:
            return hit
        tens, ops = [], []

        def tensor(d, h, ww, c, kind=_abi.NET_T_ACT16):
            tens.append((b, d, h, ww, c, kind))
            return len(tens) - 1

        def op(kind, src, dst, k=(1, 1, 1), st=(1, 1, 1), pd=(0, 0, 0), **kw):
            o = _abi.KvqNetOp()
            o.kind, o.src, o.dst, o.src2, o.dst32 = kind, src, dst, kw.get("src2", -1), kw.get("dst32", -1)
            o.kernel3[:], o.stride3[:], o.pad3[:] = tuple(k), tuple(st), tuple(pd)
            for f in ("cout", "kpad", "relu", "is_max", "per_frame", "mean_off", "std_off", "out_stride"):
                if f in kw:
                    setattr(o, f, kw[f])
            for f in ("w", "bias"):
                if kw.get(f) is not None:
                    setattr(o, f, kw[f].data_ptr())
            ops.append(o)

        od = lambda n, k, st, pd: (n + 2 * pd - k) // st + 1       # noqa: E731
        x_in = tensor(T, h1, w1, 3, _abi.NET_T_F32_PLANAR)          # slot 0: (b, 3, T, h, w) frames
        hs, ws_ = od(h1, 7, 2, 3), od(w1, 7, 2, 3)
        stem = tensor(T, hs, ws_, 64)
        op(_abi.NET_STEM8, x_in, stem, (1, 7, 7), (1, 2, 2), (0, 3, 3), cout=64, kpad=w["stem8"].shape[1], relu=1, w=w["stem8"], bias=w["stem"][1])
        hp, wp = od(hs, 3, 2, 1), od(ws_, 3, 2, 1)
        y = tensor(T, hp, wp, 64)
        op(_abi.NET_POOL, stem, y, (1, 3, 3), (1, 2, 2), (0, 1, 1), is_max=1)
        y32, off, width = -1, 0, 7168 + feat_dim
        for li, layer_mod in enumerate((self.layer1, self.layer2, self.layer3, self.layer4), 1):
            for bi, blk in enumerate(layer_mod):
                k = f"l{li}.{bi}."
                _, d, hh, ww, cin, _ = tens[y]
                planes = w[k + "1"][0].shape[0]
                t1 = tensor(d, hh, ww, planes)
                op(_abi.NET_CONV, y, t1, cout=planes, kpad=w[k + "1"][0].shape[1], relu=1, w=w[k + "1"][0], bias=w[k + "1"][1])
                ho, wo = od(hh, 3, blk.stride, 1), od(ww, 3, blk.stride, 1)
                t2 = tensor(d, ho, wo, planes)
                op(_abi.NET_CONV, t1, t2, (1, 3, 3), (1, blk.stride, blk.stride), (0, 1, 1), cout=planes, kpad=w[k + "2"][0].shape[1],
                   relu=1, w=w[k + "2"][0], bias=w[k + "2"][1])
                cout = w[k + "3"][0].shape[0]
                if blk.downsample is None:
                    ident = y32
                else:                                                # 1x1 / stride conv + BN, no ReLU, kept in fp32
                    ident = tensor(d, ho, wo, cout, _abi.NET_T_ACT32)
                    op(_abi.NET_CONV, y, ident, (1, 1, 1), (1, blk.stride, blk.stride), cout=cout, kpad=w[k + "d"][0].shape[1], relu=0,
                       w=w[k + "d"][0], bias=w[k + "d"][1])
                y, y32 = tensor(d, ho, wo, cout), tensor(d, ho, wo, cout, _abi.NET_T_ACT32)
                op(_abi.NET_CONV, t2, y, cout=cout, kpad=w[k + "3"][0].shape[1], relu=1, src2=ident, dst32=y32, w=w[k + "3"][0], bias=w[k + "3"][1])
            if li >= 2:                                              # avgpool + global_std_pool2d (:242-252), per frame
                cc = tens[y][4]
                op(_abi.NET_MEAN_STD, y, 0, per_frame=1, mean_off=off, std_off=off + cc, out_stride=width)
                off += 2 * cc
        ta = (_abi.KvqNetTensor * len(tens))()
        for i, (bb, d, h, ww, c, kind) in enumerate(tens):
            ta[i].B, ta[i].D, ta[i].H, ta[i].W, ta[i].C, ta[i].kind = bb, d, h, ww, c, kind
        oa = (_abi.KvqNetOp * len(ops))(*ops)
        handle = C.c_void_p()
        _abi.check(_abi.lib().kvq_convnet_create(oa, len(ops), ta, len(tens), 1, 1, self.operand_dtype, C.byref(handle)), "kvq_convnet_create")
        ws = torch.empty(_abi.lib().kvq_convnet_workspace_bytes(handle), dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)
        entry = (handle, ws, off, w)
        self._nets[key] = entry
        return entry

    def __del__(self):
        try:
            for handle, *_ in self.__dict__.get("_nets", {}).values():
                _abi.lib().kvq_convnet_destroy(handle)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def forward(self, batch, multi=None, layer=None):
        x = batch["simpleVQA"]
        if not x.is_cuda:
            raise _abi.KvqError("ResNet.forward needs the frames on a HIP device; there is no CPU path")
        x = x.to(torch.float32).contiguous()
        b, c, T, h1, w1 = x.shape
        feat3d = batch["feat"].to(x.device, torch.float32).reshape(b * T, -1)
        if CONVNET and IMPLICIT_CONV and c == 3:
            # one C call enqueues the whole network; the SlowFast features are appended to the pooled rows
            import ctypes as C
            handle, ws, off, _ = self._net(b, T, h1, w1, feat3d.shape[1], x.device)
            out = torch.empty(b * T, off + feat3d.shape[1], dtype=torch.float32, device=x.device)
            ins, outs = (C.c_void_p * 1)(x.data_ptr()), (C.c_void_p * 1)(out.data_ptr())
            _abi.check(_abi.lib().kvq_convnet_forward(handle, ins, outs, ws.data_ptr(), ws.numel(), _abi.stream_of(x)), "kvq_convnet_forward")
            out[:, off:] = feat3d                                            # x_3D_features (:256)
            return out.reshape(b, T, -1)
        w = self._weights(x.device)
        half = _abi.torch_dtype(self.operand_dtype)
        n = b * T
        # stem 7x7/2 reads the fp32 (b,c,T,h,w) input directly: frame index = (b, t) through the strides
        y = self._stem(x, (b, T, c, h1, w1), (c * T * h1 * w1, h1 * w1, T * h1 * w1, w1, 1), w, half)
        y = kernels.pool_nd(y.unsqueeze(1), (1, 3, 3), (1, 2, 2), (0, 1, 1), True).squeeze(1)        # maxpool 3x3/2
        out = torch.empty(n, 7168 + feat3d.shape[1], dtype=torch.float32, device=x.device)
        off = 0
        y32 = None
        for li, layer_mod in enumerate((self.layer1, self.layer2, self.layer3, self.layer4), 1):
            for bi, blk in enumerate(layer_mod):
                y, y32 = self._bottleneck(y, y32, w, f"l{li}.{bi}.", blk)
            if li >= 2:                                                    # avgpool + global_std_pool2d (:242-252)
                nn_, hh, ww, cc = y.shape
                kernels.mean_std_pool(y.reshape(nn_, hh * ww, cc), out, off, off + cc)
                off += 2 * cc
        out[:, off:] = feat3d                                                # x_3D_features (:256)
        return out.reshape(b, T, -1)

    def features(self, x):
        """The convolutional trunk alone: x (N, 3, H, W) fp32 on a HIP device -> the layer4 output, channels-last, as
        (16-bit (N, H/32, W/32, 2048), its un-rounded fp32 copy)."""
        if not x.is_cuda:
            raise _abi.KvqError("ResNet.features needs the frames on a HIP device; there is no CPU path")
        x = x.to(torch.float32).contiguous()
        n, c, h, w_ = x.shape
        w = self._weights(x.device)
        half = _abi.torch_dtype(self.operand_dtype)
        y = self._stem(x, (n, 1, c, h, w_), (c * h * w_, 0, h * w_, w_, 1), w, half)
        y = kernels.pool_nd(y.unsqueeze(1), (1, 3, 3), (1, 2, 2), (0, 1, 1), True).squeeze(1)        # maxpool 3x3/2
        y32 = None
        for li, layer_mod in enumerate((self.layer1, self.layer2, self.layer3, self.layer4), 1):
            for bi, blk in enumerate(layer_mod):
                y, y32 = self._bottleneck(y, y32, w, f"l{li}.{bi}.", blk)
        return y, (y.to(torch.float32) if y32 is None else y32)

    def _stem(self, x, dims5, strides5, w, half):
        """conv1 7x7/2 + bn1 + ReLU on the fp32 input, frames addressed as (b, t) through element strides (b,t,c,h,w):
        -> 16-bit channels-last (b*t, ho, wo, 64).  Implicit GEMM over the input packed to 8 channels (no patch matrix: the
        147-column im2col of a 224x224 frame is 12x the frame), or the materialised im2col with KVQ_IMPLICIT_CONV=0."""
        B, T, c, h, w_ = dims5
        if IMPLICIT_CONV and c <= 8:
            x8 = kernels.pack_channels_last8(x, dims5, strides5, half)
            y = kernels.conv_implicit(x8.reshape(B * T, 1, h, w_, 8), w["stem8"], w["stem"][1], (1, 7, 7), (1, 2, 2), (0, 3, 3), True)
            return y.reshape(B * T, y.shape[2], y.shape[3], 64)
        wt, bias = w["stem"]
        if T == 1:
            a, (_, ho, wo) = kernels.im2col_nd(x, (B, c, 1, h, w_), (strides5[0], strides5[2], 0, strides5[3], strides5[4]), (1, 7, 7),
                                               (1, 2, 2), (0, 3, 3), half, wt.shape[1])
        else:
            a, (_, ho, wo) = self._stem_im2col(x, half, wt.shape[1])
        return kernels.conv_gemm(a, wt, bias, True).reshape(B * T, ho, wo, 64)

    @staticmethod
    def _stem_im2col(x, half, kpad):
        b, c, T, h, w_ = x.shape
        # frames are (b, t) pairs: im2col's batch stride walks t, so run it per batch element when b > 1
        if b == 1:
            return kernels.im2col_nd(x, (T, c, 1, h, w_), (h * w_, T * h * w_, 0, w_, 1), (1, 7, 7), (1, 2, 2), (0, 3, 3),
                                     half, kpad)
        parts = [kernels.im2col_nd(x[i], (T, c, 1, h, w_), (h * w_, T * h * w_, 0, w_, 1), (1, 7, 7), (1, 2, 2),
                                   (0, 3, 3), half, kpad) for i in range(b)]
        return torch.cat([p[0] for p in parts]), parts[0][1]


class TorchvisionResNet50(ResNet):
    """``torchvision.models.resnet50()`` as a parameter container: children in torchvision's order (conv1, bn1, relu, maxpool,
    layer1..4, avgpool, fc), so that ``nn.Sequential(*list(net.children())[:-2])`` — what the reference's ``CONTRIQUE_model``
    keeps (KSVQE_model.py:1630) — has the reference's ``state_dict`` keys.  Same blocks and HIP execution as ``ResNet``."""

    def __init__(self, operand_dtype=None):
        nn.Module.__init__(self)
        self.operand_dtype = _abi.dtype_code(operand_dtype or os.environ.get("KVQ_OPERAND_DTYPE", "fp16"))
        self.inplanes = 64
        self.conv1, self.bn1 = _Conv(3, 64, 7, 2, 3), _BN(64)
        self.relu, self.maxpool = nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.avgpool, self.fc = nn.AdaptiveAvgPool2d((1, 1)), nn.Linear(2048, 1000)
        self._wcache = None

    def forward(self, *a, **k):
        raise NotImplementedError("only the convolutional trunk is used (CONTRIQUE_model); call features()")


def resnet50(pretrained=False, progress=True, **kwargs):
    """``simpleVQA_model.py:307-325``.  ImageNet weights are a network download in the reference
    (``model_zoo.load_url``); offline this returns the randomly initialised network — load a checkpoint
    with ``load_state_dict`` (keys are the reference's)."""
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)
