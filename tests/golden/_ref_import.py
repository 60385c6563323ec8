"""Import the read-only reference (/root/reference) in the BUILD container only.

Used exclusively by ``make_golden.py`` to pin the oracle and emit fixtures.  Nothing
under ``tests/`` that runs on the GPU box imports this module: /root/reference does not
exist there.  Stub recipe = SURVEY.md App. C (third-party packages that are not
installed here and whose behaviour the hot path does not depend on).
"""
import os
import sys
import tempfile
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch
    import torch.nn as nn

    sys.dont_write_bytecode = True

    class DropPath(nn.Module):               # identity in eval; never active on the inference path
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    layers = _stub("timm.models.layers", DropPath=DropPath, trunc_normal_=trunc_normal_,
                   to_2tuple=lambda x: (x, x))
    registry = _stub("timm.models.registry", register_model=lambda f: f)
    tmodels = _stub("timm.models", layers=layers, registry=registry)
    _stub("timm", models=tmodels)

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tv_t = _stub("torchvision.transforms", Normalize=_Dummy, Compose=_Dummy, Resize=_Dummy,
                 CenterCrop=_Dummy, ToTensor=_Dummy, RandomCrop=_Dummy, RandomHorizontalFlip=_Dummy)
    tv_io = _stub("torchvision.io", write_video=None, write_png=None)
    tv_ops = _stub("torchvision.ops", roi_align=None, roi_pool=None)
    tv_models = _stub("torchvision.models")
    for n in ("vgg16", "vgg16_bn", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tv_models, n, lambda *a, **k: None)
    _stub("torchvision", transforms=tv_t, io=tv_io, ops=tv_ops, models=tv_models)
    _stub("thop", profile=None)
    _stub("cv2")
    bridge = types.SimpleNamespace(set_bridge=lambda *_: None)
    _stub("decord", bridge=bridge, VideoReader=None, cpu=None, gpu=None)
    _stub("turtle", forward=None)
    _stub("ftfy", fix_text=lambda s: s)


def import_reference():
    """Returns a namespace with the reference modules the oracle is pinned against."""
    import torch

    install_stubs()
    work = tempfile.mkdtemp(prefix="kvq_ref_cwd_")
    os.makedirs(os.path.join(work, "pretrained_weights"))
    torch.save({"state_dict": {}},
               os.path.join(work, "pretrained_weights", "swin_tiny_patch244_window877_kinetics400_1k.pth"))
    os.chdir(work)                      # import-time constructor loads a CWD-relative path (App. D-1)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import torch.utils.model_zoo as mz
    mz.load_url = lambda *a, **k: {}
    import contextlib
    import importlib
    import io
    import warnings
    ns = types.SimpleNamespace()
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")          # the reference prints whole state_dicts at import
        ns.swin = importlib.import_module("models.backbones.swin_backbone")
        ns.head = importlib.import_module("models.head")
        ns.simple = importlib.import_module("models.backbones.simpleVQA_model")
        ns.fd = importlib.import_module("datasets.fusion_datasets")
        ns.trainer = importlib.import_module("trainer")
        ns.model = importlib.import_module("models.model")
    return ns
