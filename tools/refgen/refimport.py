"""Import harness for the REFERENCE implementation (/root/reference) -- build container only.

This file imports the reference's Python modules in-process so golden vectors can be captured from
the reference's own code.  Nothing of the reference is copied: the modules are loaded from where
they lie, with small stand-ins for third-party packages that are not installed here (torchvision,
timm, mmcv, mmaction, decord, tensorboardX, cv2) and with `.cuda()` neutralised (no GPU here).
The torchvision ResNet-50 stand-in is this repo's own restatement (vitta_amd.resnet) -- the trunk's
arithmetic lives in torchvision==0.8.2, which is not part of the reference checkout, so parity of
the trunk itself is unpinned.

Never shipped to / used on the GPU box: /root/reference does not exist there.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE = os.environ.get("VITTA_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Anything:
    """Callable / subclassable placeholder for names that are imported but never exercised."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]
        return type(name, (_Anything,), {})


def _mod(name, **attrs):
    m = _StubModule(name)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


class DropPath(nn.Module):
    """timm==0.6.7 drop_path semantics (per-sample bernoulli(keep) / keep); arithmetic not under
    /root/reference -> parity of the mask convention is unpinned, it only matters with masks injected."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def install():
    """Install stand-ins + CPU patches, put the reference first on sys.path."""
    if getattr(install, "_done", False):
        return
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from vitta_amd import resnet as my_resnet  # noqa: E402  (our torchvision.models.resnet50 stand-in)

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tv = _mod("torchvision")
    _mod("torchvision.transforms", Compose=Compose)
    tvm = _mod("torchvision.models", resnet50=my_resnet.resnet50, ResNet=my_resnet.ResNet)
    _mod("torchvision.models.resnet", Bottleneck=my_resnet.Bottleneck, ResNet=my_resnet.ResNet)
    _mod("torchvision.models.video")
    _mod("torchvision.models.video.resnet")
    _mod("torchvision.models.utils")
    _mod("torchvision.datasets")
    _mod("torchvision.datasets.video_utils")
    _mod("torchvision.io")
    _mod("torchvision.ops")
    _mod("torchvision.utils")

    _mod("timm")
    _mod("timm.models", create_model=_Anything())
    _mod("timm.models.layers", DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_,
         drop_path=_Anything(), to_2tuple=lambda x: (x, x))
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod("mmcv")
    _mod("mmcv.runner", load_checkpoint=_Anything())
    _mod("mmcv.cnn", normal_init=lambda *a, **k: None)
    _mod("mmcv.fileio")
    _mod("mmcv.parallel")
    _mod("mmcv.utils")
    _mod("mmaction")
    _mod("mmaction.utils", get_root_logger=lambda *a, **k: __import__("logging").getLogger("mmaction"))
    _mod("decord")
    _mod("tensorboardX", SummaryWriter=_Anything)
    _mod("cv2")

    # numpy aliases the reference's pinned numpy 1.19.5 still had (transforms_backup.py:505,527,531,689 use np.int)
    import numpy as _np
    for _alias, _typ in (("int", int), ("float", float), ("bool", bool)):
        if _alias not in _np.__dict__:
            setattr(_np, _alias, _typ)

    # no GPU in the build container: .cuda() / .to('cuda:0') become no-ops
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    _to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, (str, torch.device)) and "cuda" in str(x)) else x for x in a)
        if "device" in k and "cuda" in str(k["device"]):
            k["device"] = "cpu"
        return _to(self, *a, **k)

    torch.Tensor.to = to_cpu
    _mto = nn.Module.to

    def mto_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, (str, torch.device)) and "cuda" in str(x)) else x for x in a)
        return _mto(self, *a, **k)

    nn.Module.to = mto_cpu
    _device = torch.device

    # reference modules parse sys.argv at import time (baselines/setup_baseline.py:9, baselines/shot.py:38)
    sys.argv = [sys.argv[0]]
    # `utils`, `corpus`, `models` must resolve to the REFERENCE packages.  The reference's are namespace
    # packages (no __init__.py) while this repo's root-level drop-in shims are regular packages, and a
    # regular package anywhere on sys.path wins over a namespace portion: take the repo root OFF sys.path
    # (vitta_amd is already imported; its sub-modules resolve through vitta_amd.__path__).
    import vitta_amd  # noqa: F401
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    sys.path.insert(0, REFERENCE)
    for name in list(sys.modules):
        if name.split(".")[0] in ("utils", "corpus", "models", "baselines", "datasets_"):
            del sys.modules[name]
    install._done = True


if __name__ == "__main__":
    install()
    import corpus.main_eval  # noqa: F401
    from models.tanet_models.tanet import TSN
    m = TSN(11, 8, "RGB", base_model="resnet50", consensus_type="avg", tam=True, print_spec=False)
    print("reference imports ok; TSN params:", sum(p.numel() for p in m.parameters()))
