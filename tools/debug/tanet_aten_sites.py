"""Which ATen operators launch kernels in one overlapped TANet-R50 iteration (adaptation step + evaluation forward, 2 x 8 x 224^2),
grouped by the vitta_amd line that caused them (what is left outside the hand-written kernels)."""
import collections
import os
import sys
import tempfile
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import helpers as H  # noqa: E402
from vitta_amd import tta  # noqa: E402

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
model = H.build_tanet(101, 8, 0)
bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
mp, vp = H.write_stat_files(tmp, [np.zeros(b.num_features, np.float32) for b in bn2d], [np.ones(b.num_features, np.float32) for b in bn2d])
args = H.tanet_args(tmp, clip_length=8, input_size=224, spatiotemp_mean_clean_file=mp, spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=1e-6)
adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(dev), args)
x = torch.randn(1, 2 * 8 * 3, 224, 224, device=dev)
e = torch.randn(1, 8 * 3, 224, 224, device=dev)
tin, ein = adapter.shape_tta_input(x), adapter.shape_eval_input(e)
for _ in range(2):
    adapter.set_adapt_mode()
    adapter.step(tin, ein)
torch.cuda.synchronize()
sites = collections.defaultdict(lambda: [0, 0])
SKIP = ("empty", "empty_like", "empty_strided", "view", "_unsafe_view", "reshape", "as_strided", "detach", "alias", "t", "transpose", "permute",
        "expand", "slice", "select", "squeeze", "unsqueeze", "_local_scalar_dense", "record_stream", "is_pinned", "lift_fresh", "unbind",
        "split", "narrow", "sym_size", "sym_stride", "numel", "size", "stride", "dim", "is_contiguous", "new_empty", "new_empty_strided")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        t = out if isinstance(out, torch.Tensor) else (out[0] if isinstance(out, (tuple, list)) and out and isinstance(out[0], torch.Tensor) else None)
        if name not in SKIP and t is not None and t.is_cuda:
            st = traceback.extract_stack()
            fr = next((f for f in reversed(st) if "vitta_amd" in f.filename), None)
            where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:80]}" if fr else "(autograd engine / outside vitta_amd)"
            sites[(func.__name__, where)][0] += 1
            sites[(func.__name__, where)][1] += t.numel() * t.element_size()
        return out


adapter.set_adapt_mode()
with Log():
    adapter.step(tin, ein)
    torch.cuda.synchronize()
tot = 0
for (name, where), (n, by) in sorted(sites.items(), key=lambda kv: -kv[1][0]):
    tot += n
    print(f"{n:4d} x {by / n / 1e6:8.3f} MB  {name:28s} {where}")
print("ATen operator calls with a device result:", tot)
