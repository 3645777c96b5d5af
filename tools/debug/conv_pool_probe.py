"""What the pooled-means epilogue (VITTA_CONV_POOL) adds to conv1 of a bottleneck, per trunk shape, against the stand-alone pooling
launch it replaces (hipGraph-replay timing as tools/bench_conv.py)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_conv import time_it  # noqa: E402
from vitta_amd import _lib, conv as CV  # noqa: E402
from vitta_amd.ops import _ptr4  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
for n in (16, 8):
    for c, k, h in ((64, 64, 56), (256, 64, 56), (256, 128, 56), (512, 128, 28), (512, 256, 28), (1024, 256, 14), (1024, 512, 14), (2048, 512, 7)):
        x = torch.randn(c, n * h * h, device=dev)
        w = torch.randn(k, c, 1, 1, device=dev) * c ** -0.5
        wp = CV.make_pack(CV.pack_fwd(w))
        y = torch.empty(k, n * h * h, device=dev)
        bn = [torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev), torch.randn(k, device=dev), torch.rand(k, device=dev) + 0.5]
        geom = CV.Geometry.forward(n, h, h, 1, 1, 0)
        pool = torch.zeros(n, k, dtype=torch.int64, device=dev)
        pooled = torch.empty(n // 8, k, 8, device=dev)
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        t_plain = time_it(lambda: CV.launch(geom, x, wp, y, c, k), 10)
        t_pool = time_it(lambda: CV.launch(geom, x, wp, y, c, k, epi_bn=bn, pool=pool), 10)
        t_kernel = time_it(lambda: _lib.check(L.vitta_tam_pool_cm_f32(C.c_void_p(y.data_ptr()), _ptr4(*bn), 1e-5, k, n // 8, 8, h * h,
                                                                      C.c_void_p(pooled.data_ptr()), st()), "pool"), 10)
        print(f"frames {n:2d} {c:4d} -> {k:3d} @ {h:2d}^2: conv {t_plain:6.1f} us, with the pooled-means epilogue {t_pool:6.1f} (+{t_pool - t_plain:4.1f}), "
              f"pooling launch {t_kernel:5.1f} us", flush=True)
