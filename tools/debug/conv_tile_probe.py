"""Asymptotic rate of each tile configuration on a problem large enough to hide launch ramp and tile quantisation."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitta_amd import conv as CV
d = torch.device("cuda:0")


def t(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (c, k, h, w, ksz, n) in [(256, 256, 96, 128, 1, 16), (256, 256, 96, 128, 3, 16), (64, 64, 192, 256, 3, 16), (1024, 256, 48, 64, 1, 16)]:
    pad = ksz // 2
    g = CV.Geometry.forward(n, h, w, ksz, 1, pad)
    x = torch.randn(c, n * h * w, device=d)
    wt = torch.randn(k, c, ksz, ksz, device=d) * (c * ksz * ksz) ** -0.5
    wf = CV.pack_fwd(wt)
    y = torch.empty(k, n * h * w, device=d)
    fl = 2.0 * n * h * w * c * k * ksz * ksz
    for bm, bn in ((64, 64), (128, 64), (128, 128)):
        if k % bn:
            continue
        us = t(lambda: CV.launch(g, x, wf, y, c, k, tile=(bm << 16) | bn, ksplit=1))
        print(f"C{c} K{k} {h}x{w} k{ksz} tile {bm}x{bn} wgs {((n*h*w+bm-1)//bm)*(k//bn)}: {us:.1f} us {fl/us/1e6:.1f} TF", flush=True)
