"""Workgroup-count quantisation of the convolution launches: the same layer at a pixel count that gives exactly 768
workgroups (3 per CU x 256 CUs) and at the trunk's own size (784)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitta_amd import conv as CV

d = torch.device("cuda:0")


def t(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (c, k, h, w, ksz, n) in [(64, 64, 48, 64, 3, 16), (64, 64, 56, 56, 3, 16), (64, 64, 40, 64, 3, 16), (256, 256, 12, 16, 3, 16), (256, 256, 14, 14, 3, 16),
                             (256, 64, 48, 64, 1, 16), (256, 64, 56, 56, 1, 16), (64, 256, 48, 64, 1, 16), (64, 256, 56, 56, 1, 16),
                             (1024, 256, 12, 16, 1, 16), (1024, 256, 14, 14, 1, 16), (64, 64, 24, 32, 3, 16), (64, 64, 96, 128, 3, 16)]:
    pad = ksz // 2
    g = CV.Geometry.forward(n, h, w, ksz, 1, pad)
    x = torch.randn(c, n * h * w, device=d)
    wt = torch.randn(k, c, ksz, ksz, device=d) * (c * ksz * ksz) ** -0.5
    wf = CV.pack_fwd(wt)
    y = torch.empty(k, n * h * w, device=d)
    fl = 2.0 * n * h * w * c * k * ksz * ksz
    for ks in (0, 1):
        us = t(lambda: CV.launch(g, x, wf, y, c, k, ksplit=ks))
        print(f"C{c} K{k} {h}x{w} k{ksz} tiles {((n*h*w+63)//64)*(k//64)} ksplit {ks}: {us:.1f} us {fl/us/1e6:.1f} TF", flush=True)
