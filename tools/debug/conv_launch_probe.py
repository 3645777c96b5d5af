"""What a convolution launch costs beyond its workgroups' lifetimes: one shape, repeated back to back inside a hipGraph, with
none / one / two output tensors and the full conv3 epilogue.  python tools/debug/conv_launch_probe.py [n c k h ksz]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_conv import time_it  # noqa: E402
from vitta_amd import conv as CV  # noqa: E402

dev = torch.device("cuda:0")


def probe(n, c, k, h, ksz, reps=40):
    x = torch.randn(c, n * h * h, device=dev)
    w = torch.randn(k, c, ksz, ksz, device=dev) * (c * ksz * ksz) ** -0.5
    geom = CV.Geometry.forward(n, h, h, ksz, 1, ksz // 2)
    P = n * h * h
    wp = CV.make_pack(CV.pack_fwd(w))
    y, raw, res = torch.empty(k, P, device=dev), torch.empty(k, P, device=dev), torch.randn(k, P, device=dev)
    bn = [torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev), torch.randn(k, device=dev), torch.rand(k, device=dev) + 0.5]
    sh, s1, s2 = torch.zeros(k, device=dev), torch.zeros(k, device=dev), torch.zeros(k, device=dev)
    # many distinct output buffers: a launch must not find its own output lines still dirty in L2 from the previous repeat
    ys = [torch.empty(k, P, device=dev) for _ in range(8)]
    raws = [torch.empty(k, P, device=dev) for _ in range(8)]
    it = [0]

    def plain():
        CV.launch(geom, x, wp, ys[it[0] % 8], c, k)
        it[0] += 1

    def two():
        CV.launch(geom, x, wp, ys[it[0] % 8], c, k, y_raw=raws[it[0] % 8])
        it[0] += 1

    def full():
        CV.launch(geom, x, wp, ys[it[0] % 8], c, k, flags=CV.CONV_EPI_APPLY | CV.CONV_EPI_RELU | CV.CONV_RES | CV.CONV_STATS, y_raw=raws[it[0] % 8],
                  res=res, epi_bn=bn, stats=(sh, s1, s2))
        it[0] += 1

    def same():
        CV.launch(geom, x, wp, y, c, k)

    out = {name: time_it(fn, reps) for name, fn in (("one output, rotating buffers", plain), ("two outputs", two),
                                                    ("conv3 epilogue, two outputs", full), ("one output, same buffer", same))}
    mb = 4e-6 * k * P
    print(f"{c}->{k} {ksz}x{ksz} at {h}^2 x {n} frames (output {mb:.1f} MB): " + "  ".join(f"{a}: {b:.1f} us" for a, b in out.items()))


if __name__ == "__main__":
    if len(sys.argv) > 5:
        probe(*map(int, sys.argv[1:6]))
    else:
        for a in ((16, 256, 1024, 14, 1), (16, 512, 2048, 7, 1), (16, 1024, 256, 14, 1), (16, 64, 256, 56, 1), (16, 256, 64, 56, 1),
                  (16, 256, 256, 14, 3), (16, 512, 512, 7, 3), (16, 64, 64, 56, 3)):
            probe(*a)
