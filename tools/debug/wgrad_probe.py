"""Per-layer timing of the weight-gradient kernel on the TANet-R50 shapes (16 frames)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitta_amd import conv as CV
from bench_conv import trunk_convs, time_it
d = torch.device("cuda:0")
n, tot_us, tot_fl, seen = 16, 0.0, 0.0, {}
for name, c, k, h, ksz, s in trunk_convs():
    key = (c, k, h, ksz, s)
    if key in seen:
        seen[key][0] += 1
        continue
    pad = ksz // 2
    g = CV.Geometry.forward(n, h, h, ksz, s, pad)
    x = torch.randn(c, n * h * h, device=d)
    dy = torch.randn(k, n * g.hy * g.wy, device=d)
    gw = torch.zeros(k, c, ksz, ksz, device=d)
    us = time_it(lambda: CV.wgrad(g, x, dy, gw, c, k), 20)
    fl = 2.0 * n * g.hy * g.wy * c * k * ksz * ksz
    seen[key] = [1, us, fl]
    print(f"{name:22s} C{c:5d} K{k:5d} H{h:3d} k{ksz} s{s}: {us:7.1f} us {fl/us/1e6:6.1f} TF", flush=True)
for cnt, us, fl in seen.values():
    tot_us += cnt * us
    tot_fl += cnt * fl
print(f"total {tot_us/1e3:.3f} ms, {tot_fl/tot_us/1e6:.1f} TF")
