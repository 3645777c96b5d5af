import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitta_amd import conv as CV
d = torch.device("cuda:0")
for spec in sys.argv[1:]:
    c, k, h, w, ksz = [int(v) for v in spec.split(",")]
    n, pad = 16, ksz // 2
    g = CV.Geometry.forward(n, h, w, ksz, 1, pad)
    x = torch.randn(c, n * h * w, device=d)
    wf = CV.pack_fwd(torch.randn(k, c, ksz, ksz, device=d) * (c * ksz * ksz) ** -0.5)
    y = torch.empty(k, n * h * w, device=d)
    fl = 2.0 * n * h * w * c * k * ksz * ksz
    for _ in range(3):
        CV.launch(g, x, wf, y, c, k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        CV.launch(g, x, wf, y, c, k)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"{spec}: {us:.1f}us {fl/us/1e6:.0f}TF")
