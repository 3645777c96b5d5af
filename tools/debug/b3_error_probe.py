"""Error of the convolution forms against fp64: max |err| / max |ref| per (form, arithmetic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from vitta_amd import conv as CV
d = torch.device("cuda:0")
cases = [("pointwise", 32, 256, 256, 16, 1, 1), ("patch 3x3", 32, 128, 128, 16, 3, 1), ("gather 3x3 s2", 32, 128, 128, 16, 3, 2),
         ("gather 1x1 s2", 32, 256, 512, 16, 1, 2), ("gather 3x3 s2 big", 16, 128, 128, 56, 3, 2), ("gather 3x3 s2 4px", 32, 512, 512, 4, 3, 2)]
for name, n, c, k, h, ksz, s in cases:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, ksz, ksz, generator=g) * (c * ksz * ksz) ** -0.5
    ref = F.conv2d(x.double(), w.double(), stride=s, padding=ksz // 2)
    geom = CV.Geometry.forward(n, h, h, ksz, s, ksz // 2)
    out = []
    for arith in ("f32", "b3"):
        CV.ARITH = arith
        y = torch.zeros(k, n * geom.hy * geom.wy, device=d)
        CV.KERNEL_COUNTS = {}
        CV.launch(geom, CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k)
        kid = list(CV.KERNEL_COUNTS)[0]
        CV.KERNEL_COUNTS = None
        e = (CV.from_cm(y, n, geom.hy, geom.wy).cpu().double() - ref).abs()
        out.append(f"{arith}[k{kid}] max {e.max().item() / ref.abs().max().item():.2e} rms {e.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.2e}")
    print(f"{name:22s}", " | ".join(out))
