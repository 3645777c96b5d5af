"""Main-loop efficiency probe: long-K shapes where per-block fixed costs vanish (not a product path)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from vitta_amd import conv as CV
d = torch.device("cuda:0")
n = 16


def run(c, k, h, ksz, tile, flags=0):
    pad = ksz // 2
    gf = CV.Geometry.forward(n, h, h, ksz, 1, pad)
    x = torch.randn(c, n * h * h, device=d)
    w = torch.randn(k, c, ksz, ksz, device=d) * (c * ksz * ksz) ** -0.5
    wf = CV.pack_fwd(w)
    y = torch.empty(k, n * h * h, device=d)
    for _ in range(2):
        CV.launch(gf, x, wf, y, c, k, tile=tile, flags=flags)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        CV.launch(gf, x, wf, y, c, k, tile=tile, flags=flags)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 5 * 1e3
    fl = 2.0 * n * h * h * c * k * ksz * ksz
    print(f"C{c} K{k} H{h} k{ksz} tile {tile >> 16}x{tile & 0xffff}: {us:8.1f} us {fl / us / 1e6:6.1f} TF", flush=True)


for tile in [(128 << 16) | 128, (128 << 16) | 64, (64 << 16) | 64, (64 << 16) | 32]:
    run(4096, 512, 56, 1, tile)
    run(4096, 512, 28, 1, tile)
    run(512, 512, 56, 3, tile)
