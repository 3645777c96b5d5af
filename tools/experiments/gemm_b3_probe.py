"""gemm_b3.hip against the exact-fp32 kernel on the Video Swin-B shapes (C3: 2 views x 16 frames), plus large square-ish
shapes where neither launch overheads nor short K matter: us and TF per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vitta_amd import ops
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [(25088, 384, 128), (25088, 512, 128), (25088, 128, 512), (6272, 768, 256), (6272, 1024, 256), (6272, 256, 1024),
          (1568, 1536, 512), (1568, 2048, 512), (1568, 512, 2048), (392, 3072, 1024), (392, 4096, 1024), (392, 1024, 4096),
          (8192, 8192, 512), (8192, 8192, 4096), (16384, 1024, 4096)]
for m, n, k in shapes:
    a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * k ** -0.5
    y = torch.empty(m, n, device=dev)
    op = ops.B3Operand(w)
    gf = 2.0 * m * n * k / 1e9
    t32 = timed(lambda: ops.gemm_nt(a, w, out=y))
    tb3 = timed(lambda: ops.gemm_nt(a, op, out=y))
    print(f"M={m:6d} N={n:5d} K={k:5d} {gf:8.2f} GF | fp32 {t32:8.1f} us {gf / t32 * 1e3:6.1f} TF | b3 {tb3:8.1f} us {gf / tb3 * 1e3:6.1f} TF", flush=True)
