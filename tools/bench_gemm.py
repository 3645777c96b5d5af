"""Per-shape timing of the hand-written dense kernel (csrc/gemm.hip) against torch's library GEMM (rocBLAS / hipBLASLt)
on the Video Swin-B shapes of BASELINE config 3 (2 views x 16 frames x 224^2): python tools/bench_gemm.py [--out f.json]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from vitta_amd import ops

p = argparse.ArgumentParser()
p.add_argument("--out", default=None)
p.add_argument("--reps", type=int, default=20)
p.add_argument("--views", type=int, default=2)
p.add_argument("--tanet", action="store_true", help="the pointwise-convolution shapes of the TANet trunk at 16 frames, as GEMMs")
opt = p.parse_args()
dev = torch.device("cuda:0")
PEAK = 157.3


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


if opt.tanet:
    for (m, n, k) in [(50176, 256, 64), (50176, 64, 256), (12544, 512, 128), (12544, 128, 512), (3136, 1024, 256), (3136, 256, 1024),
                      (784, 2048, 512), (784, 512, 2048), (50176, 128, 256), (12544, 256, 512), (3136, 512, 1024)]:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev)
        y = torch.empty(m, n, device=dev)
        gf = 2.0 * m * n * k / 1e9
        line = f"M={m:6d} N={n:5d} K={k:5d} {gf:5.2f} GF"
        for tile in (2, 3):
            ops.GEMM_TILE = tile
            us = timed(lambda: ops.gemm_nt(a, w, out=y), opt.reps)
            line += f" | tile{tile} {us:6.1f} us {gf / us * 1e3:6.1f} TF"
        print(line, flush=True)
    sys.exit(0)
rows = []
tok0 = opt.views * 8 * 56 * 56
for stage, (c, blocks) in enumerate([(128, 2), (256, 2), (512, 18), (1024, 2)]):
    m = tok0 >> (2 * stage)
    for name, n, k, mode in [("qkv", 3 * c, c, 0), ("proj", c, c, 0), ("fc1+gelu", 4 * c, c, 1), ("fc2", c, 4 * c, 0),
                             ("fc2.dgrad*gelu'", 4 * c, c, 2)]:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) * k ** -0.5
        b = torch.randn(n, device=dev)
        aux = torch.randn(m, n, device=dev) if mode == 2 else None
        pre = torch.empty(m, n, device=dev) if mode == 1 else None
        y = torch.empty(m, n, device=dev)
        gf = 2.0 * m * n * k / 1e9
        row = dict(stage=stage, layer=name, M=m, N=n, K=k, gflop=gf, blocks=blocks)
        for tile in (1, 2, 3, 0):
            ops.GEMM_TILE = tile
            us = timed(lambda: ops.gemm_nt(a, w, b if mode != 2 else None, mode=mode, aux=aux, pre=pre, out=y), opt.reps)
            row[f"tile{tile}_us"] = us
            row[f"tile{tile}_tf"] = gf / us * 1e3
        wb = w.to(torch.bfloat16)
        for tile in (1, 2, 3):
            ops.GEMM_TILE = tile
            us = timed(lambda: ops.gemm_nt(a, wb, b if mode != 2 else None, mode=mode, aux=aux, pre=pre, out=y), opt.reps)
            row[f"bf16_tile{tile}_us"] = us
        ops.GEMM_TILE = 0
        row["bf16_auto_us"] = timed(lambda: ops.gemm_nt(a, wb, b if mode != 2 else None, mode=mode, aux=aux, pre=pre, out=y), opt.reps)
        if mode == 0:
            lib_fn = lambda: F.linear(a, w, b)
        elif mode == 1:
            lib_fn = lambda: F.gelu(F.linear(a, w, b))
        else:
            lib_fn = lambda: torch.mm(a, w.t()) * aux
        row["library_us"] = timed(lib_fn, opt.reps)
        row["library_tf"] = gf / row["library_us"] * 1e3
        rows.append(row)
        print(f"s{stage} {name:16s} M={m:6d} N={n:5d} K={k:5d} {gf:7.2f} GF | 128x128 {row['tile1_us']:7.1f} us {row['tile1_tf']:6.1f} TF"
              f" | 64x128 {row['tile2_us']:7.1f} us {row['tile2_tf']:6.1f} TF | 64x64 {row['tile3_us']:7.1f} us {row['tile3_tf']:6.1f} TF | auto {row['tile0_us']:7.1f} us {row['tile0_tf']:6.1f} TF | library"
              f" {row['library_us']:7.1f} us {row['library_tf']:6.1f} TF | bf16 operands 128x128 {row['bf16_tile1_us']:6.1f} 64x128"
              f" {row['bf16_tile2_us']:6.1f} 64x64 {row['bf16_tile3_us']:6.1f} auto {row['bf16_auto_us']:6.1f} us", flush=True)
tot = lambda key: sum(r[key] * r["blocks"] for r in rows)
summary = dict(bf16_auto_ms=tot("bf16_auto_us") / 1e3, ours_auto_ms=tot("tile0_us") / 1e3, library_ms=tot("library_us") / 1e3, gflop=tot("gflop"), peak_tf=PEAK,
               ours_tf=tot("gflop") / tot("tile0_us") * 1e3, library_tf=tot("gflop") / tot("library_us") * 1e3)
print(json.dumps(summary))
if opt.out:
    json.dump(dict(summary=summary, rows=rows), open(opt.out, "w"), indent=1)
