"""Per-layer timing of the hand-written convolutions on the TANet-R50 shapes (adapt pass: 16 frames of 224^2), forward
and data gradient, beside the vendor library's forward on the same shape (reference point, not product).

    python tools/bench_conv.py [--frames 16] [--out gpurun_out/conv_bench.json]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitta_amd import conv as CV  # noqa: E402

PEAK = 157.3  # TFLOP/s fp32 matrix (MI355X_MICROARCH.md)


def trunk_convs():
    """(name, C, K, H_in, k, stride, count) of the bottleneck convolutions of ResNet-50 at 224^2 (after the stem: 56^2)."""
    out = []
    h, inpl = 56, 64
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
        for b in range(blocks):
            s = stride if b == 0 else 1
            out.append((f"layer{li}.{b}.conv1", inpl, planes, h, 1, 1))
            out.append((f"layer{li}.{b}.conv2", planes, planes, h, 3, s))
            ho = CV.out_size(h, 3, s, 1)
            out.append((f"layer{li}.{b}.conv3", planes, planes * 4, ho, 1, 1))
            if b == 0:
                out.append((f"layer{li}.{b}.downsample", inpl, planes * 4, h, 1, s))
            h, inpl = ho, planes * 4
    return out


def time_it(fn, reps):
    """us per call of fn, device time: `reps` calls captured into one hipGraph (eager launches through ctypes are host-bound
    below ~20 us per call), the replay timed with events."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    graph.replay()
    graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps) * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--out", default="gpurun_out/conv_bench.json")
    ap.add_argument("--tiles", default="", help="comma list of BMxBN to sweep instead of the library's choice")
    ap.add_argument("--no-vendor", action="store_true")
    ap.add_argument("--arith", default="", help="comma list of arithmetic forms to time side by side (f32,b3); default: the library's")
    ap.add_argument("--only", default="", help="substring filter on layer names")
    opt = ap.parse_args()
    d = torch.device("cuda:0")
    n = opt.frames
    rows, seen = [], {}
    tiles = [0] + [(int(t.split("x")[0]) << 16) | int(t.split("x")[1]) for t in opt.tiles.split(",") if t]
    ariths = [a for a in opt.arith.split(",") if a] or [CV.ARITH]
    for name, c, k, h, ksz, s in trunk_convs():
        if opt.only and opt.only not in name:
            continue
        key = (c, k, h, ksz, s)
        if key in seen:
            seen[key]["count"] += 1
            continue
        pad = ksz // 2
        gf = CV.Geometry.forward(n, h, h, ksz, s, pad)
        ho = gf.hy
        x = torch.randn(c, n * h * h, device=d)
        w = torch.randn(k, c, ksz, ksz, device=d) * (c * ksz * ksz) ** -0.5
        wf, wb = CV.pack_fwd(w), CV.pack_bwd(w)
        y = torch.empty(k, n * ho * ho, device=d)
        gx = torch.empty(c, n * h * h, device=d)
        flops = 2.0 * n * ho * ho * c * k * ksz * ksz
        row = dict(name=name, C=c, K=k, H=h, k=ksz, stride=s, count=1, gflop=flops / 1e9)
        for tile, arith in [(t, a) for t in tiles for a in ariths]:
            CV.ARITH = arith
            tag = ("" if tile == 0 else f"_{tile >> 16}x{tile & 0xffff}") + (f"_{arith}" if len(ariths) > 1 else "")
            if tile and (k % (tile & 0xffff) or c % (tile & 0xffff)):
                continue
            t_f = time_it(lambda: CV.launch(gf, x, wf, y, c, k, tile=tile), opt.reps)
            geoms = CV.Geometry.dgrad(n, h, h, ksz, s, pad)
            gdst = torch.empty(c, n * ho * ho, device=d) if (s == 2 and ksz == 1) else gx

            wbp = CV.make_pack(wb)
            gm = CV.Geometry.dgrad_merged(n, h, h, ksz, s, pad) if (s == 2 and ksz == 3 and tile == 0) else None
            if gm is not None and not CV.merged_dgrad_supported(gm, wbp, k, c):
                gm = None

            def dgrad():
                if gm is not None:  # stride 2: the four parity classes in one launch
                    CV.launch(gm, y, wbp, gdst, k, c)
                    return
                for g in geoms:
                    CV.launch(g, y, wb, gdst, k, c, tile=tile)
            t_b = time_it(dgrad, opt.reps)
            row[f"fwd_us{tag}"], row[f"fwd_tf{tag}"] = t_f, flops / t_f / 1e6
            row[f"dgrad_us{tag}"], row[f"dgrad_tf{tag}"] = t_b, flops / t_b / 1e6
        if not opt.no_vendor:
            xn = torch.randn(n, c, h, h, device=d)
            t_v = time_it(lambda: F.conv2d(xn, w, stride=s, padding=pad), opt.reps)
            row["vendor_fwd_us"], row["vendor_fwd_tf"] = t_v, flops / t_v / 1e6
        seen[key] = row
        rows.append(row)
        print({kk: (round(v, 1) if isinstance(v, float) else v) for kk, v in row.items()}, flush=True)
    # the stem (stem_conv.hip) beside the library's 7x7 forward
    xs = torch.randn(n, 3, 224, 224, device=d)
    ws = torch.randn(64, 3, 7, 7, device=d) * 147 ** -0.5
    wps = CV.pack_stem(ws)
    sflops = 2.0 * n * 112 * 112 * 147 * 64
    t_s = time_it(lambda: CV.stem_conv(xs, wps), opt.reps)
    stem = dict(name="stem.conv1", gflop=sflops / 1e9, fwd_us=t_s, fwd_tf=sflops / t_s / 1e6)
    if not opt.no_vendor:
        t_v = time_it(lambda: F.conv2d(xs, ws, stride=2, padding=3), opt.reps)
        stem["vendor_fwd_us"], stem["vendor_fwd_tf"] = t_v, sflops / t_v / 1e6
    print({kk: (round(v, 1) if isinstance(v, float) else v) for kk, v in stem.items()}, flush=True)
    tot = lambda f: sum(r[f] * r["count"] for r in rows if f in r)
    summary = dict(frames=n, gflop=tot("gflop"), peak_tf=PEAK, vendor_fwd_ms=tot("vendor_fwd_us") / 1e3)
    for a in ariths:
        t = f"_{a}" if len(ariths) > 1 else ""
        summary.update({f"fwd_ms{t}": tot(f"fwd_us{t}") / 1e3, f"dgrad_ms{t}": tot(f"dgrad_us{t}") / 1e3,
                        f"fwd_tf{t}": tot("gflop") / max(tot(f"fwd_us{t}"), 1e-9) * 1e3,
                        f"dgrad_tf{t}": tot("gflop") / max(tot(f"dgrad_us{t}"), 1e-9) * 1e3})
    print(summary)
    os.makedirs(os.path.dirname(opt.out) or ".", exist_ok=True)
    json.dump(dict(summary=summary, stem=stem, rows=rows), open(opt.out, "w"), indent=1)


if __name__ == "__main__":
    main()
