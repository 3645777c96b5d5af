"""Time `vitta_frames_resample_norm_f32` on the C2 clip (2 views x 8 frames of 320x240 -> 224^2, per-view multi-scale
crops) and the evaluation clip (8 frames, short edge -> 256, centre 224), beside the host PIL chain it replaces.

    python tools/bench_frames.py [--iters 200] [--size 320x240]

Kernel time = a captured graph of `--chain` back-to-back launches replayed, divided by the chain length (launch overhead
excluded).  Algorithmic bytes = the crops' bytes read once + the fp32 clip written once."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitta_amd import data_video as DV  # noqa: E402
from vitta_amd import frames as FR  # noqa: E402

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--chain", type=int, default=40)
    ap.add_argument("--size", default="320x240")
    ap.add_argument("--tile-rows", type=int, default=None)
    ap.add_argument("--clips", type=int, default=1, help="clips per launch (views and frames replicated): the asymptotic rate")
    a = ap.parse_args()
    if a.tile_rows:
        FR.TILE_ROWS = a.tile_rows
    w, h = (int(v) for v in a.size.split("x"))
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    T, views = 8, 2
    out = {}
    for mode in ("tta", "eval"):
        n = (views * T if mode == "tta" else T) * a.clips
        frames = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
        random.seed(3)
        if mode == "tta":
            specs = []
            for _ in range(views * a.clips):
                cw, ch, ow, oh = DV.sample_multiscale_crop((w, h), (224, 224))
                specs.append(FR.ViewSpec((ow, oh, cw, ch), (224, 224)))
            fpv = T
        else:
            specs, fpv = [FR.eval_view((w, h), 256, 224)] * a.clips, T
        t0 = time.perf_counter()
        plan = FR.FramePlan(specs, (224, 224), dev, MEAN, STD)
        torch.cuda.synchronize()
        plan_ms = (time.perf_counter() - t0) * 1e3
        d_frames = torch.from_numpy(frames).to(dev)
        buf = torch.empty(n * 3, 224, 224, device=dev)
        for _ in range(3):
            FR.resample_normalise(d_frames, plan, fpv, out=buf)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(a.chain):
                FR.resample_normalise(d_frames, plan, fpv, out=buf)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (a.iters * a.chain)
        read = sum(bw * bh * 3 for (_, _, bw, bh) in plan.boxes) * fpv
        written = buf.numel() * 4
        # host chain on the same frames (PIL crop/resize per frame + stack + normalise), one thread
        from PIL import Image
        pil = [Image.fromarray(f) for f in frames]
        reps = 5 if a.clips == 1 else 1
        t0 = time.perf_counter()
        for _ in range(reps):
            random.seed(3)
            if mode == "tta":
                DV.stack_to_tensor(DV.subgroup_multiscale_crop(pil, views * a.clips, T, 224), MEAN, STD)
            else:
                DV.stack_to_tensor([DV.center_crop(DV.scale_short_edge(f, 256), 224) for f in pil], MEAN, STD)
        host_ms = (time.perf_counter() - t0) * 1e3 / reps
        out[mode] = dict(frames=n, frame_size=[w, h], kernel_us=us, algorithmic_bytes=read + written,
                         achieved_GBps=(read + written) / us * 1e-3, frac_of_8TBps=(read + written) / us * 1e-3 / 8000.0,
                         workgroups=n * -(-224 // plan.tile_rows), tile_rows=plan.tile_rows, lds_rows=plan.lds_rows,
                         plan_build_host_ms=plan_ms, upload_bytes_u8=int(frames.nbytes), upload_bytes_f32_clip=int(written),
                         host_pil_chain_ms=host_ms)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
