"""What this box's HBM delivers to plain vendor kernels, beside the moments kernel on the same bytes (SURVEY 8d:
"verify the 8 TB/s with a device-to-device copy test on the box").

    python tools/bench_hbm.py

(a) device-to-device copy of a 2.85 GB fp32 buffer (bytes moved = 2 x size), (b) a read-only reduction of it
(torch.sum: rocPRIM), (c) `vitta_moments_batched_f32` on the C2 layer shapes x 16 videos (the bench's streaming launch:
2.85 GB read once; and the same launch on bfloat16 features, 1.43 GB).  All timed with events over 20 launches after 3 warm-ups; GB/s of bytes moved."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitta_amd import ops  # noqa: E402


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
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    # hooked BN2d outputs of TANet-R50 layer3 / layer4 for 2 x 8 frames at 224^2, 16 videos' worth of frames
    shapes = []
    for blocks, planes, hw in ((6, 256, 14), (3, 512, 7)):
        for b in range(blocks):
            first_hw = hw * 2 if b == 0 else hw  # conv1/bn1 of a stage's first block still runs at the input resolution
            shapes += [(16 * 16, planes, first_hw * first_hw, 0), (16 * 16, planes, hw * hw, 0), (16 * 16, planes * 4, hw * hw, 0)]
            if b == 0:
                shapes.append((16 * 16, planes * 4, hw * hw, 0))
    numel = sum(o * c * i for o, c, i, _ in shapes)
    feats = [torch.randn(o, c, i, device=dev) for o, c, i, _ in shapes]
    plan = ops.StatPlan(shapes, dev, target_blocks=4096)
    flat = torch.empty(numel, device=dev)
    dst = torch.empty_like(flat)
    out = {"bytes": numel * 4}
    ms = timed(lambda: dst.copy_(flat))
    out["d2d_copy"] = dict(ms=ms, GBps_moved=2 * numel * 4 / ms * 1e-6)
    ms = timed(lambda: flat.sum())
    out["torch_sum_read_only"] = dict(ms=ms, GBps=numel * 4 / ms * 1e-6)
    ms = timed(lambda: plan.partials(feats))
    out["moments_batched"] = dict(ms=ms, GBps=numel * 4 / ms * 1e-6, frac_of_8TBps=numel * 4 / ms * 1e-6 / 8000.0)
    fb = [f.to(torch.bfloat16) for f in feats]
    ms = timed(lambda: plan.partials(fb))
    out["moments_batched_bf16"] = dict(ms=ms, bytes=numel * 2, GBps=numel * 2 / ms * 1e-6,
                                       frac_of_8TBps=numel * 2 / ms * 1e-6 / 8000.0)
    del fb
    if "--sweep" in sys.argv:
        out["sweep_target_blocks"] = {}
        for tb in (1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768):
            pl = ops.StatPlan(shapes, dev, target_blocks=tb)
            ms = timed(lambda: pl.partials(feats))
            out["sweep_target_blocks"][tb] = round(numel * 4 / ms * 1e-6)
        del feats, flat, dst
        # channels-last (Video Swin-B, C3): 42 hooked LayerNorm outputs, 16 videos' worth of rows
        ln = [(16 * 3136, 512, 1, 1)] * 36 + [(16 * 784, 2048, 1, 1)] + [(16 * 784, 1024, 1, 1)] * 5
        n_ln = sum(o * c for o, c, _, _ in ln)
        f_ln = [torch.randn(o, c, device=dev) for o, c, _, _ in ln]
        out["sweep_target_blocks_nhwc"] = {"bytes": n_ln * 4}
        for tb in (1024, 2048, 4096, 8192, 16384, 32768):
            pl = ops.StatPlan(ln, dev, target_blocks=tb)
            ms = timed(lambda: pl.partials(f_ln))
            out["sweep_target_blocks_nhwc"][tb] = (pl.num_blocks, round(n_ln * 4 / ms * 1e-6))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
