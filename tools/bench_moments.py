"""Micro-benchmark of the batched moments launch (GPU box): workgroup count, NT loads, in-cache vs streaming.
Two timings per configuration: one dispatch-attached event pair per launch (what bench.py reports) and a back-to-back train of launches between one event pair (kernel time + launch gap)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vitta_amd import ops

C2 = [(16, 256, 784, 0), (16, 256, 196, 0), (16, 1024, 196, 0), (16, 1024, 196, 0)] + \
     [(16, 256, 196, 0), (16, 256, 196, 0), (16, 1024, 196, 0)] * 5 + \
     [(16, 512, 196, 0), (16, 512, 49, 0), (16, 2048, 49, 0), (16, 2048, 49, 0)] + [(16, 512, 49, 0), (16, 512, 49, 0), (16, 2048, 49, 0)] * 2
assert len(C2) == 29 and sum(o * c * i for o, c, i, _ in C2) == 44556288, (len(C2), sum(o * c * i for o, c, i, _ in C2))
dev = torch.device("cuda:0") if torch.cuda.is_available() else None


def run(copies, target, nt, reps=30, train=40):
    shapes = [(o * copies, c, i, l) for o, c, i, l in C2]
    plan = ops.StatPlan(shapes, dev, target_blocks=target, nt_loads=nt)
    feats = [torch.randn(o * c * i, device=dev) for o, c, i, _ in shapes]
    nbytes = 4 * sum(f.numel() for f in feats)
    shift = torch.zeros(plan.total_channels, device=dev)
    ts = []
    for r in range(reps + 5):
        ev = ops.KernelEventPair()
        plan.moments(feats, shift, events=ev)
        torch.cuda.synchronize()
        if r >= 5:
            ts.append(ev.elapsed_ms())
    ms = float(np.median(ts))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(train):
        plan.partials(feats)
    b.record()
    torch.cuda.synchronize()
    tr = a.elapsed_time(b) / train
    return dict(copies=copies, target=target, nt=nt, blocks=plan.num_blocks, MB=round(nbytes / 1e6, 1), us=round(1e3 * ms, 1),
                TBs=round(nbytes / ms / 1e9, 2), train_us=round(1e3 * tr, 1), train_TBs=round(nbytes / tr / 1e9, 2))


if __name__ == "__main__":
    copies_list = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 6, 16]
    targets = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 2048, 4096, 8192, 16384]
    for copies in copies_list:
        for target in targets:
            for nt in (False, True):
                print(json.dumps(run(copies, target, nt)), flush=True)
