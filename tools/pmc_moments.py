"""Workload for the PMC passes: the batched moments launch at the in-step size (1 video, 178 MB) and at a
streaming size (16 videos' worth, 2.85 GB).  Run under
    rocprofv3 --pmc FETCH_SIZE  --kernel-trace -d out -o fetch -- python tools/pmc_moments.py
    rocprofv3 --pmc WRITE_SIZE  --kernel-trace -d out -o write -- python tools/pmc_moments.py
(separate passes, MI355X_MICROARCH.md section HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitta_amd import ops
from tools.bench_moments import C2  # noqa: E402  (shape list of the 29 hooked layers)

dev = torch.device("cuda:0")
for copies in (1, 16):
    shapes = [(o * copies, c, i, l) for o, c, i, l in C2]
    plan = ops.StatPlan(shapes, dev)
    feats = [torch.randn(o * c * i, device=dev) for o, c, i, _ in shapes]
    shift = torch.zeros(plan.total_channels, device=dev)
    # evict the freshly written operands of the small case from the Infinity Cache with a 1 GB sweep
    junk = torch.empty(256 * 1024 * 1024, device=dev)
    for _ in range(5):
        junk.fill_(1.0)
        torch.cuda.synchronize()
        plan.moments(feats, shift)
        torch.cuda.synchronize()
    print("copies", copies, "bytes", 4 * sum(f.numel() for f in feats), "workgroups", plan.num_blocks, flush=True)
