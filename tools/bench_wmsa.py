"""bf16 window attention at config 5's stage-0 shape (window (16, 7, 7) = 784 tokens, 4 heads, 256 windows = one 4 x 32-frame launch
per video and block): forward and backward per launch, the backward in both forms (VITTA_WMSA_BF16_BWD = one / two).

    python tools/bench_wmsa.py [--windows 256] [--heads 4] [--shift]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vitta_amd import ops, swin  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=256)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--shift", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--fp32", action="store_true", help="the fp32 kernels (wmsa.hip) on an fp32 qkv")
    ap.add_argument("--table-grad", action="store_true", help="a trainable relative-position table (its gradient binned in LDS by the one-pass backward)")
    opt = ap.parse_args()
    d = torch.device("cuda:0")
    ws = (16, 7, 7)
    n = ws[0] * ws[1] * ws[2]
    nh, c = opt.heads, opt.heads * 32
    g = torch.Generator().manual_seed(0)
    t_rows = (2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1)
    table = (torch.randn(t_rows, nh, generator=g) * 0.5).to(d)
    if opt.table_grad:
        table.requires_grad_(True)
        table.grad = torch.zeros_like(table)
    code, off = swin.relative_position_code(ws)
    code = code[:n].to(d)
    region = torch.randint(0, 4, (4, n), generator=g, dtype=torch.int32).to(d) if opt.shift else None
    io_t = torch.float32 if opt.fp32 else torch.bfloat16
    qkv = torch.randn(opt.windows, n, 3 * c, generator=g).to(d, io_t).requires_grad_(True)
    gout = torch.randn(opt.windows, n, c, generator=g).to(d, io_t)
    ops.WMSA_BF16 = not opt.fp32
    flops_f = 4.0 * n * n * 32 * opt.windows * nh

    def timed(fn):
        # wall clock between device synchronisations (round 6: the event pair around the loop reported 8.3 ms for a 1.9 ms iteration in
        # the table-gradient mode -- the timeline of the same run under rocprofv3 and tools/debug/dtable_wall_probe.py agree on 1.9)
        import time
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(opt.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6 / opt.reps

    out = [None]

    def fwd():
        out[0] = ops.WindowAttentionRel.apply(qkv, table, code, off, region, 32 ** -0.5, nh)

    tf = timed(fwd)
    print(f"forward            {tf:8.1f} us  {flops_f / tf / 1e6:6.1f} TF")
    for form in (("one",) if (opt.table_grad or opt.fp32) else ("two", "one")):
        os.environ["VITTA_WMSA_BF16_BWD"] = form

        def fb():
            fwd()
            out[0].backward(gout)

        t = timed(fb) - tf
        print(f"backward ({form})     {t:8.1f} us  {3.5 * flops_f / t / 1e6:6.1f} TF (14 N^2 d convention)")


if __name__ == "__main__":
    main()
