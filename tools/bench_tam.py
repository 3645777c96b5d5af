"""Latency of the TAM branch launches (tam_branch.hip) at the trunk's four shapes: one dependent chain of R launches inside a
replayed hipGraph, microseconds per launch -- what one such launch adds to the adaptation chain of a step (these launches are
2..128 workgroups of microseconds; their cost is latency, not throughput).

    python tools/bench_tam.py [--out file.json] [--affine]      (--affine: no weight gradients, the Adam-affine step's form)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vitta_amd import _lib  # noqa: E402
from vitta_amd.ops import _p, _ptr4, _stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--two-launch", action="store_true", help="time the two-launch entry points instead of the fused ones")
    opt = ap.parse_args()
    L = _lib.lib()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    rows = []
    for c in (64, 128, 256, 512):
        n, t = 2, 8
        o = c // 4
        r = lambda *s: torch.randn(*s, generator=g).to(d)
        pooled = torch.round(r(n, c, t) * 4096) / 4096
        pooled_tc = torch.round(pooled.permute(0, 2, 1).double() * 2.0 ** 32).to(torch.int64).contiguous()
        wg1, wg3, w0, w3 = r(2 * t, t) * 0.3, r(3, 2 * t) * 0.3, r(o, c, 3) * (3 * c) ** -0.5, r(c, o) * o ** -0.5
        bng = [torch.rand(2 * t, generator=g).to(d) + 0.5, r(2 * t) * 0.1, r(2 * t) * 0.1, torch.rand(2 * t, generator=g).to(d) + 0.5]
        bnl = [torch.rand(o, generator=g).to(d) + 0.5, r(o) * 0.1, r(o) * 0.1, torch.rand(o, generator=g).to(d) + 0.5]
        gkern, ggate = r(n * c, 3), r(n, c, t)
        sync = torch.zeros(256, dtype=torch.int32, device=d)
        kern, gate, hpre = torch.empty(n * c, 3, device=d), torch.empty(n, c, t, device=d), torch.empty(2, n, o, t, device=d)
        gbuf = torch.empty(n * c * t + n * o * t, device=d)
        dbn = [torch.zeros(2 * t, device=d), torch.zeros(2 * t, device=d), torch.zeros(o, device=d), torch.zeros(o, device=d)]
        args = (_p(pooled_tc), _p(wg1), _ptr4(*bng), 1e-5, _p(wg3), _p(w0), _ptr4(*bnl), 1e-5, _p(w3), n, c, t)
        nul = _ptr4(None, None, None, None)

        def fwd():
            if opt.two_launch:
                _lib.check(L.vitta_tam_branch_fwd_f32(*args, _p(kern), _p(gate), _p(hpre), 1, _stream()), "fwd")
            else:
                _lib.check(L.vitta_tam_branch_fwd_fused_f32(*args, _p(kern), _p(gate), _p(hpre), _p(sync), 1, _stream()), "fwd fused")

        def bwd():
            bargs = args + (_p(kern), _p(gate), _p(hpre), _p(gkern), _p(ggate), _p(gbuf), _ptr4(*dbn), nul)
            if opt.two_launch:
                _lib.check(L.vitta_tam_branch_bwd_f32(*bargs, 1, _stream()), "bwd")
            else:
                _lib.check(L.vitta_tam_branch_bwd_fused_f32(*bargs, _p(sync), 1, _stream()), "bwd fused")

        row = dict(C=c, N=n, T=t)
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            fn()
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(opt.reps):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            row[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / (10 * opt.reps), 2)
        rows.append(row)
        print(row, flush=True)
    if opt.out:
        json.dump(dict(rows=rows, two_launch=opt.two_launch, reps=opt.reps), open(opt.out, "w"), indent=1)


if __name__ == "__main__":
    main()
