"""The TAM of a block, forward and backward, as the trunk issues it: the separate launches (branches + aggregation pass; aggregation
backward + branches + bn1 backward) against the merged launches (vitta_tam_fwd_agg_f32 / vitta_tam_bwd_all_f32), each as a dependent
chain of R repetitions inside a replayed hipGraph: microseconds per TAM at the trunk's four shapes (two views x 8 frames).

    python tools/bench_tam_merge.py [--out file.json] [--frames 8]
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
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--clips", type=int, default=2)
    opt = ap.parse_args()
    L = _lib.lib()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    rows = []
    for c, hw in ((64, 3136), (128, 784), (256, 196), (512, 49)):
        n, t = opt.clips, 8
        o, P = c // 4, n * t * hw
        r = lambda *s: torch.randn(*s, generator=g).to(d)
        x1, gout = r(c, P), r(c, P)
        bn1 = [torch.rand(c, generator=g).to(d) + 0.5, r(c) * 0.2, r(c) * 0.2, torch.rand(c, generator=g).to(d) + 0.5]
        pooled = torch.round(r(n, c, t) * 4096) / 4096
        pooled_tc = torch.round(pooled.permute(0, 2, 1).double() * 2.0 ** 32).to(torch.int64).contiguous()
        wg1, wg3, w0, w3 = r(2 * t, t) * 0.3, r(3, 2 * t) * 0.3, r(o, c, 3) * (3 * c) ** -0.5, r(c, o) * o ** -0.5
        bng = [torch.rand(2 * t, generator=g).to(d) + 0.5, r(2 * t) * 0.1, r(2 * t) * 0.1, torch.rand(2 * t, generator=g).to(d) + 0.5]
        bnl = [torch.rand(o, generator=g).to(d) + 0.5, r(o) * 0.1, r(o) * 0.1, torch.rand(o, generator=g).to(d) + 0.5]
        sync = torch.zeros(8192, dtype=torch.int32, device=d)
        kern, gate, hpre = torch.empty(n * c, 3, device=d), torch.empty(n, c, t, device=d), torch.empty(2, n, o, t, device=d)
        a1, ga, dx = torch.empty(c, P, device=d), torch.empty(c, P, device=d), torch.empty(c, P, device=d)
        ggate, gkern = torch.empty(n * c * t * 4, device=d), torch.empty(n * c, 3, device=d)
        gbuf = torch.empty(n * c * t + n * o * t, device=d)
        dbn = [torch.zeros(2 * t, device=d), torch.zeros(2 * t, device=d), torch.zeros(o, device=d), torch.zeros(o, device=d)]
        dg1, db1 = torch.zeros(c, device=d), torch.zeros(c, device=d)
        args = (_p(pooled_tc), _p(wg1), _ptr4(*bng), 1e-5, _p(wg3), _p(w0), _ptr4(*bnl), 1e-5, _p(w3), n, c, t)
        nul = _ptr4(None, None, None, None)
        bargs = args + (n, _p(kern), _p(gate), _p(hpre), _p(gkern), _p(ggate), _p(gbuf), _ptr4(*dbn), nul)

        def fwd_sep():
            _lib.check(L.vitta_tam_branch_fwd_fused_f32(*args, _p(kern), _p(gate), _p(hpre), _p(sync), 1, _stream()), "fwd")
            _lib.check(L.vitta_tam_agg_fwd_cm_f32(_p(x1), _ptr4(*bn1), 1e-5, _p(gate), _p(kern), c, n, t, hw, _p(a1), _stream()), "agg")

        def fwd_mrg():
            _lib.check(L.vitta_tam_fwd_agg_f32(*args, _p(kern), _p(gate), _p(hpre), _p(sync), 1, _p(x1), _ptr4(*bn1), 1e-5, hw, _p(a1), _stream()), "m")

        def bwd_sep():
            _lib.check(L.vitta_tam_agg_bwd_cm_ld_f32(_p(x1), 0, _ptr4(*bn1), 1e-5, _p(gate), _p(kern), _p(gout), c, n, t, hw, _p(ga), _p(ggate),
                                                     _p(gkern), _stream()), "agg bwd")
            _lib.check(L.vitta_tam_branch_bwd_fused_f32(*bargs, _p(sync), 1, _stream()), "bwd")
            _lib.check(L.vitta_bn_bwd_cm_ld_f32(_p(ga), None, _p(x1), None, 0, _p(gbuf), 1.0 / hw, _ptr4(*bn1), 1e-5, None, None, None, None, 1,
                                                _p(dx), None, _p(dg1), _p(db1), c, n, t, hw, _stream()), "bn bwd")

        def bwd_mrg():
            _lib.check(L.vitta_tam_bwd_all_f32(*bargs, _p(sync), 1, _p(x1), 0, _ptr4(*bn1), 1e-5, _p(gout), hw, _p(ga), None, None, None, None, 1,
                                               _p(dx), _p(dg1), _p(db1), _stream()), "bwd m")

        row = dict(C=c, HW=hw, N=n, T=t)
        fwd_sep()
        for name, fn in (("fwd_separate", fwd_sep), ("fwd_merged", fwd_mrg), ("bwd_separate", bwd_sep), ("bwd_merged", bwd_mrg)):
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
        json.dump(dict(rows=rows, reps=opt.reps), open(opt.out, "w"), indent=1)


if __name__ == "__main__":
    main()
