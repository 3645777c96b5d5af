"""LayerNorm passes (vitta_ln_fwd_mixed / vitta_ln_bwd_mixed) at the Video Swin-B stage shapes: time per launch under hipGraph replay
and the HBM rate of the bytes each pass must move.  python tools/bench_ln.py [--views 4 --frames 32] [--bf16]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vitta_amd import _lib
from vitta_amd.ops import _p, _stream

p = argparse.ArgumentParser()
p.add_argument("--views", type=int, default=4)
p.add_argument("--frames", type=int, default=32)
p.add_argument("--bf16", action="store_true", help="bf16 branch / y / gy / gbranch (the bf16 recipe's data flow)")
p.add_argument("--reps", type=int, default=50)
opt = p.parse_args()
dev = torch.device("cuda:0")
L = _lib.lib()
f = dict(dtype=torch.float32, device=dev)
side = torch.bfloat16 if opt.bf16 else torch.float32
out = []
for c, hw in ((128, 56), (256, 28), (512, 14), (1024, 7), (512, 28), (1024, 14), (2048, 7)):
    rows = opt.views * (opt.frames // 2) * hw * hw
    rps = rows // opt.views
    x, branch = torch.randn(rows, c, **f), torch.randn(rows, c, **f).to(side)
    scale = torch.rand(opt.views, **f)
    w, b = torch.rand(c, **f) + 0.5, torch.randn(c, **f)
    xnew, y = torch.empty_like(x), torch.empty(rows, c, dtype=side, device=dev)
    mean, rstd = torch.empty(rows, **f), torch.empty(rows, **f)
    nb = int(L.vitta_ln_num_partials(rows))
    partial, shift = torch.empty(nb * 2 * c, **f), torch.zeros(c, **f)
    gy, gxnew = torch.randn(rows, c, **f).to(side), torch.randn(rows, c, **f)
    gx, gbranch = torch.empty_like(x), torch.empty(rows, c, dtype=side, device=dev)
    fl_f = (_lib.LN_BRANCH_BF16 | _lib.LN_Y_BF16) if opt.bf16 else 0
    fl_b = (_lib.LN_GY_BF16 | _lib.LN_GBRANCH_BF16) if opt.bf16 else 0
    sb = 2 if opt.bf16 else 4

    def fwd():
        _lib.check(L.vitta_ln_fwd_mixed(_p(x), _p(branch), _p(scale), rows, rps, c, _p(w), _p(b), 1e-5, _p(xnew), _p(y), _p(mean),
                                        _p(rstd), _p(shift), _p(partial), fl_f, _stream()), "fwd")

    def bwd():
        _lib.check(L.vitta_ln_bwd_mixed(_p(gy), _p(gxnew), _p(xnew), _p(mean), _p(rstd), _p(w), _p(b), _p(scale), None, None, None,
                                        None, rows, rps, c, _p(gx), _p(gbranch), _p(partial), fl_b, _stream()), "bwd")
    rec = dict(C=c, rows=rows)
    for name, fn, nbytes in (("fwd", fwd, rows * c * (4 + sb + 4 + sb)), ("bwd", bwd, rows * c * (sb + 4 + 4 + 4 + sb))):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(opt.reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / opt.reps
        rec[name + "_us"], rec[name + "_GBps"] = round(us, 2), round(nbytes / us / 1e3, 1)
    out.append(rec)
    print(json.dumps(rec), flush=True)
print(json.dumps(dict(views=opt.views, frames=opt.frames, bf16=opt.bf16, total_fwd_us=round(sum(r["fwd_us"] for r in out), 1),
                      total_bwd_us=round(sum(r["bwd_us"] for r in out), 1))))
