"""gemm_x3.hip (both operands pre-split into three bfloat16 terms) on the dense shapes of Video Swin-B at BASELINE config 3's size
(2 views x 16 frames x 224^2: tokens 50176 / 12544 / 3136 / 784 at stages 1-4): error against fp64 beside gemm.hip's exact-fp32
kernel, and hipGraph-replay timing of both (+ the split pass of the activation)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_conv import time_it  # noqa: E402
from vitta_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def split3(x):
    y = torch.empty(x.shape[0], 3 * x.shape[1], device=dev, dtype=torch.bfloat16)
    _lib.check(L.vitta_split3_f32(P(x), P(y), x.shape[0], x.shape[1], ST()), "split3")
    return y


def join3(y3, cols):
    return y3.view(y3.shape[0], cols // 16, 3, 16).float().sum(2).reshape(y3.shape[0], cols)


rows = []
tot = {"x3": 0.0, "x3_out3": 0.0, "f32": 0.0, "split": 0.0}
flops = 0.0
check = "--no-check" not in sys.argv
for tokens, c in ((50176, 128), (12544, 256), (3136, 512), (784, 1024)):
    for n, k in ((3 * c, c), (c, c), (4 * c, c), (c, 4 * c)):
        a = torch.randn(tokens, k, device=dev)
        w = torch.randn(n, k, device=dev) * k ** -0.5
        bias = torch.randn(n, device=dev)
        a3, w3 = split3(a), split3(w)
        y = torch.empty(tokens, n, device=dev)
        y3 = torch.empty(tokens, 3 * n, device=dev, dtype=torch.bfloat16)
        yf = torch.empty(tokens, n, device=dev)

        def f_x():
            _lib.check(L.vitta_gemm_nt_x3(P(a3), P(w3), P(bias), None, P(y), None, tokens, n, k, 0, 0, ST()), "x3")

        def f_o():
            _lib.check(L.vitta_gemm_nt_x3(P(a3), P(w3), P(bias), None, P(y3), None, tokens, n, k, 0, 1, ST()), "x3")

        def f_g():
            ops.gemm_nt(a, w, bias, out=yf)

        def f_s():
            _lib.check(L.vitta_split3_f32(P(a), P(a3), tokens, k, ST()), "split3")
        err = {}
        if check:
            f_x(); f_o(); f_g()
            sl = slice(0, min(tokens, 2048))
            ref = a[sl].double() @ w.double().t() + bias.double()
            sc = ref.abs().max().item()
            err = dict(split_err=(join3(a3[sl], k).double() - a[sl].double()).abs().max().item() / a.abs().max().item(),
                       x3_err=(y[sl].double() - ref).abs().max().item() / sc, x3_out3_err=(join3(y3[sl], n).double() - ref).abs().max().item() / sc,
                       f32_err=(yf[sl].double() - ref).abs().max().item() / sc)
            # the ragged last tile and the far end of the rows
            tl = slice(tokens - 200, tokens)
            ref2 = a[tl].double() @ w.double().t() + bias.double()
            err["x3_tail_err"] = (y[tl].double() - ref2).abs().max().item() / sc
        us = {"x3": time_it(f_x, 10), "x3_out3": time_it(f_o, 10), "f32": time_it(f_g, 10), "split": time_it(f_s, 10)}
        fl = 2.0 * tokens * n * k
        flops += fl
        for kk in tot:
            tot[kk] += us[kk]
        rows.append(dict(tokens=tokens, n=n, k=k, **{f"{kk}_us": round(v, 1) for kk, v in us.items()},
                         **{f"{kk}_tf": round(fl / v / 1e6, 1) for kk, v in us.items() if kk != "split"}, **{kk: float(f"{v:.2e}") for kk, v in err.items()}))
        print(rows[-1], flush=True)
summary = {f"{kk}_ms": round(v / 1e3, 3) for kk, v in tot.items()}
summary.update({f"{kk}_tf": round(flops / v / 1e6, 1) for kk, v in tot.items() if kk != "split"})
print(summary)
json.dump(dict(rows=rows, summary=summary), open(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/gemm_x3_probe.json", "w"), indent=1)
