"""gemm_f32x.hip (exact fp32 MFMA, 128 x 128 tiles on the LDS-DMA ring) beside gemm.hip (64 x 64 tiles, register staging) on the
dense shapes of Video Swin-B at BASELINE config 3's size (2 views x 16 frames x 224^2: tokens 50176 / 12544 / 3136 / 784 at stages
1-4; --eval: the one-view evaluation pass, half the tokens): error against fp64 and hipGraph-replay timing, every epilogue mode."""
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
div = 2 if "--eval" in sys.argv else 1
rows = []
tot = {"x": 0.0, "x_gelu": 0.0, "x_dgelu": 0.0, "g": 0.0, "g_gelu": 0.0, "g_dgelu": 0.0}
flops = 0.0
for tokens, c in ((50176 // div, 128), (12544 // div, 256), (3136 // div, 512), (784 // div, 1024)):
    for n, k in ((3 * c, c), (c, c), (4 * c, c), (c, 4 * c)):
        a = torch.randn(tokens, k, device=dev)
        w = torch.randn(n, k, device=dev) * k ** -0.5
        bias = torch.randn(n, device=dev)
        aux = torch.randn(tokens, n, device=dev)
        y, yg, pre = (torch.empty(tokens, n, device=dev) for _ in range(3))

        def fx(mode):
            return lambda: _lib.check(L.vitta_gemm_nt_f32x(P(a), P(w), P(bias), P(aux) if mode == 2 else None, P(y), P(pre) if mode == 1 else None,
                                                          tokens, n, k, mode, ST()), "f32x")

        def fg(mode):
            return lambda: ops.gemm_nt(a, w, bias, mode=mode, aux=aux if mode == 2 else None, pre=pre if mode == 1 else None, out=yg)
        err = {}
        sl = slice(max(0, tokens - 1500), tokens)  # includes the ragged last tile
        h = a[sl].double() @ w.double().t()
        refs = {0: h + bias.double(), 1: torch.nn.functional.gelu(h + bias.double()),
                2: h * (0.5 * (1 + torch.erf(aux[sl].double() / 2 ** 0.5)) + aux[sl].double() * torch.exp(-0.5 * aux[sl].double() ** 2) / (2 * torch.pi) ** 0.5)}
        for mode in (0, 1, 2):
            fx(mode)(); fg(mode)()
            sc = refs[mode].abs().max().item()
            err[f"x_err{mode}"] = float(f"{(y[sl].double() - refs[mode]).abs().max().item() / sc:.2e}")
            err[f"g_err{mode}"] = float(f"{(yg[sl].double() - refs[mode]).abs().max().item() / sc:.2e}")
            if mode == 1:
                err["x_pre_err"] = float(f"{(pre[sl].double() - (h + bias.double())).abs().max().item() / sc:.2e}")
        us = {"x": time_it(fx(0), 10), "x_gelu": time_it(fx(1), 10), "x_dgelu": time_it(fx(2), 10), "g": time_it(fg(0), 10), "g_gelu": time_it(fg(1), 10),
              "g_dgelu": time_it(fg(2), 10)}
        fl = 2.0 * tokens * n * k
        flops += fl
        for kk in tot:
            tot[kk] += us[kk]
        rows.append(dict(tokens=tokens, n=n, k=k, tiles128=((tokens + 127) // 128) * (n // 128), **{f"{kk}_us": round(v, 1) for kk, v in us.items()},
                         x_tf=round(fl / us["x"] / 1e6, 1), g_tf=round(fl / us["g"] / 1e6, 1), **err))
        print(rows[-1], flush=True)
summary = {f"{kk}_ms": round(v / 1e3, 3) for kk, v in tot.items()}
summary.update({f"{kk}_tf": round(flops / v / 1e6, 1) for kk, v in tot.items()})
print(summary)
json.dump(dict(rows=rows, summary=summary), open("gpurun_out/gemm_f32x_probe%s.json" % ("_eval" if div == 2 else ""), "w"), indent=1)
