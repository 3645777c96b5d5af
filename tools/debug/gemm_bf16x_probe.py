"""Timing of gemm_bf16x.hip on the dense shapes of Video Swin-B at BASELINE config 5's size (4 views x 32 frames x 224^2: tokens
200704 / 50176 / 12544 / 3136 at stages 1-4) beside gemm.hip's bf16 kernel (fp32 activations rounded while staging) and
torch.matmul on bf16 tensors (hipBLASLt).  hipGraph-replay timing as tools/bench_conv.py."""
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
rows = []
tot = {"bf16x": 0.0, "bf16x_o16": 0.0, "bf16x_gelu16": 0.0, "gemm_bf16": 0.0, "torch": 0.0}
flops = 0.0
for tokens, c in ((200704, 128), (50176, 256), (12544, 512), (3136, 1024)):
    for n, k in ((3 * c, c), (c, c), (4 * c, c), (c, 4 * c)):
        a32 = torch.randn(tokens, k, device=dev)
        w32 = torch.randn(n, k, device=dev) * k ** -0.5
        a, w = a32.to(torch.bfloat16), w32.to(torch.bfloat16)
        y = torch.empty(tokens, n, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def f_x():
            _lib.check(L.vitta_gemm_nt_bf16x_f32(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), None, C.c_void_p(y.data_ptr()), tokens, n, k,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "bf16x")

        def f_g():
            ops.gemm_nt(a32, w, out=y)

        yt = torch.empty(tokens, n, device=dev, dtype=torch.bfloat16)

        def f_t():
            torch.matmul(a, w.t(), out=yt)
        pre = torch.empty(tokens, n, device=dev, dtype=torch.bfloat16)

        def f_o():
            ops.gemm_bf16x(a, w, None, out_bf16=True)

        def f_m():
            ops.gemm_bf16x(a, w, None, mode=1, pre=pre, out_bf16=True)
        us = {"bf16x": time_it(f_x, 10), "bf16x_o16": time_it(f_o, 10), "bf16x_gelu16": time_it(f_m, 10), "gemm_bf16": time_it(f_g, 10),
              "torch": time_it(f_t, 10)}
        fl = 2.0 * tokens * n * k
        flops += fl
        for kk in tot:
            tot[kk] += us[kk]
        rows.append(dict(tokens=tokens, n=n, k=k, **{f"{kk}_us": round(v, 1) for kk, v in us.items()}, **{f"{kk}_tf": round(fl / v / 1e6, 1) for kk, v in us.items()}))
        print(rows[-1], flush=True)
summary = {f"{kk}_ms": round(v / 1e3, 3) for kk, v in tot.items()}
summary.update({f"{kk}_tf": round(flops / v / 1e6, 1) for kk, v in tot.items()})
print(summary)
json.dump(dict(rows=rows, summary=summary), open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_bf16x_probe.json", "w"), indent=1)
