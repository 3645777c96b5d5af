"""Stream-K form of the fp32 dense kernel (vitta_gemm_nt_sk_f32) against the plain 64 x 64 launch on the Video Swin-B shapes whose
tile count quantises badly on 256 CUs: python tools/bench_gemm_sk.py [--views 2]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vitta_amd import _lib
from vitta_amd.ops import _p, _stream

p = argparse.ArgumentParser()
p.add_argument("--views", type=int, default=2)
p.add_argument("--reps", type=int, default=30)
opt = p.parse_args()
dev = torch.device("cuda:0")
L = _lib.lib()


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


m2 = opt.views * 8 * 14 * 14
shapes = [(m2, 512, 512, 0), (m2, 512, 2048, 0), (m2, 512, 1536, 0), (m2, 2048, 512, 1), (m2, 2048, 512, 2), (m2, 1536, 512, 0),
          (4 * m2, 256, 1024, 0), (4 * m2, 256, 256, 0), (m2 // 4, 1024, 4096, 0), (m2 // 4, 1024, 1024, 0)]
for (m, n, k, mode) in shapes:
    a = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * k ** -0.5
    b = torch.randn(n, device=dev)
    aux = torch.randn(m, n, device=dev) if mode == 2 else None
    pre = torch.empty(m, n, device=dev) if mode == 1 else None
    y0, y1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    gf = 2.0 * m * n * k / 1e9
    tiles = ((m + 63) // 64) * ((n + 63) // 64)
    us0 = timed(lambda: _lib.check(L.vitta_gemm_nt_f32(_p(a), _p(w), _p(b), _p(aux), _p(y0), _p(pre), m, n, k, mode, 3, _stream()), "plain"),
                opt.reps)
    rec = dict(M=m, N=n, K=k, mode=mode, tiles=tiles, plain_us=round(us0, 1), plain_tf=round(gf / us0 * 1e3, 1))
    for grid in (256, 512, 768, 1024):
        ws = torch.zeros(int(L.vitta_gemm_nt_sk_workspace_bytes(grid)) // 4, device=dev)
        fn = lambda: _lib.check(L.vitta_gemm_nt_sk_f32(_p(a), _p(w), _p(b), _p(aux), _p(y1), _p(pre), m, n, k, mode, grid, _p(ws),
                                                       ws.numel() * 4, _stream()), "sk")
        us = timed(fn, opt.reps)
        err = (y1 - y0).abs().max().item() / y0.abs().max().item()
        assert int(ws[:65536].view(torch.int32).abs().sum()) == 0, "counters not left zero"
        rec[f"sk{grid}_us"], rec[f"sk{grid}_err"] = round(us, 1), float(f"{err:.1e}")
    print(json.dumps(rec), flush=True)
