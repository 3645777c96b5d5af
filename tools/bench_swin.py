"""Video Swin-B TTA iteration timing on MI355X (BASELINE config 2 shape: 2 views x 16 frames x 224^2,
LN-affine Adam).  Not the driver's bench line (that is bench.py / TANet); used to size the W-MSA work."""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn as nn

from vitta_amd import data, scripts, tta
from vitta_amd import synthetic as S
from vitta_amd.bns_utils import choose_layers
from vitta_amd.norm_stats import ComputeNormStatsHook

p = argparse.ArgumentParser()
p.add_argument("--steps", type=int, default=10)
p.add_argument("--warmup", type=int, default=3)
p.add_argument("--size", type=int, default=224)
p.add_argument("--frames", type=int, default=16)
p.add_argument("--no-graph", action="store_true")
p.add_argument("--sgd", action="store_true")
p.add_argument("--views", type=int, default=2)
p.add_argument("--window-depth", type=int, default=8, help="temporal window (16 = the SSv2 recipe of BASELINE config 4)")
p.add_argument("--blas", default=None, choices=["hipblaslt", "hipblas", "default"], help="torch.backends.cuda.preferred_blas_library")
p.add_argument("--wmsa-bf16", action="store_true", help="bf16-operand window attention (vitta_wmsa_rel_*_bf16, BASELINE config 5)")
p.add_argument("--library-dense", action="store_true", help="qkv / proj / MLP on torch's library GEMMs + ATen GELU instead of csrc/gemm.hip")
p.add_argument("--dense-bf16", action="store_true", help="bf16-operand dense layers (vitta_gemm_nt_bf16w_f32)")
p.add_argument("--aten-sites", action="store_true", help="instead of timing: one eager step under torch.profiler, ATen kernels grouped by the vitta_amd line that launched them")
p.add_argument("--sequential", action="store_true", help="adapt(i); eval(i) on one stream (default: overlapped schedule)")
opt = p.parse_args()
if opt.blas:
    torch.backends.cuda.preferred_blas_library(opt.blas)
dev = torch.device("cuda:0")
if opt.library_dense:
    from vitta_amd import swin as _swin
    _swin.FUSED_DENSE = False
if opt.wmsa_bf16:
    from vitta_amd import ops as _ops
    _ops.WMSA_BF16 = True
if opt.dense_bf16:
    from vitta_amd import ops as _ops
    _ops.DENSE_BF16 = True
tmp = tempfile.mkdtemp()
model = S.build_swin(101, 0, window_size=(opt.window_depth, 7, 7)).to(dev)
lns = [m for _, m in choose_layers(model, [nn.LayerNorm])][1:]
hooks = [ComputeNormStatsHook(m, clip_len=opt.frames, stat_type="spatiotemp", before_norm=False, batch_size=1) for m in lns]
with torch.no_grad():
    model(S.seeded_randn((1, opt.views, 3, opt.frames, opt.size, opt.size), 1000, dev))
means = [h.batch_mean.cpu().numpy() for h in hooks]
vars_ = [h.batch_var.cpu().numpy() for h in hooks]
for h in hooks:
    h.close()
mp, vp = S.write_stat_files(tmp, means, vars_, tag="swin")
args = scripts.swin_ucf101_args([])
args.datatype, args.input_size, args.scale_size, args.workers, args.verbose = "synthetic", opt.size, opt.size, 0, False
args.clip_length, args.result_dir, args.num_classes = opt.frames, tmp, 101
args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
args.update_only_bn_affine = not opt.sgd
args.n_augmented_views, args.window_size = opt.views, (opt.window_depth, 7, 7)
args.synthetic_n_videos, args.synthetic_device = 8, dev
adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model), args)
tta_set = data.build_videoswin_dataset(args, "val", "tta")
eval_set = data.build_videoswin_dataset(args, "val", "eval")


def one(i):
    x, _ = tta_set[i % 8]
    if not opt.sequential:
        adapter.set_adapt_mode()
        return adapter.step(x.unsqueeze(0), eval_set[(i - 1) % 8][0].unsqueeze(0))
    ev, _ = eval_set[i % 8]
    if adapter._graph is not None:
        adapter.adapt_step(x.unsqueeze(0))
        return adapter.evaluate(ev.unsqueeze(0))
    adapter.set_adapt_mode()
    adapter.adapt_step(x.unsqueeze(0))
    adapter.close_hooks()
    out = adapter.evaluate(ev.unsqueeze(0))
    adapter.add_hooks_back()
    return out


for i in range(opt.warmup):
    one(i)
torch.cuda.synchronize()
if opt.aten_sites:
    import collections
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.defaultdict(lambda: [0, 0])
    WATCH = ("copy_", "clone", "_to_copy", "add", "add_", "zeros", "zero_", "fill_", "mul", "mul_", "cat", "zeros_like", "empty_like", "sum", "mean", "div")

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.__name__.split(".")[0]
            if name in WATCH and isinstance(out, torch.Tensor) and out.is_cuda:
                st = traceback.extract_stack()
                fr = next((f for f in reversed(st) if "vitta_amd" in f.filename), None)
                where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}" if fr else "(autograd engine / outside vitta_amd)"
                key = (func.__name__, str(out.dtype).replace("torch.", ""), where)
                sites[key][0] += 1
                sites[key][1] += out.numel() * out.element_size()
            return out
    with Log():
        one(0)
        torch.cuda.synchronize()
    for (name, dt, where), (n, by) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{n:5d} x {by / n / 1e6:8.2f} MB  {name:22s} {dt:9s} {where}")
    sys.exit(0)
if not opt.no_graph:
    adapter.capture_graphs(tta_set[0][0].unsqueeze(0), eval_set[0][0].unsqueeze(0), overlap_eval=not opt.sequential)
    one(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(opt.steps):
    one(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / opt.steps
print(json.dumps(dict(arch="swin_b", ms_per_video=1e3 * dt, videos_per_s=1 / dt, graph=not opt.no_graph,
                      frames=opt.frames, size=opt.size, views=opt.views, window=(opt.window_depth, 7, 7), schedule="sequential" if opt.sequential else "overlapped", optimizer="sgd_all" if opt.sgd else "adam_ln_affine", wmsa="bf16 operands" if opt.wmsa_bf16 else "fp32", dense="library" if opt.library_dense else ("gemm.hip bf16 operands" if opt.dense_bf16 else "gemm.hip fp32"),
                      max_mem_GB=torch.cuda.max_memory_allocated() / 1e9)))
