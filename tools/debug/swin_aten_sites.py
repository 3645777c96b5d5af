import sys, os, collections
sys.argv = [sys.argv[0], "--views", "2", "--frames", "16", "--window-depth", "8", "--no-graph", "--steps", "1", "--warmup", "3"]
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
src = open("/root/repo/tools/bench_swin.py").read()
cut = src.index("if opt.aten_sites:")
ns = {"__name__": "bench_swin_prefix", "__file__": "/root/repo/tools/bench_swin.py"}
exec(compile(src[:cut], "bench_swin_prefix", "exec"), ns)
one = ns["one"]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    one(0)
    torch.cuda.synchronize()
ev = prof.events()
# map: cpu op events that launched kernels named *elementwise_kernel_manual_unroll* / vectorized
agg = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for k in e.kernels:
            if "elementwise" in k.name or "reduce_kernel" in k.name or "copyBuffer" in k.name.lower():
                fr = next((s for s in (e.stack or []) if "vitta_amd" in s), "(no vitta frame)")
                agg[(e.name, str(e.input_shapes)[:80], fr[-90:], k.name[:40])] += 1
for (name, shp, fr, kn), n in agg.most_common(40):
    print(n, name, shp, "|", fr, "|", kn)
