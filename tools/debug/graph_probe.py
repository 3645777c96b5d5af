"""Probe which configuration breaks hipGraph capture (each case in its own process)."""
import json
import subprocess
import sys

CASE = r'''
import sys, json, torch, numpy as np, torch.nn as nn
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import helpers as H
from vitta_amd import data, tta
size, dropout, engine, first, consis = SIZE, DROPOUT, ENGINE, FIRST, CONSIS
dev = torch.device("cuda:0")
model = H.build_tanet(101, 8, 0)
if not dropout:
    model.base_model.fc = nn.Identity()
bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
import tempfile
tmp = tempfile.mkdtemp()
mp, vp = H.write_stat_files(tmp, [np.zeros(b.num_features, np.float32) for b in bn2d], [np.ones(b.num_features, np.float32) for b in bn2d])
args = H.tanet_args(tmp, clip_length=8, input_size=size, spatiotemp_mean_clean_file=mp, spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=1e-4, if_pred_consistency=consis)
if first:
    a0 = tta.ViTTAAdapter(tta.SingleDeviceParallel(H.build_tanet(101, 8, 0)).to(dev), args, use_engine=engine)
    a0.set_adapt_mode(); a0.adapt_step(a0.shape_tta_input(torch.randn(1, 48, size, size, device=dev))); del a0
adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(dev), args, use_engine=engine)
x = adapter.shape_tta_input(torch.randn(1, 48, size, size, device=dev))
ev = adapter.shape_eval_input(torch.randn(1, 24, size, size, device=dev))
for i in range(2):
    adapter.set_adapt_mode(); adapter.adapt_step(x); adapter.close_hooks(); adapter.evaluate(ev); adapter.add_hooks_back()
torch.cuda.synchronize()
adapter.capture_graphs(x, ev)
adapter.adapt_step(x); adapter.evaluate(ev)
torch.cuda.synchronize()
print("CASE_OK")
'''

cases = [dict(SIZE=64, DROPOUT=False, ENGINE=True, FIRST=False, CONSIS=True),
         dict(SIZE=64, DROPOUT=True, ENGINE=True, FIRST=False, CONSIS=True),
         dict(SIZE=112, DROPOUT=False, ENGINE=True, FIRST=False, CONSIS=True),
         dict(SIZE=64, DROPOUT=True, ENGINE=True, FIRST=True, CONSIS=True),
         dict(SIZE=64, DROPOUT=False, ENGINE=False, FIRST=False, CONSIS=True),
         dict(SIZE=64, DROPOUT=False, ENGINE=True, FIRST=False, CONSIS=False),
         dict(SIZE=224, DROPOUT=False, ENGINE=True, FIRST=False, CONSIS=True)]
for c in cases:
    code = CASE
    for k, v in c.items():
        code = code.replace(k, repr(v))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    ok = "CASE_OK" in r.stdout
    print(json.dumps(c), "->", "ok" if ok else f"FAIL rc={r.returncode}", flush=True)
    if not ok:
        print("   ", (r.stderr.strip().splitlines() or ["<no stderr>"])[-3:], flush=True)
