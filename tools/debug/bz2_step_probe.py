import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, numpy as np, tempfile, pathlib
import helpers as H
from test_host_cpu import run_product_tta
g = H.golden("tta3_bz2.npz")
recs = run_product_tta(g, "sgd", pathlib.Path(tempfile.mkdtemp()), torch.device("cuda:0"), None, batch_size=2)
rows = int(g["sample_rows"])
for i, rec in enumerate(recs):
    k = f"sgd_step{i}_"
    out = []
    for name, gr in rec["grads"].items():
        key = k + f"grad::{name}"
        if key not in g.files: continue
        ref = torch.from_numpy(g[key])
        out.append((name.split("base_model.")[-1], (gr[:rows]-ref).abs().max().item()/ref.abs().max().item(), 4*float(g[k+f"noise_grad::{name}"])/ref.abs().max().item()))
    print("step", i, "loss_reg", rec["loss_reg"], float(g[k+"loss_reg"]), " ".join(f"{n}:{e:.1e}(floor {f:.1e})" for n,e,f in out))
