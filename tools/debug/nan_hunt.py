"""Repeat the SGD-all online loop (64^2 synthetic, graph + overlap) until a NaN shows up; print the per-video rows."""
import os, sys, re, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch, torch.nn as nn
import helpers as H
from vitta_amd import scripts, tta

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
model = H.build_tanet(101, 8, 0)
bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
mp, vp = H.write_stat_files(tmp, [np.zeros(b.num_features, np.float32) for b in bn2d], [np.ones(b.num_features, np.float32) for b in bn2d])
variant = sys.argv[1] if len(sys.argv) > 1 else "default"
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    a = scripts.tanet_ucf101_args([])
    a.datatype, a.clip_length, a.input_size, a.workers = "synthetic", 8, 64, 0
    a.synthetic_n_videos, a.verbose, a.num_classes = 10, True, 101
    a.spatiotemp_mean_clean_file, a.spatiotemp_var_clean_file = mp, vp
    a.update_only_bn_affine = False
    if variant == "no_overlap":
        a.overlap_eval = False
    if variant == "no_graph":
        a.hip_graph = False
    lines = []

    class Log:
        def debug(self, msg):
            lines.append(msg)
    tta.tta_standard(tta.SingleDeviceParallel(model).to(dev), torch.nn.CrossEntropyLoss().to(dev), args=a, logger=Log(), writer=None)
    rows = [l for l in lines if l.startswith("TTA Epoch1")]
    bad = [i for i, l in enumerate(rows) if "nan" in l.lower()]
    print(variant, "trial", trial, "first nan at video", bad[0] if bad else None, flush=True)
    if bad:
        for l in rows[max(0, bad[0] - 2):bad[0] + 2]:
            print("   ", re.sub(r"\s+", " ", l)[:200])
