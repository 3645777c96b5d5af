"""Stability of the online loop: 300 synthetic videos through tta_standard (graph replay, overlapped evaluation);
memory must not grow and the per-video time must stay flat."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
import torch.nn as nn
import helpers as H
from vitta_amd import tta

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
model = H.build_tanet(101, 8, 0)
bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
mp, vp = H.write_stat_files(tmp, [np.zeros(b.num_features, np.float32) for b in bn2d], [np.ones(b.num_features, np.float32) for b in bn2d])
args = H.tanet_args(tmp, clip_length=8, input_size=112, spatiotemp_mean_clean_file=mp, spatiotemp_var_clean_file=vp,
                    update_only_bn_affine=True, lr=1e-6, synthetic_n_videos=300, verbose=True)
marks = []


class Log:
    def debug(self, msg):
        if msg.startswith("TTA Epoch1") and len(marks) % 50 == 0:
            print(len(marks), round(time.time() - t0, 2), "s", round(torch.cuda.memory_allocated() / 1e6), "MB alloc",
                  round(torch.cuda.memory_reserved() / 1e6), "MB reserved", flush=True)
        if msg.startswith("TTA Epoch1"):
            marks.append(time.time())


t0 = time.time()
res = tta.tta_standard(tta.SingleDeviceParallel(model).to(dev), torch.nn.CrossEntropyLoss().to(dev), args=args, logger=Log(), writer=None)
torch.cuda.synchronize()
d = np.diff(marks)
print("top1", res, "videos", len(marks), "ms/video first100 %.2f last100 %.2f" % (1e3 * d[10:110].mean(), 1e3 * d[-100:].mean()))
