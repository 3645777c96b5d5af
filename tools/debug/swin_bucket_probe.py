"""Which bucket signals of Video Swin-B fire in each step, and which unit parameters are not written straight into the arena
(one process, FORCE_EXCHANGES with a one-rank gloo group): python tools/debug/swin_bucket_probe.py"""
import json, os, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", VITTA_TEST_SWIN_SGD_ALL="1")
import numpy as np, torch
import helpers as H
from vitta_amd import data, scripts, tta
torch.distributed.init_process_group("gloo", rank=0, world_size=1)
tta.FORCE_EXCHANGES = True
dev = torch.device("cuda:0")
g = H.golden("tta3_swin.npz"); cfg = json.loads(str(g["config"])); T, size = cfg["T"], cfg["size"]
ch = g["src_channels"]; offs = np.concatenate([[0], np.cumsum(ch)]); tmp = tempfile.mkdtemp()
mp_, vp_ = H.write_stat_files(tmp, [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))], [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
model = H.build_swin(101, 0, drop_path_rate=0.0); model.cls_head.dropout = None
args = scripts.swin_ucf101_args([])
args.datatype, args.input_size, args.scale_size, args.workers, args.verbose = "synthetic", size, size, 0, False
args.result_dir, args.num_classes, args.batch_size = tmp, 101, 1
args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp_, vp_
args.update_only_bn_affine, args.lr = False, cfg["lr_sgd"]
ad = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(dev), args)
ds = data.SyntheticVideoDataset(6, 2, T, size, 101, "swin", seed0=cfg["seed0"])
real = ad._signal
fired = []
def spy(i):
    fired.append((i, ad._armed is not None))
    return real(i)
ad._signal = spy
names = {id(p): n for n, p in ad.model.named_parameters()}
for step in range(3):
    fired.clear()
    ad.set_adapt_mode()
    ad.adapt_step(ad.shape_tta_input(ds[step][0].unsqueeze(0).to(dev)))
    plan = ad.bucket_plan()
    nd = [names[id(p)] for b in plan["blocks"] for p in b.parameters() if id(p) in ad.arena.ranges and not getattr(p, "_vitta_direct_grad", False)]
    print("step", step, "buckets", [(s, hi - lo) for s, lo, hi in plan["buckets"]], "signals fired", fired, "from backward so far", ad.n_from_backward,
          "not direct:", len(nd), nd[:6])
