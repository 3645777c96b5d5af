"""A/B of the hand-written trunk against the module path: per-parameter gradient deviations of one adaptation step."""
import json, os, sys, tempfile
import numpy as np, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from vitta_amd import trunk, tta
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = H.golden("tta3.npz"); cfg = json.loads(str(g["config"])); T = cfg["T"]
res = {}
for fast in (True, False):
    tmp = tempfile.mkdtemp()
    model = H.build_tanet(101, T, 0)
    ch = g["src_channels"]; offs = np.concatenate([[0], np.cumsum(ch)])
    mp, vp = H.write_stat_files(tmp, [g["src_means"][offs[i]:offs[i+1]] for i in range(len(ch))], [g["src_vars"][offs[i]:offs[i+1]] for i in range(len(ch))])
    args = H.tanet_args(tmp, clip_length=T, input_size=size, batch_size=1, spatiotemp_mean_clean_file=mp, spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=cfg["lr_adam"])
    trunk.ENABLED = fast
    ad = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).cuda(), args)
    ad.model.module.base_model.fc = nn.Identity()
    x = H.seeded_randn((1, 2*T*3, size, size), 7).cuda()
    ad.set_adapt_mode()
    _, lr_, lc_ = ad.adapt_step(ad.shape_tta_input(x))
    res[fast] = (float(lr_), float(lc_), {k: v.grad.detach().clone() for k, v in ad.model.named_parameters() if v.requires_grad},
                 ad.engine.plan.s1.clone(), ad.engine.plan.s2.clone(), ad.engine.plan.coef_a.clone(), ad.engine.plan.coef_b.clone())
a, b = res[True], res[False]
print("loss", a[0], b[0], a[1], b[1])
for nm, i in (("s1", 3), ("s2", 4), ("coef_a", 5), ("coef_b", 6)):
    d = (a[i] - b[i]).abs(); print(nm, "max abs diff", d.max().item(), "max ref", b[i].abs().max().item(), "n differing sign", int(((a[i] * b[i]) < 0).sum()))
rows = []
for k, gb in b[2].items():
    ga = a[2][k]; e = (ga - gb).abs()
    rows.append((e.max().item() / (gb.abs().max().item() + 1e-12), k, int(e.argmax()), ga.flatten()[e.argmax()].item(), gb.flatten()[e.argmax()].item(), gb.abs().max().item()))
rows.sort(reverse=True)
for r in rows[:12]: print("%.2e %s idx %d fast %.4e ref %.4e max %.3e" % r)
print("median rel", np.median([r[0] for r in rows]))
