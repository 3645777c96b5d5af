"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference) in the build container.

    python tools/refgen/gen_golden.py [section ...]

Every fixture stores only seeds/shapes (inputs are regenerated from seeds with tests/helpers.py)
and the reference's OUTPUTS.  Sections: l2ops layers tam tanet tta sampler opts dp
This script is the committed "generating script" the golden vectors came from; it is never run on
the GPU box (no /root/reference there).
"""
import io
import json
import logging
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refimport  # noqa: E402

os.chdir(HERE)  # reference modules append the working directory to sys.path: it must not be the repo root (see below)
ARGV = sys.argv[1:]  # refimport.install() resets sys.argv (the reference parses it at import time)
refimport.install()
# the repo root carries drop-in packages with the reference's names (corpus/, utils/, models/): bind those names to
# the REFERENCE before anything puts the repo root on sys.path
import corpus.basics as _ref_basics  # noqa: E402
import utils.opts as _ref_opts  # noqa: E402
assert _ref_basics.__file__.startswith(refimport.REFERENCE) and _ref_opts.__file__.startswith(refimport.REFERENCE)
sys.path.insert(0, os.path.join(refimport.REPO, "tests"))
import helpers as H  # noqa: E402

OUT = H.GOLDEN_DIR
os.makedirs(OUT, exist_ok=True)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def t2n(t):
    return t.detach().cpu().numpy().copy()  # copy: params keep changing in place after capture


# --------------------------------------------------------------------------------------------
def gen_l2ops():
    """A1-A5: moments, 3-step EMA + loss + gradient for every reg_type, prediction consistency."""
    from utils.norm_stats_utils import CombineNormStatsRegHook_onereg, ComputeNormStatsHook
    from utils.pred_consistency_utils import compute_pred_consis
    out = {}
    cases = {"bn2d_small": ("bn2d", (2 * 2 * 4, 8, 7, 7), 1, 4), "bn2d_odd": ("bn2d", (16, 7, 5, 3), 1, 8),
             "bn2d_full": ("bn2d", (16, 1024, 14, 14), 1, 8), "ln_small": ("ln", (4, 4, 7, 7, 16), 4, None),
             "ln_mid": ("ln", (2, 8, 7, 7, 96), 4, None)}
    meta = {}
    for name, (kind, shape, cdim, clip) in cases.items():
        x = H.channel_feature(shape, 11, cdim)
        mod = H.feature_module(kind, shape[cdim])
        hook = ComputeNormStatsHook(mod, clip_len=clip, stat_type="spatiotemp", before_norm=False, batch_size=shape[0])
        mod(x)
        out[f"mom_{name}_mean"], out[f"mom_{name}_var"] = t2n(hook.batch_mean), t2n(hook.batch_var)
        meta[name] = dict(kind=kind, shape=shape, cdim=cdim, clip=clip, seed=11)
        hook.close()

    # three successive hook invocations on one module
    ema_cases = {"bn2d": ("bn2d", (2 * 2 * 4, 8, 7, 7), 1, 4, 2), "ln": ("ln", (4, 4, 7, 7, 16), 4, None, 2)}
    for cname, (kind, shape, cdim, clip, views) in ema_cases.items():
        c = shape[cdim]
        g = torch.Generator().manual_seed(5)
        src_mean = torch.randn(c, generator=g) * 0.5
        src_var = torch.rand(c, generator=g) + 0.5
        out[f"ema_{cname}_src_mean"], out[f"ema_{cname}_src_var"] = t2n(src_mean), t2n(src_var)
        for reg in ("l1_loss", "mse_loss", "kld"):
            for mom in (0.1, 0.05):
                mod = H.feature_module(kind, c)
                hook = CombineNormStatsRegHook_onereg(
                    mod, clip_len=clip, spatiotemp_stats_clean_tuple=(src_mean.numpy(), src_var.numpy()), reg_type=reg,
                    moving_avg=True, momentum=mom, stat_type_list=["spatiotemp"], reduce_dim=True, before_norm=False,
                    if_sample_tta_aug_views=True, n_augmented_views=views)
                key = f"ema_{cname}_{reg}_{mom}"
                for step in range(3):
                    x = H.channel_feature(shape, 100 + step, cdim, offset_scale=1.0).requires_grad_(True)
                    mod(x)
                    r = hook.r_feature
                    (gx,) = torch.autograd.grad(r, x)
                    out[f"{key}_r{step}"] = t2n(r)
                    out[f"{key}_emamean{step}"] = t2n(hook.mean_avgmeter_spatiotemp.avg)
                    out[f"{key}_emavar{step}"] = t2n(hook.var_avgmeter_spatiotemp.avg)
                    out[f"{key}_gx{step}"] = t2n(gx)
                hook.close()
        meta[f"ema_{cname}"] = dict(kind=kind, shape=shape, cdim=cdim, clip=clip, views=views)

    for shape in ((1, 2, 101), (3, 4, 174)):
        z = (H.seeded_randn(shape, 7) * 3).requires_grad_(True)
        loss = compute_pred_consis(z)
        (gz,) = torch.autograd.grad(loss, z)
        out[f"consis_{shape[0]}_{shape[1]}_{shape[2]}_loss"] = t2n(loss)
        out[f"consis_{shape[0]}_{shape[1]}_{shape[2]}_grad"] = t2n(gz)
    out["meta"] = np.array(json.dumps(meta))
    save("l2ops.npz", **out)


# --------------------------------------------------------------------------------------------
def ref_tanet(num_class, T, seed, **kw):
    """Reference TSN carrying exactly the weights of this repo's seeded builder."""
    from models.tanet_models.tanet import TSN
    mine = H.build_tanet(num_class, T, seed, **kw)
    ref = TSN(num_class, T, "RGB", base_model="resnet50", consensus_type="avg", tam=True, print_spec=False,
              partial_bn=False)
    missing = ref.load_state_dict(mine.state_dict(), strict=True)
    ref.eval()
    return ref, mine


class Wrap(nn.Module):
    """names get the `module.` prefix nn.DataParallel would add"""

    def __init__(self, m):
        super().__init__()
        self.module = m

    def forward(self, *a, **k):
        return self.module(*a, **k)


def gen_layers():
    """A0: ordered candidate-layer names and hook selections for TANet (85 -> 47, stat idx 24..52)."""
    from utils.BNS_utils import choose_layers
    ref, _ = ref_tanet(11, 8, 0)
    model = Wrap(ref)
    chosen = choose_layers(model, [nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d])
    names = [n for n, _ in chosen]
    kinds = [type(m).__name__ for _, m in chosen]
    hooked = [i for i, n in enumerate(names) if any(b in n for b in ["layer3", "layer4"])]
    stat_idx, k = [], 0
    for i, kd in enumerate(kinds):
        if kd == "BatchNorm1d":
            stat_idx.append(-1)
        else:
            stat_idx.append(k)
            k += 1
    save("layers_tanet.npz", names=np.array(names), kinds=np.array(kinds), hooked=np.array(hooked),
         stat_idx=np.array(stat_idx), channels=np.array([m.num_features for _, m in chosen]))


# --------------------------------------------------------------------------------------------
def gen_tam():
    """A9: TAM forward/backward, eval-mode BN1d, weights stored in the fixture."""
    from models.tanet_models.temporal_module import TAM
    out = {}
    for name, (c, t, n, hw) in {"c64_t8": (64, 8, 2, 7), "c16_t16": (16, 16, 1, 6)}.items():
        torch.manual_seed(3)
        tam = TAM(c, t)
        H.perturb_affine(tam, 4)
        with torch.no_grad():
            for m in tam.modules():
                if isinstance(m, nn.BatchNorm1d):
                    m.running_mean.normal_(0, 0.1)
                    m.running_var.uniform_(0.5, 1.5)
        tam.eval()
        x = H.seeded_randn((n * t, c, hw, hw), 9).requires_grad_(True)
        gout = H.seeded_randn((n * t, c, hw, hw), 10)
        y = tam(x)
        params = [p for p in tam.parameters()]
        grads = torch.autograd.grad(y, [x] + params, gout)
        out[f"{name}_y"] = t2n(y)
        out[f"{name}_gx"] = t2n(grads[0])
        for (pn, _), gp in zip(tam.named_parameters(), grads[1:]):
            out[f"{name}_g_{pn}"] = t2n(gp)
        for k, v in tam.state_dict().items():
            out[f"{name}_sd_{k}"] = t2n(v)
        out[f"{name}_dims"] = np.array([c, t, n, hw])
    save("tam.npz", **out)


# --------------------------------------------------------------------------------------------
def gen_tanet():
    """A8: full TANet forward (eval mode) + per-hooked-layer batch moments from the reference's hooks."""
    from utils.norm_stats_utils import ComputeNormStatsHook
    K, T, size = 11, 8, 64
    ref, _ = ref_tanet(K, T, 0)
    x = H.seeded_randn((2, T, 3, size, size), 21)
    bn2d = [(n, m) for n, m in ref.named_modules() if isinstance(m, nn.BatchNorm2d)]
    hooks = [ComputeNormStatsHook(m, clip_len=T, stat_type="spatiotemp", before_norm=False, batch_size=2) for _, m in bn2d]
    with torch.no_grad():
        logits = ref(x)
    out = dict(logits=t2n(logits), names=np.array([n for n, _ in bn2d]))
    out["means"] = np.concatenate([t2n(h.batch_mean) for h in hooks])
    out["vars"] = np.concatenate([t2n(h.batch_var) for h in hooks])
    out["channels"] = np.array([m.num_features for _, m in bn2d])
    for h in hooks:
        h.close()
    save("tanet_fwd.npz", **out)


# --------------------------------------------------------------------------------------------
class _Capture:
    """Records what the reference's tta_standard does internally without modifying it."""

    def __init__(self):
        self.model = None
        self.steps = []
        self.eval_logits = []
        self.drop_masks = []
        self.consis = []


class _Perturbed(torch.utils.data.Dataset):
    """x * (1 + eps * noise): an fp32-round-off sized perturbation of the input clips."""

    def __init__(self, base, eps, seed):
        self.base, self.eps, self.seed = base, eps, seed

    def __len__(self):
        return len(self.base)

    def __getitem__(self, i):
        x, y = self.base[i]
        return x * (1 + self.eps * H.seeded_randn(tuple(x.shape), self.seed + i)), y


def run_reference_tta(args, model_origin, n_videos, batch_size, capture, perturb=0.0, perturb_seed=90000,
                      entry="tta_standard"):
    import corpus.basics as B
    import copy as _copy

    real_deepcopy = _copy.deepcopy

    def instrument(c):
        capture.model = c
        eps_p = getattr(capture, "perturb_params", 0.0)
        if eps_p:  # every parameter x (1 + eps N(0, 1)): what another fp32 implementation of every LAYER amounts to
            with torch.no_grad():
                for j, p_ in enumerate(c.parameters()):
                    p_.mul_(1 + eps_p * H.seeded_randn(tuple(p_.shape), perturb_seed + 7919 * (j + 1)))
        for m in c.modules():
            if isinstance(m, nn.Dropout):
                m.register_forward_hook(lambda mod, i, o: capture.drop_masks.append(t2n(o != 0)) if mod.training else None)
        c.register_forward_hook(lambda mod, i, o: capture.eval_logits.append(t2n(o)) if not mod.training else None)

    def spy_deepcopy(obj, *a, **k):
        c = real_deepcopy(obj, *a, **k)
        if isinstance(obj, nn.Module) and (capture.model is None or getattr(capture, "episodic", False)) \
                and isinstance(obj, Wrap):
            instrument(c)
        return c

    B.cp.deepcopy = spy_deepcopy
    real_consis = B.compute_pred_consis

    def spy_consis(p):
        v = real_consis(p)
        capture.consis.append(t2n(v))
        return v

    B.compute_pred_consis = spy_consis

    def spy_step(opt_cls):
        real = opt_cls.step

        def step(self, *a, **k):
            hooks = []
            for m in capture.model.modules():
                for h in m._forward_hooks.values():
                    owner = getattr(h, "__self__", None)
                    if owner is not None and hasattr(owner, "r_feature") and owner not in hooks:
                        hooks.append(owner)
            rec = dict(loss_reg=float(sum(float(h.r_feature) for h in hooks)),
                       r_features=np.array([float(h.r_feature) for h in hooks], dtype=np.float64))
            named = dict(capture.model.named_parameters())
            rec["grad_sq"] = float(sum(float((p.grad.double() ** 2).sum()) for p in named.values() if p.grad is not None))
            for key in SAMPLED_PARAMS:
                rec[f"grad::{key}"] = t2n(named[key].grad[:SAMPLE_ROWS]) if named[key].grad is not None else None
            r = real(self, *a, **k)
            rec["param_sum"] = float(sum(float(p.double().sum()) for p in named.values()))
            for key in SAMPLED_PARAMS:
                rec[f"param::{key}"] = t2n(named[key][:SAMPLE_ROWS])
            emas = [h.mean_avgmeter_spatiotemp.avg for h in hooks if hasattr(h.mean_avgmeter_spatiotemp.avg, "numel")
                    and h.mean_avgmeter_spatiotemp.avg.numel() > 1]
            rec["ema_mean_sum"] = float(sum(float(e.double().sum()) for e in emas))
            capture.steps.append(rec)
            return r

        opt_cls.step = step
        return real

    real_sgd, real_adam = spy_step(torch.optim.SGD), spy_step(torch.optim.Adam)

    from vitta_amd.data import SyntheticVideoDataset

    def fake_dataset(args, split="val", dataset_type=None):
        views = args.n_augmented_views if dataset_type == "tta" else 1
        ds = SyntheticVideoDataset(n_videos, views, args.clip_length, args.input_size, args.num_classes, "tanet", seed0=500)
        return _Perturbed(ds, perturb, perturb_seed) if perturb else ds

    B.get_dataset_tanet = fake_dataset
    logger = logging.getLogger("refgen")
    logger.addHandler(logging.NullHandler())
    try:
        if entry == "tta_standard":
            res = B.tta_standard(Wrap(model_origin), nn.CrossEntropyLoss(), args=args, logger=logger, writer=None)
        else:  # the epoch-style function adapts the model it is given: hand it a private, instrumented copy
            own = real_deepcopy(Wrap(model_origin))
            instrument(own)
            res, _ = B.test_time_adapt(own, nn.CrossEntropyLoss(), args=args, logger=logger, writer=None)
    finally:
        B.cp.deepcopy = real_deepcopy
        B.compute_pred_consis = real_consis
        torch.optim.SGD.step, torch.optim.Adam.step = real_sgd, real_adam
    return res


SAMPLE_ROWS = 4  # fixtures keep the first rows of each sampled tensor
SAMPLED_PARAMS = ["module.base_model.layer3.0.net.bn1.weight", "module.base_model.layer4.2.net.bn3.bias",
                  "module.base_model.layer4.1.tam.G.0.weight", "module.base_model.layer3.5.net.conv2.weight",
                  "module.new_fc.weight", "module.base_model.layer1.0.net.bn1.weight"]


def source_stats_for(ref, T, size):
    """Source statistics = the reference's own ComputeNormStatsHook on a seeded calibration batch,
    perturbed so the alignment loss is non-degenerate."""
    from utils.norm_stats_utils import ComputeNormStatsHook
    bn2d = [m for m in ref.modules() if isinstance(m, nn.BatchNorm2d)]
    hooks = [ComputeNormStatsHook(m, clip_len=T, stat_type="spatiotemp", before_norm=False, batch_size=2) for m in bn2d]
    with torch.no_grad():
        ref(H.seeded_randn((2, T, 3, size, size), 1000))
    g = torch.Generator().manual_seed(77)
    means = [t2n(h.batch_mean + 0.05 * torch.randn(h.batch_mean.shape, generator=g)) for h in hooks]
    vars_ = [t2n(h.batch_var * (1 + 0.2 * torch.rand(h.batch_var.shape, generator=g))) for h in hooks]
    for h in hooks:
        h.close()
    return means, vars_


NOISE_TRIALS = int(os.environ.get("VITTA_REFGEN_NOISE_TRIALS", "8"))


def gen_tta(batch_size=1, tag="tta3", size=64, n_steps=3):
    """A11/A7: n_steps online steps through the reference's own tta_standard, SGD-all and Adam-affine (tta3: three steps at 64^2;
    tta1_224: ONE step at the benchmarked size 2 x 8 x 224^2 -- BASELINE configs 2 / 4 -- so that the full-size GPU test compares
    with the reference itself, not with the product's host logic on the CPU)."""
    from utils.opts import get_opts
    K_dataset, T, n_videos = "ucf101", 8, n_steps * batch_size
    out = {}
    for mode in ("sgd", "adam"):
        ref, _ = ref_tanet(101, T, 0)
        means, vars_ = source_stats_for(ref, T, size)
        with tempfile.TemporaryDirectory() as tmp:
            mp, vp = H.write_stat_files(tmp, means, vars_)
            args = get_opts()
            args.arch, args.dataset, args.clip_length, args.workers = "tanet", K_dataset, T, 0
            args.input_size, args.verbose, args.batch_size = size, False, batch_size
            args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
            args.num_classes, args.gpus, args.result_dir = 101, [0], tmp
            args.update_only_bn_affine = mode == "adam"
            args.lr = 5e-5 if mode == "sgd" else 1e-3
            cap = _Capture()
            torch.manual_seed(1234)
            res = run_reference_tta(args, ref, n_videos, batch_size, cap)
            # noise floor: the reference against ITSELF with inputs perturbed at fp32 round-off level and
            # the same dropout masks (same seed, same draw order); worst of NOISE_TRIALS perturbation draws (three until
            # round 3: one tensor's floor then came out 10x below its neighbours' -- a single small sample of a chaotic
            # trajectory -- and the GPU tests papered over it with a median-of-others floor; eight draws per step instead).
            # Odd draws also perturb every PARAMETER by 1e-7 relative: an input-only perturbation models round-off in the first
            # layer, another implementation of the network rounds differently in EVERY layer.
            caps = []
            for trial in range(NOISE_TRIALS):
                c2 = _Capture()
                c2.perturb_params = 1e-7 if trial % 2 else 0.0
                torch.manual_seed(1234)
                run_reference_tta(args, ref, n_videos, batch_size, c2, perturb=1e-7, perturb_seed=90000 + 1000 * trial)
                caps.append(c2)
        assert len(cap.steps) == n_steps, len(cap.steps)
        for i in range(n_steps):
            a = cap.steps[i]
            noise = {}

            def bump(key, val):
                noise[key] = max(noise.get(key, 0.0), float(val))

            for cap2 in caps:
                # masks are recovered as (output != 0): entries whose INPUT is exactly 0 may differ, nothing else
                assert (cap.drop_masks[i] != cap2.drop_masks[i]).mean() < 1e-3
                b = cap2.steps[i]
                bump("loss_reg", abs(a["loss_reg"] - b["loss_reg"]))
                bump("loss_consis", np.abs(cap.consis[i] - cap2.consis[i]))
                bump("eval_logits", np.abs(cap.eval_logits[i] - cap2.eval_logits[i]).max())
                for key in SAMPLED_PARAMS:
                    if a.get(f"grad::{key}") is not None:
                        bump(f"grad::{key}", np.abs(a[f"grad::{key}"] - b[f"grad::{key}"]).max())
                    bump(f"param::{key}", np.abs(a[f"param::{key}"] - b[f"param::{key}"]).max())
            for key, val in noise.items():
                out[f"{mode}_step{i}_noise_{key}"] = np.array(val)
        out[f"{mode}_top1"] = np.array(res)
        for i, rec in enumerate(cap.steps):
            for k, v in rec.items():
                if v is not None:
                    out[f"{mode}_step{i}_{k}"] = np.asarray(v)
            out[f"{mode}_step{i}_loss_consis"] = cap.consis[i]
            out[f"{mode}_step{i}_eval_logits"] = cap.eval_logits[i]
            bits, shape = H.pack_mask(cap.drop_masks[i])
            out[f"{mode}_step{i}_dropmask"] = bits
            out[f"{mode}_step{i}_dropmask_shape"] = shape
        if mode == "sgd":
            out["src_means"] = np.concatenate(means)
            out["src_vars"] = np.concatenate(vars_)
            out["src_channels"] = np.array([len(m) for m in means])
    out["sampled_params"] = np.array(SAMPLED_PARAMS)
    out["sample_rows"] = np.array(SAMPLE_ROWS)
    out["config"] = np.array(json.dumps(dict(T=T, size=size, n_videos=n_videos, batch_size=batch_size, seed0=500,
                                             lr_sgd=5e-5, lr_adam=1e-3, n_steps=n_steps)))
    save(f"{tag}.npz", **out)


def gen_tta224():
    gen_tta(batch_size=1, tag="tta1_224", size=224, n_steps=1)


def gen_episodic():
    """N4: if_tta_standard='tta_standard' (episodic): model re-initialised per video, momentum_mvg = 1, two
    gradient steps per video; two videos through the reference's own tta_standard."""
    from utils.opts import get_opts
    T, size, n_videos = 8, 64, 2
    ref, _ = ref_tanet(101, T, 0)
    means, vars_ = source_stats_for(ref, T, size)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        mp, vp = H.write_stat_files(tmp, means, vars_)
        args = get_opts()
        args.arch, args.dataset, args.clip_length, args.workers = "tanet", "ucf101", T, 0
        args.input_size, args.verbose, args.batch_size = size, False, 1
        args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
        args.num_classes, args.gpus, args.result_dir, args.lr = 101, [0], tmp, 5e-5
        args.if_tta_standard, args.momentum_mvg, args.n_gradient_steps = "tta_standard", 1.0, 2
        cap = _Capture()
        cap.episodic = True
        torch.manual_seed(99)
        res = run_reference_tta(args, ref, n_videos, 1, cap)
        floor = 0.0
        for trial in range(3):  # the reference against itself, inputs perturbed by 1e-7 relative
            c2 = _Capture()
            c2.episodic = True
            torch.manual_seed(99)
            run_reference_tta(args, ref, n_videos, 1, c2, perturb=1e-7, perturb_seed=90000 + 1000 * trial)
            floor = max(floor, max(float(np.abs(a - b).max()) for a, b in zip(cap.eval_logits, c2.eval_logits)))
        out["noise_eval_logits"] = np.array(floor)
    assert len(cap.steps) == 4 and len(cap.eval_logits) == 2, (len(cap.steps), len(cap.eval_logits))
    for i, rec in enumerate(cap.steps):
        out[f"step{i}_loss_reg"] = np.array(rec["loss_reg"])
        out[f"step{i}_loss_consis"] = cap.consis[i]
        bits, shape = H.pack_mask(cap.drop_masks[i])
        out[f"step{i}_dropmask"], out[f"step{i}_dropmask_shape"] = bits, shape
    for v in range(2):
        out[f"video{v}_eval_logits"] = cap.eval_logits[v]
    out["src_means"], out["src_vars"] = np.concatenate(means), np.concatenate(vars_)
    out["src_channels"] = np.array([len(m) for m in means])
    out["top1"] = np.array(res)
    save("episodic.npz", **out)


def gen_epoch():
    """N4, second half: the epoch-style `test_time_adapt` (if_tta_standard falsy): four videos adapted two per
    step (Adam on the BN affine parameters), hooks closed, then `validate_brief` over the list in one batch."""
    from utils.opts import get_opts
    T, size, n_videos, bz = 8, 64, 4, 2
    ref, _ = ref_tanet(101, T, 0)
    means, vars_ = source_stats_for(ref, T, size)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        mp, vp = H.write_stat_files(tmp, means, vars_)
        args = get_opts()
        args.arch, args.dataset, args.clip_length, args.workers = "tanet", "ucf101", T, 0
        args.input_size, args.verbose, args.batch_size, args.batch_size_eval = size, False, bz, n_videos
        args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
        args.num_classes, args.gpus, args.result_dir, args.lr = 101, [0], tmp, 1e-3
        args.if_tta_standard, args.update_only_bn_affine, args.n_epoch_adapat = False, True, 1
        cap = _Capture()
        torch.manual_seed(77)
        res = run_reference_tta(args, ref, n_videos, bz, cap, entry="test_time_adapt")
        floor = 0.0
        for trial in range(3):
            c2 = _Capture()
            torch.manual_seed(77)
            run_reference_tta(args, ref, n_videos, bz, c2, perturb=1e-7, perturb_seed=90000 + 1000 * trial,
                              entry="test_time_adapt")
            floor = max(floor, float(np.abs(cap.eval_logits[0] - c2.eval_logits[0]).max()))
        out["noise_eval_logits"] = np.array(floor)
    assert len(cap.steps) == 2 and len(cap.eval_logits) == 1, (len(cap.steps), len(cap.eval_logits))
    for i, rec in enumerate(cap.steps):
        out[f"step{i}_loss_reg"] = np.array(rec["loss_reg"])
        out[f"step{i}_loss_consis"] = cap.consis[i]
        bits, shape = H.pack_mask(cap.drop_masks[i])
        out[f"step{i}_dropmask"], out[f"step{i}_dropmask_shape"] = bits, shape
        for key in SAMPLED_PARAMS:
            out[f"step{i}_param::{key}"] = rec[f"param::{key}"]
    out["eval_logits"] = cap.eval_logits[0]
    out["src_means"], out["src_vars"] = np.concatenate(means), np.concatenate(vars_)
    out["src_channels"] = np.array([len(m) for m in means])
    out["sampled_params"] = np.array(SAMPLED_PARAMS)
    out["top1"] = np.array(res)
    save("epoch.npz", **out)


def synthetic_frames(n, w, h, seed):
    """Seeded smooth-ish RGB frames as PIL images."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        base = rng.randint(0, 256, size=(h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
        img = Image.fromarray(base).resize((w, h), Image.BICUBIC)
        out.append(img)
    return out


def gen_data():
    """N1: the reference's own TANet transform classes on seeded frames + the Swin index samplers."""
    import random as pyrandom
    from models.tanet_models.transforms import (SubgroupWise_MultiScaleCrop_TANet, Stack_TANet, ToTorchFormatTensor_TANet,
                                                GroupNormalize_TANet)
    from models.videoswintransformer_models.transforms_backup import SampleFrames
    out = {}
    for case, (w, h, views, T, size) in {"a": (340, 256, 2, 8, 224), "b": (320, 240, 4, 4, 112)}.items():
        frames = synthetic_frames(views * T, w, h, 11)
        pyrandom.seed(5)
        crop = SubgroupWise_MultiScaleCrop_TANet(input_size=size, n_temp_clips=views, clip_len=T)
        imgs, _ = crop((frames, 0))
        arr, _ = Stack_TANet(roll=False)((imgs, 0))
        ten, _ = ToTorchFormatTensor_TANet(div=True)((arr, 0))
        ten, _ = GroupNormalize_TANet([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])((ten, 0))
        out[f"tanet_{case}_shape"] = np.array(ten.shape)
        out[f"tanet_{case}_sub"] = t2n(ten[:, ::16, ::16])
        out[f"tanet_{case}_chsum"] = t2n(ten.double().sum((1, 2)))
        out[f"tanet_{case}_cfg"] = np.array([w, h, views, T, size])
    for T in (8, 16, 32):
        for n in (300, 100, 37, 16, 9):
            sf = SampleFrames(clip_len=T, frame_interval=2, num_clips=1, test_mode=True, frame_uniform=True)
            out[f"swin_uniform_T{T}_n{n}"] = np.asarray(sf.get_seq_frames(n))
            for clips in (1, 4):
                sf = SampleFrames(clip_len=T, frame_interval=2, num_clips=clips, test_mode=True, frame_uniform=False)
                offs = sf._sample_clips(n)
                idx = np.mod(offs[:, None] + np.arange(T)[None, :] * 2, n).reshape(-1)
                out[f"swin_dense_T{T}_n{n}_c{clips}"] = idx
    save("data_pipeline.npz", **out)


def gen_dp():
    """8e: the reference with batch_size=2 -- what two data-parallel ranks must reproduce."""
    gen_tta(batch_size=2, tag="tta3_bz2")


# --------------------------------------------------------------------------------------------
def gen_sampler():
    """Frame-index samplers run as unbound methods of the reference dataset class."""
    from models.tanet_models.video_dataset import Video_TANetDataSet

    class Rec:
        def __init__(self, n):
            self.num_frames = n

    class Self:
        pass

    out = {}
    for T in (8, 16):
        for n in (100, 37, 300, 16, 9, 5, 64, 17):
            for V in (2, 4):
                s = Self()
                s.num_segments, s.new_length, s.n_tta_aug_views, s.test_sample = T, 1, V, "uniform-1"
                for style in ("uniform", "dense", "uniform_equidist", "dense_equidist"):
                    idx = Video_TANetDataSet._sample_tta_augmented_views(s, Rec(n), style)
                    out[f"tta_{style}_T{T}_n{n}_V{V}"] = np.asarray(idx)
            for ts in ("uniform-1", "uniform-2", "dense-1", "dense-3"):
                s = Self()
                s.num_segments, s.new_length, s.test_sample = T, 1, ts
                out[f"test_{ts}_T{T}_n{n}"] = np.asarray(Video_TANetDataSet._get_test_indices(s, Rec(n)))
    out["numpy_version"] = np.array(np.__version__)
    save("sampler.npz", **out)


def gen_opts():
    """Flag names and defaults of the reference parser."""
    from utils.opts import get_opts
    a = get_opts()
    d = {k: repr(getattr(a, k)) for k in sorted(vars(a))}
    with open(os.path.join(OUT, "opts_defaults.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print("wrote opts_defaults.json", len(d))


# --------------------------------------------------------------------------------------------
def ref_swin(num_class, seed, window_size=(8, 7, 7), **kw):
    from models.videoswintransformer_models.recognizer3d import Recognizer3D
    mine = H.build_swin(num_class, seed, window_size=tuple(window_size), **kw)
    ref = Recognizer3D(num_classes=num_class, patch_size=(2, 4, 4), window_size=tuple(window_size), drop_path_rate=0.2)
    ref.load_state_dict(mine.state_dict(), strict=True)
    ref.eval()
    return ref, mine


SWIN_BLOCKS = ["module.backbone.layers.2", "module.backbone.layers.3", "module.backbone.norm"]
SWIN_SAMPLED = ["module.backbone.layers.2.blocks.0.norm1.weight", "module.backbone.layers.3.blocks.1.norm2.bias",
                "module.backbone.norm.weight", "module.backbone.layers.2.blocks.5.attn.qkv.weight",
                "module.backbone.layers.2.blocks.3.attn.relative_position_bias_table",
                "module.backbone.layers.0.blocks.1.mlp.fc1.weight", "module.cls_head.fc_cls.weight",
                "module.backbone.layers.2.downsample.norm.weight"]


def gen_swin():
    """A0/A10 for Video Swin-B: layer selection (53 -> 52 -> 42), full forward + per-layer moments."""
    from utils.BNS_utils import choose_layers
    from utils.norm_stats_utils import ComputeNormStatsHook
    ref, _ = ref_swin(11, 0)
    chosen = choose_layers(Wrap(ref), [nn.LayerNorm])
    names = [n for n, _ in chosen]
    kept = names[1:]
    hooked = [i for i, n in enumerate(kept) if any(b in n for b in SWIN_BLOCKS)]
    save("layers_swin.npz", names=np.array(names), hooked=np.array(hooked),
         channels=np.array([m.normalized_shape[0] for _, m in chosen]))
    x = H.seeded_randn((1, 2, 3, 16, 112, 112), 31)
    lns = [m for _, m in choose_layers(ref, [nn.LayerNorm])][1:]
    hooks = [ComputeNormStatsHook(m, clip_len=16, stat_type="spatiotemp", before_norm=False, batch_size=1) for m in lns]
    with torch.no_grad():
        vid, view = ref(x)
    save("swin_fwd.npz", vid=t2n(vid), view=t2n(view), means=np.concatenate([t2n(h.batch_mean) for h in hooks]),
         vars=np.concatenate([t2n(h.batch_var) for h in hooks]), channels=np.array([m.normalized_shape[0] for m in lns]))
    for h in hooks:
        h.close()


def run_reference_tta_swin(args, model_origin, n_videos, capture, perturb=0.0, perturb_seed=90000):
    import corpus.basics as B
    import copy as _copy
    from timm.models.layers import DropPath

    real_deepcopy = _copy.deepcopy
    capture.droppath = []

    def spy_deepcopy(obj, *a, **k):
        c = real_deepcopy(obj, *a, **k)
        if isinstance(obj, nn.Module) and capture.model is None:
            capture.model = c
            eps_p = getattr(capture, "perturb_params", 0.0)
            if eps_p:  # (as run_reference_tta: another fp32 implementation rounds differently in EVERY layer)
                with torch.no_grad():
                    for j, p_ in enumerate(c.parameters()):
                        p_.mul_(1 + eps_p * H.seeded_randn(tuple(p_.shape), perturb_seed + 7919 * (j + 1)))
            for m in c.modules():
                if isinstance(m, nn.Dropout) and m.p > 0:
                    m.register_forward_hook(lambda mod, i, o: capture.drop_masks.append(t2n(o != 0)) if mod.training else None)
                if isinstance(m, DropPath) and m.drop_prob > 0:
                    m.register_forward_hook(
                        lambda mod, i, o: capture.droppath.append(t2n(o.flatten(1).abs().sum(1) > 0)) if mod.training else None)
            c.register_forward_hook(lambda mod, i, o: capture.eval_logits.append(t2n(o[0])) if not mod.training else None)
        return c

    B.cp.deepcopy = spy_deepcopy
    real_consis = B.compute_pred_consis

    def spy_consis(p):
        v = real_consis(p)
        capture.consis.append(t2n(v))
        return v

    B.compute_pred_consis = spy_consis
    saved_steps = {}
    for cls in (torch.optim.SGD, torch.optim.Adam):
        real = cls.step

        def step(self, *a, _real=real, **k):
            hooks = []
            for m in capture.model.modules():
                for h in m._forward_hooks.values():
                    owner = getattr(h, "__self__", None)
                    if owner is not None and hasattr(owner, "r_feature") and owner not in hooks:
                        hooks.append(owner)
            rec = dict(loss_reg=float(sum(float(h.r_feature.detach()) for h in hooks)))
            named = dict(capture.model.named_parameters())
            for key in SWIN_SAMPLED:
                rec[f"grad::{key}"] = t2n(named[key].grad[:SAMPLE_ROWS]) if named[key].grad is not None else None
            if getattr(capture, "all_affine", False):  # EVERY LayerNorm weight / bias gradient, whole tensors, in named_parameters order
                ln_names = [f"{mn}.{pn}" for mn, mod in capture.model.named_modules() if isinstance(mod, nn.LayerNorm)
                            for pn in ("weight", "bias")]
                rec["affine_names"] = np.array(ln_names)
                rec["affine_sizes"] = np.array([named[n].numel() for n in ln_names])
                rec["affine_grads"] = np.concatenate([t2n(named[n].grad).ravel() if named[n].grad is not None
                                                      else np.zeros(named[n].numel(), np.float32) for n in ln_names])
            r = _real(self, *a, **k)
            for key in SWIN_SAMPLED:
                rec[f"param::{key}"] = t2n(named[key][:SAMPLE_ROWS])
            capture.steps.append(rec)
            return r

        saved_steps[cls] = real
        cls.step = step

    from vitta_amd.data import SyntheticVideoDataset

    def fake_dataset(args, split="val", dataset_type=None):
        views = args.n_augmented_views if dataset_type == "tta" else 1
        ds = SyntheticVideoDataset(n_videos, views, args.clip_length, args.input_size, args.num_classes, "swin",
                                   seed0=getattr(capture, "seed0", 800))
        return _Perturbed(ds, perturb, perturb_seed) if perturb else ds

    B.get_dataset_videoswin = fake_dataset
    logger = logging.getLogger("refgen")
    logger.addHandler(logging.NullHandler())
    try:
        res = B.tta_standard(Wrap(model_origin), nn.CrossEntropyLoss(), args=args, logger=logger, writer=None)
    finally:
        B.cp.deepcopy = real_deepcopy
        B.compute_pred_consis = real_consis
        for cls, real in saved_steps.items():
            cls.step = real
    return res


def gen_tta_swin(tag="tta3_swin", T=16, size=64, n_videos=3, views=2, window=(8, 7, 7), K=101, dataset="ucf101", trials=3,
                 all_affine=False, seed0=800):
    """A11/A7 for Video Swin-B: online steps through the reference's tta_standard, both optimizers.
    tta3_swin: three steps at 64^2 (stage-2 / 3 windows clamp to 4 x 4 / 2 x 2).  Round 6: tta1_224_swin = ONE step at BASELINE
    config 3's real size (2 views x 16 frames x 224^2: window (8, 7, 7) unclamped on the 14 x 14 / 7 x 7 planes of stages 2 / 3, where
    the 42 hooked LayerNorms live, shift mask in the backward) and tta1_c5_swin = ONE step at config 5's shape (4 views x 32 frames x
    112^2, K = 174, window (16, 7, 7): recognizer3d.py:36-40); both also record EVERY LayerNorm affine gradient whole (all_affine)
    with per-tensor noise floors, trials perturbed re-runs (odd ones also perturb every parameter by 1e-7 relative)."""
    from utils.opts import get_opts
    from utils.norm_stats_utils import ComputeNormStatsHook
    from utils.BNS_utils import choose_layers
    n_steps = n_videos
    out = {}
    for mode in ("sgd", "adam"):
        ref, _ = ref_swin(K, 0, window_size=window)
        lns = [m for _, m in choose_layers(ref, [nn.LayerNorm])][1:]
        hooks = [ComputeNormStatsHook(m, clip_len=T, stat_type="spatiotemp", before_norm=False, batch_size=1) for m in lns]
        with torch.no_grad():
            ref(H.seeded_randn((1, 2, 3, T, size, size), 1000))
        g = torch.Generator().manual_seed(77)
        means = [t2n(h.batch_mean + 0.05 * torch.randn(h.batch_mean.shape, generator=g)) for h in hooks]
        vars_ = [t2n(h.batch_var * (1 + 0.2 * torch.rand(h.batch_var.shape, generator=g))) for h in hooks]
        for h in hooks:
            h.close()
        with tempfile.TemporaryDirectory() as tmp:
            mp, vp = H.write_stat_files(tmp, means, vars_)
            args = get_opts()
            args.arch, args.dataset, args.clip_length, args.workers = "videoswintransformer", dataset, T, 0
            args.input_size, args.scale_size, args.verbose, args.batch_size = size, size, False, 1
            args.num_clips, args.test_crops, args.frame_uniform, args.frame_interval = 1, 1, True, 2
            args.patch_size, args.window_size = (2, 4, 4), tuple(window)
            args.n_augmented_views = views
            args.lambda_pred_consis, args.momentum_mvg, args.chosen_blocks = 0.05, 0.05, SWIN_BLOCKS
            args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
            args.num_classes, args.gpus, args.result_dir = K, [0], tmp
            args.update_only_bn_affine = mode == "adam"
            args.lr = 1e-5 if mode == "sgd" else 1e-3
            cap = _Capture()
            cap.all_affine, cap.seed0 = all_affine, seed0
            torch.manual_seed(4321)
            import time as _time
            t0 = _time.time()
            res = run_reference_tta_swin(args, ref, n_videos, cap)
            print(f"  {tag} {mode}: reference run {_time.time() - t0:.1f} s", flush=True)
            caps = []
            for trial in range(trials):
                c2 = _Capture()
                c2.all_affine, c2.seed0 = all_affine, seed0
                c2.perturb_params = 1e-7 if (all_affine and trial % 2) else 0.0
                torch.manual_seed(4321)
                run_reference_tta_swin(args, ref, n_videos, c2, perturb=1e-7, perturb_seed=90000 + 1000 * trial)
                caps.append(c2)
        assert len(cap.steps) == n_steps and len(cap.drop_masks) == n_steps, (len(cap.steps), len(cap.drop_masks))
        per = len(cap.droppath) // n_steps
        for i in range(n_steps):
            a = cap.steps[i]
            noise = {}

            def bump(key, val):
                noise[key] = max(noise.get(key, 0.0), float(val))

            for c2 in caps:
                assert all((x == y).all() for x, y in zip(cap.droppath, c2.droppath))
                b = c2.steps[i]
                bump("loss_reg", abs(a["loss_reg"] - b["loss_reg"]))
                bump("loss_consis", np.abs(cap.consis[i] - c2.consis[i]))
                bump("eval_logits", np.abs(cap.eval_logits[i] - c2.eval_logits[i]).max())
                for key in SWIN_SAMPLED:
                    if a.get(f"grad::{key}") is not None:
                        bump(f"grad::{key}", np.abs(a[f"grad::{key}"] - b[f"grad::{key}"]).max())
                    bump(f"param::{key}", np.abs(a[f"param::{key}"] - b[f"param::{key}"]).max())
            if all_affine:  # per-tensor floor of every LayerNorm affine gradient: worst |difference| over the perturbed re-runs
                offs = np.concatenate([[0], np.cumsum(a["affine_sizes"])])
                fl = np.zeros(len(a["affine_sizes"]))
                for c2 in caps:
                    dlt = np.abs(a["affine_grads"] - c2.steps[i]["affine_grads"])
                    fl = np.maximum(fl, np.array([dlt[offs[j]:offs[j + 1]].max() for j in range(len(fl))]))
                out[f"{mode}_step{i}_affine_noise"] = fl
            for key, val in noise.items():
                out[f"{mode}_step{i}_noise_{key}"] = np.array(val)
            for k, v in a.items():
                if v is not None:
                    out[f"{mode}_step{i}_{k}"] = np.asarray(v)
            out[f"{mode}_step{i}_loss_consis"] = cap.consis[i]
            out[f"{mode}_step{i}_eval_logits"] = cap.eval_logits[i]
            bits, shape = H.pack_mask(cap.drop_masks[i])
            out[f"{mode}_step{i}_dropmask"], out[f"{mode}_step{i}_dropmask_shape"] = bits, shape
            out[f"{mode}_step{i}_droppath"] = np.stack(cap.droppath[i * per:(i + 1) * per])
        if mode == "sgd":
            out["src_means"], out["src_vars"] = np.concatenate(means), np.concatenate(vars_)
            out["src_channels"] = np.array([len(m) for m in means])
    out["sampled_params"] = np.array(SWIN_SAMPLED)
    out["sample_rows"] = np.array(SAMPLE_ROWS)
    out["config"] = np.array(json.dumps(dict(T=T, size=size, n_videos=n_videos, batch_size=1, seed0=seed0, lr_sgd=1e-5,
                                             lr_adam=1e-3, momentum_mvg=0.05, lambda_pred_consis=0.05, views=views,
                                             window=list(window), K=K, dataset=dataset, n_steps=n_steps)))
    save(f"{tag}.npz", **out)


def gen_tta224_swin():
    gen_tta_swin(tag="tta1_224_swin", T=16, size=224, n_videos=1, views=2, window=(8, 7, 7), K=101, trials=NOISE_TRIALS,
                 all_affine=True, seed0=810)


def gen_tta_c5():
    gen_tta_swin(tag="tta1_c5_swin", T=32, size=112, n_videos=1, views=4, window=(16, 7, 7), K=174, dataset="somethingv2",
                 trials=NOISE_TRIALS, all_affine=True, seed0=820)


def gen_bns():
    """N3: BNFeatureHook (stat_reg='BNS'): BN-input statistics vs the layer's running statistics, 3 steps."""
    from utils.BNS_utils import BNFeatureHook
    out = {}
    cases = {"bn2d": (nn.BatchNorm2d, 8, (16, 8, 7, 7)), "bn1d_rows": (nn.BatchNorm1d, 16, (64, 16)),
             "bn1d_nct": (nn.BatchNorm1d, 16, (2, 16, 8))}
    for name, (cls, c, shape) in cases.items():
        for reg in ("l1_loss", "mse_loss", "kld"):
            mod = cls(c).eval()
            g = torch.Generator().manual_seed(9)
            with torch.no_grad():
                mod.running_mean.copy_(torch.randn(c, generator=g) * 0.3)
                mod.running_var.copy_(torch.rand(c, generator=g) + 0.5)
            hook = BNFeatureHook(mod, reg_type=reg, running_manner=True, use_src_stat_in_reg=True, momentum=0.1)
            for step in range(3):
                x = H.channel_feature(shape, 300 + step, 1, offset_scale=0.5).requires_grad_(True)
                mod(x)
                (gx,) = torch.autograd.grad(hook.r_feature, x)
                key = f"{name}_{reg}_{step}"
                out[key + "_r"], out[key + "_mean"], out[key + "_var"], out[key + "_gx"] = (
                    t2n(hook.r_feature), t2n(hook.mean), t2n(hook.var), t2n(gx))
            hook.close()
    save("bns.npz", **out)


SECTIONS = dict(tta224=gen_tta224, l2ops=gen_l2ops, layers=gen_layers, tam=gen_tam, tanet=gen_tanet, tta=gen_tta, sampler=gen_sampler,
                opts=gen_opts, dp=gen_dp, swin=gen_swin, tta_swin=gen_tta_swin, tta224_swin=gen_tta224_swin, tta_c5=gen_tta_c5, bns=gen_bns, episodic=gen_episodic, data=gen_data, epoch=gen_epoch)

if __name__ == "__main__":
    for n in (ARGV or list(SECTIONS)):
        print(f"== {n}")
        SECTIONS[n]()
