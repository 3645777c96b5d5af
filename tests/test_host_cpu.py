"""Host logic of the product against goldens from the reference -- CPU, no GPU needed.
The HIP launches are replaced by tests/oracle_backend.py (test infrastructure)."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn

import helpers as H
from oracle.oracle_backend import OracleBackend
from vitta_amd import data, tta
from vitta_amd.bns_utils import choose_layers, collect_bn_params, freeze_except_bn


def assert_logits_close(got, ref, frac=1e-3):
    """fp32 tolerance of the path (stated): |delta| <= 1e-3 * max|reference logits|, identical top-1."""
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= frac * scale, ((got - ref).abs().max().item(), scale)
    # identical top-1, except where the reference itself has a tie inside the tolerance band
    for row_got, row_ref in zip(got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])):
        a, b = int(row_got.argmax()), int(row_ref.argmax())
        assert a == b or (row_ref[b] - row_ref[a]).item() <= 2 * frac * scale, (a, b)


@pytest.fixture(scope="module")
def tanet11():
    return H.build_tanet(11, 8, 0)


def test_layer_selection_matches_reference(tanet11):
    g = H.golden("layers_tanet.npz")
    model = tta.SingleDeviceParallel(tanet11)
    chosen = choose_layers(model, [nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d])
    assert [n for n, _ in chosen] == [str(s) for s in g["names"]]
    assert [type(m).__name__ for _, m in chosen] == [str(s) for s in g["kinds"]]
    assert len(chosen) == 85
    args = H.tanet_args("/tmp")
    hooked = [i for i, _, _ in tta.select_hooked(args, chosen)]
    assert hooked == g["hooked"].tolist() and len(hooked) == 47
    bn2d_hooked = [i for i in hooked if g["kinds"][i] == "BatchNorm2d"]
    assert len(bn2d_hooked) == 29 and len(hooked) - len(bn2d_hooked) == 18
    assert sorted(int(g["stat_idx"][i]) for i in bn2d_hooked) == list(range(24, 53))


def test_affine_param_collection(tanet11):
    import copy
    model = copy.deepcopy(tanet11)
    kinds = [nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d]
    freeze_except_bn(model, kinds)
    params, names = collect_bn_params(model, kinds)
    assert len(params) == 170 and sum(p.numel() for p in params) == 55520
    assert all(p.requires_grad for p in params)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 55520


def test_sampler_matches_reference(monkeypatch):
    g = H.golden("sampler.npz")
    # the fixture was captured by running the reference on numpy 2.x (linspace(dtype=int) floors)
    monkeypatch.setattr(data, "LINSPACE_INT_MODE", "floor")
    for key in g.files:
        if key == "numpy_version":
            continue
        parts = key.split("_")
        if parts[0] == "tta":
            style = "_".join(parts[1:-3])
            T, n, V = int(parts[-3][1:]), int(parts[-2][1:]), int(parts[-1][1:])
            got = data.tta_view_indices(n, T, V, style)
        else:
            ts, T, n = parts[1], int(parts[2][1:]), int(parts[3][1:])
            got = data.test_indices(n, T, ts)
        np.testing.assert_array_equal(np.asarray(got), g[key], err_msg=key)
    # default mode = numpy 1.19.5 truncation (the reference's pinned environment): start offsets of a
    # video shorter than T stay at 0 instead of wrapping to the last frame
    monkeypatch.setattr(data, "LINSPACE_INT_MODE", "trunc")
    assert data.tta_view_indices(5, 8, 2).tolist() == [1, 1, 2, 2, 3, 4, 4, 5] * 2
    # SURVEY 8c item 9 probe values
    assert data.tta_view_indices(100, 8, 2).tolist() == [1, 13, 26, 38, 51, 63, 76, 88, 12, 24, 37, 49, 62, 74, 87, 99]
    assert data.test_indices(100, 8).tolist() == [7, 19, 32, 44, 57, 69, 82, 94]


def test_opts_surface_matches_reference():
    from vitta_amd.opts import get_opts
    ref = json.load(open(H.GOLDEN_DIR + "/opts_defaults.json"))
    mine = {k: repr(v) for k, v in vars(get_opts([])).items()}
    extensions = {"hip_graph", "overlap_eval", "device_preprocess", "wmsa_bf16", "dense_bf16", "prefetch_input"}  # flags this build adds on top of the reference surface
    assert set(mine) - set(ref) == extensions
    mine = {k: v for k, v in mine.items() if k not in extensions}
    assert set(ref) == set(mine)
    for k in ref:
        assert mine[k] == ref[k], (k, mine[k], ref[k])


def test_tanet_forward_matches_reference(tanet11):
    g = H.golden("tanet_fwd.npz")
    x = H.seeded_randn((2, 8, 3, 64, 64), 21)
    bn2d = [(n, m) for n, m in tanet11.named_modules() if isinstance(m, nn.BatchNorm2d)]
    assert [n for n, _ in bn2d] == [str(s) for s in g["names"]]
    from vitta_amd.norm_stats import ComputeNormStatsHook
    hooks = [ComputeNormStatsHook(m, clip_len=8, stat_type="spatiotemp", before_norm=False, batch_size=2,
                                  backend=OracleBackend()) for _, m in bn2d]
    with torch.no_grad():
        logits = tanet11(x)
    for h in hooks:
        h.close()
    assert_logits_close(logits, torch.from_numpy(g["logits"]))
    means = torch.cat([h.batch_mean for h in hooks])
    vars_ = torch.cat([h.batch_var for h in hooks])
    # 53 layers deep in fp32, different (equivalent) op order in TAM and the residual add
    torch.testing.assert_close(means, torch.from_numpy(g["means"]), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(vars_, torch.from_numpy(g["vars"]), rtol=2e-3, atol=1e-5)


def run_product_tta(g, mode, tmp_path, device, backend_factory, batch_size=1, use_engine=None):
    """Drive the product's adapter exactly like tta_standard does, with the recorded dropout masks."""
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    model = H.build_tanet(101, T, 0)
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    means = [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    vars_ = [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    mp, vp = H.write_stat_files(str(tmp_path), means, vars_)
    args = H.tanet_args(tmp_path, clip_length=T, input_size=size, batch_size=batch_size,
                        spatiotemp_mean_clean_file=mp, spatiotemp_var_clean_file=vp,
                        update_only_bn_affine=(mode == "adam"), lr=cfg["lr_sgd"] if mode == "sgd" else cfg["lr_adam"])
    n_steps = int(cfg.get("n_steps", 3))
    masks = [H.unpack_mask(g[f"{mode}_step{i}_dropmask"], g[f"{mode}_step{i}_dropmask_shape"]) for i in range(n_steps)]
    wrapped = tta.SingleDeviceParallel(model).to(device)
    tta.BACKEND_FACTORY = backend_factory
    try:
        adapter = tta.ViTTAAdapter(wrapped, args, use_engine=use_engine)
    finally:
        tta.BACKEND_FACTORY = None
    adapter.model.module.base_model.fc = H.ReplayDropout(0.8, masks)
    tta_set = data.SyntheticVideoDataset(cfg["n_videos"], 2, T, size, 101, "tanet", seed0=cfg["seed0"])
    eval_set = data.SyntheticVideoDataset(cfg["n_videos"], 1, T, size, 101, "tanet", seed0=cfg["seed0"])
    records = []
    for step in range(n_steps):
        idx = range(step * batch_size, (step + 1) * batch_size)
        x = torch.stack([tta_set[i][0] for i in idx]).to(device)
        adapter.set_adapt_mode()
        _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
        named = dict(adapter.model.named_parameters())
        adapter.close_hooks()
        ev = torch.stack([eval_set[i][0] for i in idx]).to(device)
        logits = adapter.evaluate(adapter.shape_eval_input(ev))
        adapter.add_hooks_back()
        records.append(dict(loss_reg=float(loss_reg), loss_consis=float(loss_consis), eval_logits=logits.cpu(),
                            params={k: named[k].detach().cpu().clone() for k in map(str, g["sampled_params"])},
                            grads={k: (named[k].grad.detach().cpu().clone() if named[k].grad is not None else None)
                                   for k in map(str, g["sampled_params"])},
                            param_sum=float(sum(float(p.double().sum()) for p in named.values()))))
    return records


# Tolerances of the step-level parity tests.  The problem is ill-conditioned by construction: L1 terms
# give sign(.) gradients (a channel whose EMA sits within round-off of its source statistic flips its
# whole contribution), Adam's first update is lr * sign(g), and three successive steps compound it.
# The fixture therefore carries the reference's OWN noise floor: the reference re-run with the input
# clips perturbed by 1e-7 relative (fp32 round-off) and the same dropout masks.  Each quantity must
# agree with the reference within max(BASE, 4 x its noise floor -- one perturbation is one sample): BASE is the tight fp32 bound that holds
# on the first step (identical weights), the floor takes over on later steps.  On the CPU the product
# stays 10-100x inside the floor; top-1 must be identical unless the reference itself has a tie.
BASE = dict(loss_rel=1e-5, logit_frac=2e-3, grad_frac=5e-3, param_lr_mult=0.05)
TOL_CPU = BASE


def assert_logits_close_abs(got, ref, bound):
    assert (got - ref).abs().max().item() <= bound, ((got - ref).abs().max().item(), bound)
    for row_got, row_ref in zip(got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])):
        a, b = int(row_got.argmax()), int(row_ref.argmax())
        assert a == b or (row_ref[b] - row_ref[a]).item() <= 2 * bound, (a, b)


def _norm_affine(name):
    """weight / bias of a BatchNorm / LayerNorm (TANet: bnK, downsample.1, the TAM branches' G.1 / L.1; Swin: normK, norm)."""
    import re
    owner, _, leaf = name.rpartition(".")
    return leaf in ("weight", "bias") and re.search(r"(^|\.)(bn\d*|norm\d*|downsample\.1|G\.1|L\.1)$", owner) is not None


def check_tta_records(g, mode, records, base, floor_mult=4.0, outliers=None):
    """Every sampled quantity within max(BASE bound, floor_mult x the reference's own noise floor).  The floors of the TANet
    fixtures are the worst of EIGHT perturbed re-runs of the reference per step (inputs, and in every second draw also every
    parameter, perturbed by 1e-7 relative: tools/refgen/gen_golden.py).
    outliers = (count, cap) (GPU suites, split-bf16 arithmetic only): on steps AFTER the first, at most `count` sampled
    gradient tensors of a step may exceed their own bound, by no more than `cap` x max|g| -- every other tensor stays under
    its strict bound, and the exact-fp32 arithmetic runs with outliers=None.  Why any: the alignment loss is L1, a hooked
    BatchNorm channel contributes +-momentum / C to its d beta (and the like to d gamma) through sign(ema - source); a channel
    whose EMA sits within round-off of its source statistic flips with ANY change of summation order, and one flip among the
    four sampled channels of layer4.2.net.bn3.bias is 14 % of that tensor's max|g| (2 x 0.1 / 2048 against 7e-4) -- a
    quantum the eight reference re-runs either show for a tensor or do not (tools/debug/bz2_step_probe.py prints the table
    for both arithmetic forms: the exact-fp32 kernels show 12 % on layer1.0.net.bn1.weight in the same step).  Round 3 covered
    this with a median-of-others floor (up to 120 % of max|g| on the third step); a counted, capped allowance keeps every
    other tensor's check strict."""
    rows = int(g["sample_rows"])
    cfg = json.loads(str(g["config"]))
    lr = cfg["lr_sgd"] if mode == "sgd" else cfg["lr_adam"]
    report = []
    for i, rec in enumerate(records):
        k = f"{mode}_step{i}_"
        for q in ("loss_reg", "loss_consis"):
            ref = float(g[k + q])
            bound = max(base["loss_rel"] * abs(ref), floor_mult * float(g[k + "noise_" + q])) + 1e-7
            assert abs(rec[q] - ref) <= bound, (i, q, rec[q], ref, bound)
            report.append((i, q, abs(rec[q] - ref), bound))
        ref = torch.from_numpy(g[k + "eval_logits"])
        bound = max(base["logit_frac"] * ref.abs().max().item(), floor_mult * float(g[k + "noise_eval_logits"]))
        assert_logits_close_abs(rec["eval_logits"], ref, bound)
        report.append((i, "eval_logits", (rec["eval_logits"] - ref).abs().max().item(), bound))
        over = []
        for name, gr in rec["grads"].items():
            key = k + f"grad::{name}"
            if key not in g.files:
                assert gr is None and mode == "adam", name  # frozen parameter in affine-only mode
                continue
            ref = torch.from_numpy(g[key])
            bound = max(base["grad_frac"] * ref.abs().max().item(), floor_mult * float(g[k + f"noise_grad::{name}"])) + 1e-10
            err = (gr[:rows] - ref).abs().max().item()
            if err > bound and i > 0 and outliers is not None:
                # (the allowance covers what the sign-flip argument covers -- affine tensors of normalisation layers, where a
                # flipped L1 term lands as a quantum --, not convolution / dense weights or anything else)
                # ... any other tensor may take the step's one allowance only while it stays within twice its own bound (a flipped term
                # reaches the tensors upstream of its layer attenuated)
                cap = float(outliers[1]) * ref.abs().max().item() if _norm_affine(name) else 2.0 * bound
                assert err <= cap, (i, name, err, bound, "beyond the outlier cap")
                over.append((name, err, bound))
            else:
                assert err <= bound, (i, name, err, bound)
            report.append((i, "grad " + name, err, bound))
        assert outliers is None or len(over) <= int(outliers[0]), (i, "sampled gradients over their own bound", over)
        for name, p in rec["params"].items():
            ref = torch.from_numpy(g[k + f"param::{name}"])
            gkey = k + f"grad::{name}"
            gmax = 1.0 if (mode == "adam" or gkey not in g.files) else max(1e-3, float(np.abs(g[gkey]).max()))
            bound = max(base["param_lr_mult"] * lr * gmax, floor_mult * float(g[k + f"noise_param::{name}"])) + 1e-7
            err = (p[:rows] - ref).abs().max().item()
            assert err <= bound, (i, name, err, bound)
    return report


@pytest.mark.parametrize("mode", ["sgd", "adam"])
@pytest.mark.parametrize("use_engine", [True, False])
def test_three_tta_steps_match_reference(tmp_path, mode, use_engine):
    """the reference's own tta_standard, three online steps, both optimizer modes."""
    g = H.golden("tta3.npz")
    recs = run_product_tta(g, mode, tmp_path, torch.device("cpu"), OracleBackend, use_engine=use_engine)
    check_tta_records(g, mode, recs, TOL_CPU)


def test_batch_of_two_matches_reference(tmp_path):
    """the reference with batch_size=2 (what two data-parallel ranks reproduce)."""
    g = H.golden("tta3_bz2.npz")
    recs = run_product_tta(g, "sgd", tmp_path, torch.device("cpu"), OracleBackend, batch_size=2)
    check_tta_records(g, "sgd", recs, TOL_CPU)


def _one_step_grads(tmp_path, use_engine, device, backend_factory, **over):
    """One adaptation step of the small TANet; returns (loss_reg, {name: grad}, {name: param delta})."""
    g = H.golden("tta3.npz")
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    model = H.build_tanet(101, T, 0)
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    means = [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    vars_ = [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    mp, vp = H.write_stat_files(str(tmp_path), means, vars_)
    args = H.tanet_args(tmp_path, clip_length=T, input_size=size, batch_size=1, spatiotemp_mean_clean_file=mp,
                        spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=cfg["lr_adam"], **over)
    tta.BACKEND_FACTORY = backend_factory
    try:
        adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(device), args, use_engine=use_engine)
    finally:
        tta.BACKEND_FACTORY = None
    adapter.model.module.base_model.fc = torch.nn.Identity()  # no dropout: the two runs must see the same forward
    before = {k: v.detach().clone() for k, v in adapter.model.named_parameters() if v.requires_grad}
    views = 2 if over.get("if_sample_tta_aug_views", True) else 1
    x = data.SyntheticVideoDataset(2, views, T, size, 101, "tanet", seed0=cfg["seed0"])[0][0].unsqueeze(0).to(device)
    adapter.set_adapt_mode()
    _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
    named = {k: v for k, v in adapter.model.named_parameters() if v.requires_grad}
    grads = {k: v.grad.detach().cpu().clone() for k, v in named.items()}
    delta = {k: (v.detach() - before[k]).cpu() for k, v in named.items()}
    return float(loss_reg), loss_consis, grads, delta


@pytest.mark.parametrize("over", [dict(if_pred_consistency=False), dict(if_sample_tta_aug_views=False)])
def test_statistics_loss_alone_still_adapts(tmp_path, over):
    """The paper's 'no consistency' ablation (corpus/basics.py:660-668: loss = loss_reg): with the batched engine the
    statistics gradient is injected by nodes of the model's graph, which loss_reg.backward() must still reach.
    Engine == stand-alone hooks (the reference's formulation), and both really update the affine parameters."""
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    lr_e, lc_e, g_e, d_e = _one_step_grads(tmp_path / "a", True, torch.device("cpu"), OracleBackend, **over)
    lr_h, lc_h, g_h, d_h = _one_step_grads(tmp_path / "b", False, torch.device("cpu"), OracleBackend, **over)
    assert lc_e is None and lc_h is None
    assert abs(lr_e - lr_h) <= 1e-5 * abs(lr_h)
    tot_e = sum(float(v.abs().sum()) for v in g_e.values())
    tot_h = sum(float(v.abs().sum()) for v in g_h.values())
    assert tot_h > 0 and tot_e > 0.5 * tot_h, (tot_e, tot_h)
    for k in g_h:
        bound = 5e-3 * float(g_h[k].abs().max()) + 1e-9
        assert float((g_e[k] - g_h[k]).abs().max()) <= bound, k
    assert sum(float(v.abs().sum()) for v in d_e.values()) > 0


@pytest.mark.parametrize("reg_type", ["l1_loss", "mse_loss"])
def test_before_norm_hooks_on_the_batched_engine_equal_the_stand_alone_hooks(tmp_path, reg_type):
    """--before_norm (utils/norm_stats_utils.py:185: the hooked feature is the norm layer's INPUT): the batched engine takes such
    hooks since round 4 (it only ever sees a feature tensor) -- same statistics loss and the same gradients as the per-layer
    hooks that follow the reference's formulation; and the gradients differ from the after-norm configuration's (the option is
    honoured, not ignored)."""
    for d in "abc":
        (tmp_path / d).mkdir()
    over = dict(before_norm=True, reg_type=reg_type)
    lr_e, _, g_e, d_e = _one_step_grads(tmp_path / "a", True, torch.device("cpu"), OracleBackend, **over)
    lr_h, _, g_h, _ = _one_step_grads(tmp_path / "b", False, torch.device("cpu"), OracleBackend, **over)
    lr_a, _, g_a, _ = _one_step_grads(tmp_path / "c", True, torch.device("cpu"), OracleBackend, reg_type=reg_type)
    assert abs(lr_e - lr_h) <= 1e-5 * abs(lr_h), (lr_e, lr_h)
    for k in g_h:
        bound = 5e-3 * float(g_h[k].abs().max()) + 1e-9
        assert float((g_e[k] - g_h[k]).abs().max()) <= bound, k
    assert abs(lr_e - lr_a) > 1e-3 * abs(lr_a)
    assert sum(float(v.abs().sum()) for v in d_e.values()) > 0


def test_reference_import_paths_resolve():
    """Every module path DESIGN.md section 1 lists as the Python boundary imports and exposes the reference's names."""
    import importlib
    for mod, names in {
        "utils.opts": ["get_opts"], "corpus.main_eval": ["eval"],
        "corpus.basics": ["tta_standard", "test_time_adapt", "validate", "validate_brief", "compute_statistics", "get_model"],
        "utils.norm_stats_utils": ["CombineNormStatsRegHook_onereg", "ComputeNormStatsHook", "compute_regularization"],
        "utils.BNS_utils": ["choose_layers", "freeze_except_bn", "collect_bn_params", "BNFeatureHook"],
        "utils.pred_consistency_utils": ["compute_pred_consis"], "utils.utils_": ["MovingAverageTensor"],
        "models.tanet_models.tanet": ["TSN"], "models.tanet_models.temporal_module": ["TAM", "TemporalBottleneck"],
        "models.tanet_models.basic_ops": ["ConsensusModule"],
        "models.videoswintransformer_models.recognizer3d": ["Recognizer3D"],
        "models.videoswintransformer_models.swin_transformer": ["SwinTransformer3D", "WindowAttention3D", "PatchMerging"],
        "models.videoswintransformer_models.i3d_head": ["I3DHead"],
    }.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)


def test_device_prefetcher_on_a_cpu_device_is_the_plain_iterator():
    """vitta_amd/prefetch.py: same batches, same order, same StopIteration; `ahead()` is a no-op without a GPU."""
    import torch
    from vitta_amd.prefetch import DevicePrefetcher
    batches = [(torch.full((2, 3), float(i)), torch.tensor([i, i + 1])) for i in range(4)]
    pf = DevicePrefetcher(iter(batches), torch.device("cpu"))
    got = []
    for _ in range(4):
        got.append(next(pf))
        pf.ahead()
    assert all(a is b for ga, ba in zip(got, batches) for a, b in zip(ga, ba))
    import pytest
    with pytest.raises(StopIteration):
        next(pf)
    assert pf.uploads == 0
