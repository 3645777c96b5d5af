"""Video Swin-B restatement + LayerNorm statistics hooks against goldens from the reference (CPU)."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn

import helpers as H
from oracle.oracle_backend import OracleBackend
from test_host_cpu import BASE, assert_logits_close, check_tta_records
from vitta_amd import data, scripts, tta
from vitta_amd.bns_utils import choose_layers, collect_bn_params, freeze_except_bn

SWIN_BLOCKS = ["module.backbone.layers.2", "module.backbone.layers.3", "module.backbone.norm"]


@pytest.fixture(scope="module")
def swin11():
    return H.build_swin(11, 0)


def test_swin_layer_selection_matches_reference(swin11):
    g = H.golden("layers_swin.npz")
    chosen = choose_layers(tta.SingleDeviceParallel(swin11), [nn.LayerNorm])
    assert [n for n, _ in chosen] == [str(s) for s in g["names"]] and len(chosen) == 53
    args = scripts.swin_ucf101_args([])
    cands = tta.candidate_layers_for(args, tta.SingleDeviceParallel(swin11))
    assert len(cands) == 52
    hooked = [i for i, _, _ in tta.select_hooked(args, cands)]
    assert hooked == g["hooked"].tolist() and len(hooked) == 42
    assert sum(m.normalized_shape[0] for _, _, m in tta.select_hooked(args, cands)) == 25600
    kinds = [nn.LayerNorm]
    import copy
    model = copy.deepcopy(swin11)
    freeze_except_bn(model, kinds)
    params, _ = collect_bn_params(model, kinds)
    assert sum(p.numel() for p in params) == 57600


def test_patch_gather_equals_the_composed_slices_forward_and_backward():
    """swin.PatchGather (PatchMerging's cat of the four pixel parities, swin_transformer.py:281-286, as one autograd node) ==
    the slices + cat it replaces, values and gradient bit for bit."""
    import torch
    from vitta_amd.swin import PatchGather
    for shape in [(2, 4, 8, 6, 16), (1, 2, 2, 2, 4), (3, 1, 14, 14, 8)]:
        x = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
        ref = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
        w = torch.randn_like(ref)
        (gr,) = torch.autograd.grad((ref * w).sum(), x)
        x2 = x.detach().clone().requires_grad_(True)
        out = PatchGather.apply(x2)
        (g2,) = torch.autograd.grad((out * w).sum(), x2)
        assert torch.equal(out, ref) and torch.equal(g2, gr)


def test_swin_forward_matches_reference(swin11):
    g = H.golden("swin_fwd.npz")
    from vitta_amd.norm_stats import ComputeNormStatsHook
    lns = [m for _, m in choose_layers(swin11, [nn.LayerNorm])][1:]
    hooks = [ComputeNormStatsHook(m, clip_len=16, stat_type="spatiotemp", before_norm=False, batch_size=1,
                                  backend=OracleBackend()) for m in lns]
    with torch.no_grad():
        vid, view = swin11(H.seeded_randn((1, 2, 3, 16, 112, 112), 31))
    for h in hooks:
        h.close()
    assert_logits_close(view, torch.from_numpy(g["view"]), 1e-4)
    assert_logits_close(vid, torch.from_numpy(g["vid"]), 1e-4)
    torch.testing.assert_close(torch.cat([h.batch_mean for h in hooks]), torch.from_numpy(g["means"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(torch.cat([h.batch_var for h in hooks]), torch.from_numpy(g["vars"]), rtol=1e-4, atol=1e-6)


def run_product_tta_swin(g, mode, tmp_path, device, backend_factory, use_engine=None, all_affine=False):
    """Replays a reference-generated Video Swin fixture (tools/refgen/gen_golden.py tta_swin / tta224_swin / tta_c5: the reference's own
    tta_standard) through the product: same weights, clips, dropout / DropPath masks, optimizer.  The fixture's config names the
    clip (T, size, views), the window, the class count and the number of steps.  all_affine: every LayerNorm weight / bias gradient of
    each step is returned whole (records[i]["affine"] = {name: tensor})."""
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    views, window, K = cfg.get("views", 2), tuple(cfg.get("window", (8, 7, 7))), cfg.get("K", 101)
    n_steps = cfg.get("n_steps", 3)
    model = H.build_swin(K, 0, window_size=window)
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    mp, vp = H.write_stat_files(str(tmp_path), [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    args = scripts.swin_ucf101_args([])
    args.datatype, args.input_size, args.scale_size, args.workers, args.verbose = "synthetic", size, size, 0, False
    args.result_dir, args.num_classes, args.batch_size = str(tmp_path), K, 1
    args.dataset = cfg.get("dataset", "ucf101")
    args.clip_length, args.n_augmented_views, args.window_size = T, views, window
    args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
    args.update_only_bn_affine = mode == "adam"
    args.lr = cfg["lr_sgd"] if mode == "sgd" else cfg["lr_adam"]
    tta.BACKEND_FACTORY = backend_factory
    try:
        adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(device), args, use_engine=use_engine)
    finally:
        tta.BACKEND_FACTORY = None
    masks = [H.unpack_mask(g[f"{mode}_step{i}_dropmask"], g[f"{mode}_step{i}_dropmask_shape"]) for i in range(n_steps)]
    adapter.model.module.cls_head.dropout = H.ReplayDropout(0.5, masks)
    tape = H.MaskTape([m for i in range(n_steps) for m in g[f"{mode}_step{i}_droppath"]])
    from vitta_amd.swin import DropPath, SwinTransformerBlock3D
    for blk in adapter.model.modules():
        if isinstance(blk, SwinTransformerBlock3D) and isinstance(blk.drop_path, DropPath):
            blk.drop_path = H.ReplayDropPath(blk.drop_path.drop_prob, tape)
    tta_set = data.SyntheticVideoDataset(cfg["n_videos"], views, T, size, K, "swin", seed0=cfg["seed0"])
    eval_set = data.SyntheticVideoDataset(cfg["n_videos"], 1, T, size, K, "swin", seed0=cfg["seed0"])
    records = []
    for step in range(n_steps):
        x = tta_set[step][0].unsqueeze(0).to(device)
        adapter.set_adapt_mode()
        _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
        named = dict(adapter.model.named_parameters())
        affine = None
        if all_affine:
            affine = {str(n): (named[str(n)].grad.detach().cpu().clone() if named[str(n)].grad is not None else None)
                      for n in g[f"{mode}_step{step}_affine_names"]}
        adapter.close_hooks()
        logits = adapter.evaluate(adapter.shape_eval_input(eval_set[step][0].unsqueeze(0).to(device)))
        adapter.add_hooks_back()
        names = [str(s) for s in g["sampled_params"]]
        records.append(dict(loss_reg=float(loss_reg), loss_consis=float(loss_consis), eval_logits=logits.cpu(),
                            params={k: named[k].detach().cpu().clone() for k in names},
                            grads={k: (named[k].grad.detach().cpu().clone() if named[k].grad is not None else None)
                                   for k in names}, affine=affine))
    assert tape.pos == len(tape.masks)
    return records


def check_affine_gradients(g, mode, rec, step=0, grad_frac=5e-3, floor_mult=2.0, max_over=0, cap=0.30, cos_min=0.9999):
    """EVERY LayerNorm weight / bias gradient of a step, whole tensors, against the reference's (fixtures generated with all_affine:
    tta1_224_swin, tta1_c5_swin).  Per tensor: max |difference| <= max(grad_frac x max|g_ref|, floor_mult x the tensor's own noise floor
    -- the worst difference over the reference's perturbed re-runs).  max_over tensors may exceed that (the L1 alignment term
    back-propagates sign(ema - source) per hooked channel: a channel within round-off of its source statistic flips with any change
    of summation order and moves its layer's d gamma / d beta by a +-momentum / C quantum, cf. test_host_cpu.check_tta_records), each
    by no more than cap x max|g_ref|; the cosine of the whole affine gradient >= cos_min.  Returns (cosine, worst in-bound error /
    bound, list of the tensors over their bound)."""
    k = f"{mode}_step{step}_"
    names = [str(n) for n in g[k + "affine_names"]]
    sizes = g[k + "affine_sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    ref_all = torch.from_numpy(g[k + "affine_grads"]).double()
    noise = g[k + "affine_noise"]
    got = []
    over, worst = [], 0.0
    for j, n in enumerate(names):
        ref = ref_all[offs[j]:offs[j + 1]]
        a = rec["affine"][n]
        if a is None:  # a parameter without gradient on our side must have a zero gradient in the reference as well
            assert float(ref.abs().max()) == 0.0, n
            got.append(torch.zeros_like(ref))
            continue
        a = a.double().flatten()
        got.append(a)
        assert torch.isfinite(a).all(), n
        gmax = float(ref.abs().max())
        bound = max(grad_frac * gmax, floor_mult * float(noise[j])) + 1e-12
        err = float((a - ref).abs().max())
        if err > bound:
            assert err <= cap * gmax, (n, err, bound, gmax, "beyond the outlier cap")
            over.append((n, err, bound))
        else:
            worst = max(worst, err / bound)
    va = torch.cat(got)
    cos = float(torch.dot(va, ref_all) / (va.norm() * ref_all.norm()))
    print(f"{mode}: {len(names)} LayerNorm affine tensors, {va.numel()} elements; cosine {cos:.8f}; worst in-bound error / bound {worst:.3f}; "
          f"over their bound: {over}")
    assert cos >= cos_min, cos
    assert len(over) <= max_over, over
    return cos, worst, over


@pytest.mark.parametrize("mode,use_engine", [("sgd", True), ("adam", True), ("sgd", False)])
def test_three_swin_tta_steps_match_reference(tmp_path, mode, use_engine):
    g = H.golden("tta3_swin.npz")
    recs = run_product_tta_swin(g, mode, tmp_path, torch.device("cpu"), OracleBackend, use_engine=use_engine)
    check_tta_records(g, mode, recs, BASE)


def test_config3_swin_full_size_cpu_oracle_path_matches_the_reference_itself(tmp_path):
    """Round 6: the CPU oracle path (what the GPU full-size tests were compared with in rounds 2-5) against the REFERENCE at BASELINE
    config 3's real size -- Video Swin-B, 2 views x 16 frames x 224^2, window (8, 7, 7) unclamped on the 14 x 14 / 7 x 7 planes of
    stages 2 / 3, where 20 of the 24 blocks and all 42 hooked LayerNorms live, shift mask in the backward
    (swin_transformer.py:138-169, 215-274, 316-329).  tests/golden/tta1_224_swin.npz = ONE step of the reference's own tta_standard
    (Adam on the LN affine parameters) with its DropPath / dropout masks, every LayerNorm affine gradient whole, and the noise floors of
    eight perturbed re-runs (tools/refgen/gen_golden.py tta224_swin)."""
    g = H.golden("tta1_224_swin.npz")
    recs = run_product_tta_swin(g, "adam", tmp_path, torch.device("cpu"), OracleBackend, all_affine=True)
    check_tta_records(g, "adam", recs, BASE, floor_mult=2.0)
    check_affine_gradients(g, "adam", recs[0], grad_frac=2e-3, floor_mult=2.0, max_over=0)
