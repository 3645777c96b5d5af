"""The hand-written trunk (vitta_amd/trunk.py: channel-major planes, vitta_conv_f32 everywhere, BN / ReLU / residual /
moments in the convolution epilogues) against the module-by-module path of the same model, and against the CPU oracle
path at the benchmarked size."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn

import helpers as H

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _adapter(tmp_path, size, fast, mode="adam", **over):
    from vitta_amd import trunk, tta
    g = H.golden("tta3.npz")
    cfg = json.loads(str(g["config"]))
    T = cfg["T"]
    model = H.build_tanet(101, T, 0)
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    means = [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    vars_ = [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    mp, vp = H.write_stat_files(str(tmp_path), means, vars_)
    args = H.tanet_args(tmp_path, clip_length=T, input_size=size, batch_size=1, spatiotemp_mean_clean_file=mp,
                        spatiotemp_var_clean_file=vp, update_only_bn_affine=(mode == "adam"),
                        lr=cfg["lr_adam"] if mode == "adam" else cfg["lr_sgd"], **over)
    trunk.ENABLED = fast
    adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(_dev()), args)
    adapter.model.module.base_model.fc = nn.Identity()  # no dropout: both paths see the same forward
    return adapter, T


@pytest.mark.parametrize("size", [64, 112])
def test_eval_forward_equals_module_path(tmp_path, size):
    from vitta_amd import trunk
    try:
        adapter, T = _adapter(tmp_path, size, True)
        x = H.seeded_randn((1, T * 3, size, size), 5).to(_dev())
        adapter.close_hooks()
        trunk.ENABLED = True
        fast = adapter.evaluate(adapter.shape_eval_input(x)).clone()
        trunk.ENABLED = False
        ref = adapter.evaluate(adapter.shape_eval_input(x)).clone()
    finally:
        trunk.ENABLED = True
    assert (fast - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert int(fast.argmax()) == int(ref.argmax())


@pytest.mark.parametrize("size,over", [(64, {}), (112, {}), (64, dict(if_pred_consistency=False)), (64, dict(reg_type="mse_loss")),
                                       (112, dict(reg_type="mse_loss")), (112, dict(reg_type="mse_loss", before_norm=True)),
                                       (64, dict(before_norm=True))])
def test_adapt_step_equals_module_path(tmp_path, size, over, abi_calls):
    """One adaptation step (statistics alignment on the hooked layers + consistency, Adam on the BN affine parameters):
    losses, every affine gradient and the evaluation logits after the update, hand-written trunk vs module path."""
    from vitta_amd import trunk
    res = {}
    try:
        # "p1" / "p2": the MODULE path again on an input perturbed by one part in 10^6 -- the yardstick for the
        # post-update logits (Adam's first update is lr * sign(g): a gradient within round-off of zero flips a whole step)
        for fast in (True, False, "p1", "p2"):
            (tmp_path / str(fast)).mkdir()
            adapter, T = _adapter(tmp_path / str(fast), size, fast is True, **over)
            x = H.seeded_randn((1, 2 * T * 3, size, size), 7)
            if isinstance(fast, str):
                x = x * (1.0 + 1e-6 * torch.sign(H.seeded_randn(tuple(x.shape), 70 + int(fast[1]))))
            x = x.to(_dev())
            adapter.set_adapt_mode()
            calls0 = dict(abi_calls.abi)
            _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
            if fast is True:  # the node ran (before_norm hooks included: raw-output statistics, raw injection)
                assert abi_calls.abi.get("vitta_conv_f32", 0) > calls0.get("vitta_conv_f32", 0)
                assert abi_calls.abi.get("vitta_bn_bwd_cm_f32", 0) > calls0.get("vitta_bn_bwd_cm_f32", 0)
            grads = {k: v.grad.detach().clone() for k, v in adapter.model.named_parameters() if v.requires_grad}
            adapter.close_hooks()
            ev = adapter.evaluate(adapter.shape_eval_input(H.seeded_randn((1, T * 3, size, size), 8).to(_dev()))).clone()
            res[fast] = (float(loss_reg), None if loss_consis is None else float(loss_consis), grads, ev)
    finally:
        trunk.ENABLED = True
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) <= 2e-5 * abs(b[0]) + 1e-7, (a[0], b[0])
    if b[1] is not None:
        assert abs(a[1] - b[1]) <= 1e-4 * abs(b[1]) + 1e-6, (a[1], b[1])
    # The L1 objective is discontinuous by construction (sign(.) coefficients: a channel whose EMA sits within round-off
    # of its source statistic flips its whole contribution; tools/debug/trunk_ab.py shows exactly one flipped coefficient
    # behind the largest deviation), so the element-wise comparison runs on the smooth objective (mse_loss) and the L1
    # runs are held to the direction of the whole gradient.
    smooth = over.get("reg_type") == "mse_loss"
    va = torch.cat([a[2][k].flatten() for k in b[2]])
    vb = torch.cat([b[2][k].flatten() for k in b[2]])
    cos = float(torch.dot(va, vb) / (va.norm() * vb.norm()))
    assert cos >= 0.9995, cos
    if smooth:
        for k, gb in b[2].items():  # (sums of many cancelling terms, e.g. the 16 BatchNorm1d weights of a TAM: L2, not max)
            fl = max((res[q][2][k] - gb).norm().item() for q in ("p1", "p2"))  # the module path under a 1e-6 input change
            assert (a[2][k] - gb).norm().item() <= max(2e-2 * gb.norm().item(), 4.0 * fl) + 1e-9, \
                (k, (a[2][k] - gb).norm().item(), gb.norm().item(), fl)
    floor = max((res[k][3] - b[3]).abs().max().item() for k in ("p1", "p2"))
    assert (a[3] - b[3]).abs().max().item() <= max(2e-3 * b[3].abs().max().item(), 4.0 * floor), (floor, b[3].abs().max().item())


@pytest.mark.parametrize("size", [112, 224])
def test_pooling_from_the_convolution_epilogue_equals_the_pooling_launch(tmp_path, size, abi_calls):
    """trunk.POOL_FOLD: TAM's pooled means come out of conv1's accumulators (VITTA_CONV_POOL, atomics-ordered sums, frame-major)
    instead of a vitta_tam_pool_cm_f32 launch per block: same evaluation logits and the same adaptation step to summation-order
    accuracy (mse alignment: no sign() discontinuity), and the stand-alone launch is gone wherever a plane has >= 32 pixels."""
    from vitta_amd import trunk
    res = {}
    old = trunk.POOL_FOLD
    try:
        for fold in (True, False):
            (tmp_path / str(fold)).mkdir()
            trunk.POOL_FOLD = fold
            adapter, T = _adapter(tmp_path / str(fold), size, True, reg_type="mse_loss")
            x = H.seeded_randn((1, 2 * T * 3, size, size), 7).to(_dev())
            adapter.set_adapt_mode()
            n0 = abi_calls.abi.get("vitta_tam_pool_cm_f32", 0)
            _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
            pools = abi_calls.abi.get("vitta_tam_pool_cm_f32", 0) - n0
            grads = {k: v.grad.detach().clone() for k, v in adapter.model.named_parameters() if v.requires_grad}
            adapter.close_hooks()
            ev = adapter.evaluate(adapter.shape_eval_input(H.seeded_randn((1, T * 3, size, size), 8).to(_dev()))).clone()
            res[fold] = (float(loss_reg), float(loss_consis), grads, ev, pools)
    finally:
        trunk.POOL_FOLD = old
    a, b = res[True], res[False]
    small = {112: 2, 224: 0}[size]  # blocks whose input planes have fewer than 32 pixels keep the launch (112^2 input: layer4.1 / 4.2 see 4 x 4)
    assert b[4] == 16 and a[4] == small, (a[4], b[4])
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-4 * abs(b[1]) + 1e-7
    for k, gb in b[2].items():
        # (summation-order noise of the pooled means, amplified through up to sixteen blocks of ReLU masks on the way back to the stem:
        # 2e-3 of the norm there, 1e-5 at the last stage; test_adapt_step_equals_module_path uses 2e-2 for the same reason)
        assert (a[2][k] - gb).norm().item() <= 1e-2 * gb.norm().item() + 1e-9, (k, (a[2][k] - gb).norm().item(), gb.norm().item())
    # (logits AFTER the update: Adam's first step is lr * sign(g), a gradient within round-off of zero moves its parameter by 2 lr)
    assert (a[3] - b[3]).abs().max().item() <= 2e-3 * b[3].abs().max().item()


def test_sgd_all_step_equals_module_path(tmp_path):
    """The reference's default optimizer (SGD over ALL parameters, corpus/basics.py:547-560) on the hand-written trunk:
    convolution weight gradients from vitta_conv_wgrad_f32, TAM / head weights from their own kernels, the stem as torch
    modules in front of the node -- against the module path (library convolutions + autograd): losses, the direction of
    the whole gradient, sampled convolution-weight gradients, and the evaluation logits after the update."""
    from vitta_amd import trunk
    res = {}
    try:
        for fast in (True, False):
            (tmp_path / str(fast)).mkdir()
            adapter, T = _adapter(tmp_path / str(fast), 64, fast, mode="sgd", reg_type="mse_loss")
            x = H.seeded_randn((1, 2 * T * 3, 64, 64), 7).to(_dev())
            adapter.set_adapt_mode()
            if fast:
                xin = adapter.shape_tta_input(x).view(-1, 3, 64, 64)
                assert trunk.TrunkRunner(adapter.model.module.base_model).eligible(xin)
            _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
            grads = {k: v.grad.detach().clone() for k, v in adapter.model.named_parameters() if v.requires_grad and v.grad is not None}
            adapter.close_hooks()
            ev = adapter.evaluate(adapter.shape_eval_input(H.seeded_randn((1, T * 3, 64, 64), 8).to(_dev()))).clone()
            res[fast] = (float(loss_reg), float(loss_consis), grads, ev)
    finally:
        trunk.ENABLED = True
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) <= 2e-5 * abs(b[0]) + 1e-7 and abs(a[1] - b[1]) <= 1e-4 * abs(b[1]) + 1e-6
    assert sorted(a[2]) == sorted(b[2])
    va = torch.cat([a[2][k].flatten() for k in b[2]])
    vb = torch.cat([b[2][k].flatten() for k in b[2]])
    assert float(torch.dot(va, vb) / (va.norm() * vb.norm())) >= 0.9995
    for k in ("module.base_model.layer1.0.net.conv2.weight", "module.base_model.layer2.0.net.downsample.0.weight",
              "module.base_model.layer3.2.net.conv3.weight", "module.base_model.layer4.1.net.conv1.weight",
              "module.base_model.conv1.weight", "module.base_model.layer2.1.tam.L.0.weight", "module.new_fc.weight"):
        assert (a[2][k] - b[2][k]).norm().item() <= 2e-2 * b[2][k].norm().item() + 1e-9, k
    assert (a[3] - b[3]).abs().max().item() <= 2e-3 * b[3].abs().max().item()


@pytest.mark.parametrize("before_norm,dead_gamma", [(False, False), (True, False), (True, True), (False, True)])
def test_source_statistics_producer_runs_on_the_trunk_node(before_norm, dead_gamma, abi_calls):
    """compute_statistics' hooks (ComputeNormStatsHook on all 53 BatchNorm2d, corpus/basics.py:220-307,
    utils/norm_stats_utils.py:18-101): the node supplies every hook's batch moments from the convolution epilogues (the stem's
    from its raw output), of the BN output or -- before_norm -- of its INPUT, summed directly from the raw convolution output
    (VITTA_CONV_STATS_RAW); same numbers as the module-by-module path (library convolutions + the stand-alone moments kernel on
    every materialised feature).  dead_gamma: trained networks have BatchNorm channels with gamma = 0 or ~ 1e-8; the input
    moments of such a channel are ordinary numbers (round 3 recovered them by dividing the output moments by gamma)."""
    from vitta_amd import trunk
    from vitta_amd.norm_stats import ComputeNormStatsHook
    dev = torch.device("cuda:0")
    model = H.build_tanet(11, 8, 0).to(dev).eval()
    if dead_gamma:
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight[::7] = 0.0
                    m.weight[1::7] = 1e-8
    x = H.seeded_randn((2, 8, 3, 64, 64), 5).to(dev)
    bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    res = {}
    for on in (True, False):
        hooks = [ComputeNormStatsHook(m, clip_len=8, stat_type="spatiotemp", before_norm=before_norm, batch_size=2) for m in bn2d]
        old, trunk.ENABLED = trunk.ENABLED, on
        try:
            if on:
                assert trunk.TrunkRunner(model.base_model).eligible(x.view(-1, 3, 64, 64))
            with torch.no_grad():
                model(x)
        finally:
            trunk.ENABLED = old
        res[on] = ([h.batch_mean.cpu().double() for h in hooks], [h.batch_var.cpu().double() for h in hooks])
        for h in hooks:
            h.close()
        if on:
            abi_calls.assert_tanet_trunk()
    assert len(res[True][0]) == 53
    for a, b in zip(res[True][0], res[False][0]):
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-6
    for a, b in zip(res[True][1], res[False][1]):
        assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item() + 1e-7


@pytest.mark.parametrize("b,v,t,k,d,p", [(1, 2, 8, 101, 2048, 0.8), (2, 2, 8, 400, 2048, 0.5), (1, 2, 16, 174, 2048, 0.0), (3, 1, 4, 11, 512, 0.3),
                                         (2, 4, 8, 101, 2048, 0.8)])
def test_fused_head_equals_the_module_chain_in_fp64(b, v, t, k, d, p):
    """ops.TanetHead (dropout -> new_fc -> segment consensus -> compute_pred_consis -> mean over the views: ONE launch behind ATen's
    dropout forward, ONE backward) against the reference's chain in fp64 on the same dropout mask (same seed, same draw):
    video logits, consistency loss, and the gradients w.r.t. the pooled features, the head's weight and bias -- with upstream
    gradients on BOTH outputs.  models/tanet_models/tanet.py:243-251, utils/pred_consistency_utils.py:15-31."""
    from oracle import vitta_oracle as O
    from vitta_amd import ops
    dev = _dev()
    feat = (H.seeded_randn((b * v * t, d), 3) * 0.5 + 0.2).to(dev).requires_grad_(True)
    w = (H.seeded_randn((k, d), 4) * 0.05).to(dev).requires_grad_(True)
    bias = (H.seeded_randn((k,), 5) * 0.1).to(dev).requires_grad_(True)
    gout = H.seeded_randn((b, k), 6).to(dev)
    gl = 0.7
    torch.manual_seed(1234)
    out, loss = ops.TanetHead.apply(feat, w, bias, p, True, t, v)
    (gl * loss + (out * gout).sum()).backward()
    got = [out.detach().cpu().double(), float(loss.detach()), feat.grad.cpu().double(), w.grad.cpu().double(), bias.grad.cpu().double()]
    torch.manual_seed(1234)
    if p > 0:
        _, mask = torch.native_dropout(feat.detach(), p, True)
    else:
        mask = torch.ones_like(feat, dtype=torch.bool)
    f64 = feat.detach().cpu().double().requires_grad_(True)
    w64, b64 = w.detach().cpu().double().requires_grad_(True), bias.detach().cpu().double().requires_grad_(True)
    y = f64 * mask.cpu().double() / (1.0 - p)
    lv = (y @ w64.t() + b64).view(b * v, t, k).mean(1).view(b, v, k)
    loss_ref = O.compute_pred_consis(lv)
    out_ref = lv.mean(1)
    (gl * loss_ref + (out_ref * gout.cpu().double()).sum()).backward()
    ref = [out_ref.detach(), float(loss_ref.detach()), f64.grad, w64.grad, b64.grad]
    assert (got[0] - ref[0]).abs().max().item() <= 1e-5 * ref[0].abs().max().item() + 1e-6
    assert abs(got[1] - ref[1]) <= 1e-5 * abs(ref[1]) + 1e-7
    for a, r, name in zip(got[2:], ref[2:], ("d feat", "d weight", "d bias")):
        assert (a - r).abs().max().item() <= 2e-5 * r.abs().max().item() + 1e-8, (name, (a - r).abs().max().item(), r.abs().max().item())
    # a loss-only backward (what the adaptation step does: the video logits are only logged) and the frozen head of affine mode
    feat2 = feat.detach().clone().requires_grad_(True)
    torch.manual_seed(1234)
    out2, loss2 = ops.TanetHead.apply(feat2, w.detach(), bias.detach(), p, True, t, v)
    loss2.backward()
    f3 = feat.detach().cpu().double().requires_grad_(True)
    y3 = f3 * mask.cpu().double() / (1.0 - p)
    O.compute_pred_consis((y3 @ w64.detach().t() + b64.detach()).view(b * v, t, k).mean(1).view(b, v, k)).backward()
    assert (feat2.grad.cpu().double() - f3.grad).abs().max().item() <= 2e-5 * f3.grad.abs().max().item() + 1e-9


@pytest.mark.parametrize("mode,over", [("adam", {}), ("sgd", {}), ("adam", dict(if_pred_consistency=False))])
def test_adapt_step_with_the_fused_head_equals_the_module_chain(tmp_path, mode, over, abi_calls, monkeypatch):
    """One adaptation step with the stock head (nn.Dropout, p = 0 so both arms see the same forward) through ops.TanetHead /
    ops.WeightedLoss against the module chain (VITTA_FUSED_HEAD=0: Linear over the frames, consensus, PredConsis, ATen's loss
    arithmetic): losses and every gradient."""
    from vitta_amd import tta
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(tta, "FUSED_HEAD", fused)
        (tmp_path / str(fused)).mkdir()
        adapter, T = _adapter(tmp_path / str(fused), 64, True, mode=mode, **over)
        adapter.model.module.base_model.fc = nn.Dropout(p=0.0)
        x = H.seeded_randn((1, 2 * T * 3, 64, 64), 7).to(_dev())
        adapter.set_adapt_mode()
        calls0 = dict(abi_calls.abi)
        out, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
        ran = abi_calls.abi.get("vitta_tanet_head_fwd_f32", 0) > calls0.get("vitta_tanet_head_fwd_f32", 0)
        assert ran == fused
        grads = {k: v.grad.detach().clone() for k, v in adapter.model.named_parameters() if v.requires_grad and v.grad is not None}
        res[fused] = (out.clone(), float(loss_reg), None if loss_consis is None else float(loss_consis), grads)
    a, b = res[True], res[False]
    assert (a[0] - b[0]).abs().max().item() <= 1e-5 * b[0].abs().max().item() + 1e-6
    assert abs(a[1] - b[1]) <= 1e-6 * abs(b[1]) + 1e-8
    if b[2] is not None:
        assert abs(a[2] - b[2]) <= 1e-5 * abs(b[2]) + 1e-8
    assert set(a[3]) == set(b[3])
    va = torch.cat([a[3][k].flatten() for k in b[3]])
    vb = torch.cat([b[3][k].flatten() for k in b[3]])
    assert float(torch.dot(va, vb) / (va.norm() * vb.norm())) >= 0.99999
    for k, gb in b[3].items():
        assert (a[3][k] - gb).norm().item() <= 1e-3 * gb.norm().item() + 1e-7 * max(1.0, float(gb.numel()) ** 0.5), k


def test_reference_order_forward_zero_grad_backward_keeps_the_forward_state(tmp_path, abi_calls):
    """corpus/basics.py:669-671 issues `optimizer.zero_grad()` BETWEEN the forward and `loss.backward()`.  The arena's tail holds
    forward state of the running step (the trunk's fixed-point pooled sums that the TAM backward reads, the engine's [cnt | s1 | s2]):
    zero_grad() through the public optimizer surface must clear gradients only -- the gradients of that order equal the
    adapter's own step, and the two fills are distinct calls (FlatArena.zero_step is the step's)."""
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    a1, T = _adapter(tmp_path / "a", 64, True)
    a2, _ = _adapter(tmp_path / "b", 64, True)
    x = H.seeded_randn((1, 2 * T * 3, 64, 64), 7).to(_dev())
    res = []
    for k, ad in enumerate((a1, a2)):
        ad.set_adapt_mode()
        inp = ad.shape_tta_input(x)
        ad.arena.zero_step()
        ad.arena.grad.fill_(123.0)  # stale gradients: zero_grad() has to remove them in the reference order
        if k == 0:
            ad.arena.zero_grad()
        tail = lambda: ad.arena._grad_all[ad.arena.grad.numel():].view(torch.int32).clone()  # (bit patterns: fixed-point words live there)
        tail_before = tail()
        _, loss_reg, loss_consis = ad.forward_losses(inp, 1)
        tail_fwd = tail()
        assert (tail_fwd != tail_before).any()  # the forward left state there (pooled sums / statistics)
        if k == 1:
            ad.optimizer.zero_grad()  # the reference's position
            assert torch.equal(tail(), tail_fwd)
        ad.arena.before_backward()
        ad._backward(ad.total_loss(loss_reg, loss_consis))
        ad.arena.after_backward()
        res.append((float(loss_reg), float(loss_consis), ad.arena.grad.detach().clone()))
    abi_calls.assert_tanet_trunk()
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    assert torch.isfinite(res[1][2]).all() and res[1][2].abs().max() > 0 and res[1][2].abs().max() < 100.0
    # (two runs of the same kernels: the d gamma / d beta atomics arrive in another order)
    torch.testing.assert_close(res[1][2], res[0][2], rtol=1e-4, atol=1e-5 * float(res[0][2].abs().max()) + 1e-12)
