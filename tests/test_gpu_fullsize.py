"""The BENCHMARKED workloads at full size against the CPU oracle path: one adaptation step + the evaluation forward of
BASELINE config 2 (TANet-R50, 2 views x 8 frames x 224^2, 101 classes, Adam on the BN affine parameters) and config 3
(Video Swin-B, 2 views x 16 frames x 224^2, LN affine) -- HIP path (hand-written trunk / fused W-MSA + LayerNorm) vs
the same step on the CPU with the oracle backend (stock torch ops in the reference's op order, oracle/), identical
weights and clips, dropout / DropPath off.

Tolerances (fp32 both sides, different reduction orders):
  loss_reg rel 5e-5, loss_consis rel 1e-3; gradients of the norm-affine parameters: cosine of the whole gradient >= 0.999
  and per sampled tensor |d| <= 2e-2 max|g| (the L1 objective's sign(.) coefficients make single channels flip, so the
  direction of the whole gradient is the stronger statement); evaluation logits after the update |d| <= 2e-3 max|logit|,
  identical top-1.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

import helpers as H

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _compare(res, sampled, loss_reg_rel=5e-5):
    c, g = res["cpu"], res["cuda"]
    assert g["loss_reg"] == pytest.approx(c["loss_reg"], rel=loss_reg_rel), (g["loss_reg"], c["loss_reg"])
    assert g["loss_consis"] == pytest.approx(c["loss_consis"], rel=1e-3, abs=1e-6), (g["loss_consis"], c["loss_consis"])
    keys = sorted(c["grads"])
    assert keys == sorted(g["grads"]) and len(keys) > 0
    va = torch.cat([g["grads"][k].flatten() for k in keys]).double()
    vb = torch.cat([c["grads"][k].flatten() for k in keys]).double()
    assert torch.isfinite(va).all() and vb.norm() > 0
    cos = float(torch.dot(va, vb) / (va.norm() * vb.norm()))
    assert cos >= 0.999, cos
    for k in sampled:
        a, b = g["grads"][k], c["grads"][k]
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item() + 1e-9, (k, (a - b).abs().max().item(), b.abs().max().item())
    le, lc = g["logits"], c["logits"]
    assert (le - lc).abs().max().item() <= 2e-3 * lc.abs().max().item(), ((le - lc).abs().max().item(), lc.abs().max().item())
    assert int(le.argmax()) == int(lc.argmax())


def test_config2_tanet_full_size_step_matches_cpu_oracle(tmp_path):
    """TANet-R50 2 x 8 x 224^2, K = 101: the exact workload bench.py times."""
    from oracle import cpu_path
    from oracle.oracle_backend import OracleBackend
    from vitta_amd import data, trunk, tta
    T, size = 8, 224
    # source statistics: the moments of a seeded calibration clip on the 53 BatchNorm2d outputs (as compute_statistics
    # would write them), perturbed so that every channel has a non-trivial alignment term
    model0 = H.build_tanet(101, T, 0)
    bn2d = [m for m in model0.modules() if isinstance(m, nn.BatchNorm2d)]
    g = torch.Generator().manual_seed(11)
    means = [(torch.randn(b.num_features, generator=g) * 0.2).numpy() for b in bn2d]
    vars_ = [(torch.rand(b.num_features, generator=g) + 0.5).numpy() for b in bn2d]
    mp, vp = H.write_stat_files(str(tmp_path), means, vars_)
    args = H.tanet_args(tmp_path, clip_length=T, input_size=size, batch_size=1, spatiotemp_mean_clean_file=mp,
                        spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=5e-5)
    x = data.SyntheticVideoDataset(1, 2, T, size, 101, "tanet", seed0=21)[0][0].unsqueeze(0)
    xe = data.SyntheticVideoDataset(1, 1, T, size, 101, "tanet", seed0=21)[0][0].unsqueeze(0)
    res = {}
    for dev, backend in ((torch.device("cpu"), OracleBackend()), (_dev(), None)):
        model = H.build_tanet(101, T, 0)
        model.base_model.fc = nn.Identity()  # dropout off: both sides see the same forward
        if dev.type == "cpu":
            adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model), args, engine_backend=backend, use_engine=False)
            cpu_path.install_reference_order(adapter.model)
        else:
            adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(dev), args)
        adapter.set_adapt_mode()
        if dev.type == "cuda":  # the hand-written trunk must be the path under test
            assert trunk.TrunkRunner(adapter.model.module.base_model).eligible(adapter.shape_tta_input(x.to(dev)).view(-1, 3, size, size))
        _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x.to(dev)))
        grads = {k: v.grad.detach().cpu().clone() for k, v in adapter.model.named_parameters() if v.requires_grad and v.grad is not None}
        adapter.close_hooks()
        logits = adapter.evaluate(adapter.shape_eval_input(xe.to(dev))).detach().cpu().clone()
        res[dev.type] = dict(loss_reg=float(loss_reg), loss_consis=float(loss_consis), grads=grads, logits=logits)
    sampled = ["module.base_model.layer4.2.net.bn3.weight", "module.base_model.layer3.0.net.bn2.bias",
               "module.base_model.layer1.1.net.bn1.weight", "module.base_model.bn1.bias"]
    _compare(res, [k for k in sampled if k in res["cpu"]["grads"]])


def test_config3_swin_full_size_step_matches_cpu_oracle(tmp_path):
    """Video Swin-B 2 x 16 x 224^2, window (8, 7, 7), K = 101 (tta_swin_ucf101.py:27-40)."""
    from oracle.oracle_backend import OracleBackend
    from vitta_amd import data, scripts, tta
    from vitta_amd.bns_utils import choose_layers
    T, size, views = 16, 224, 2

    def build():
        m = H.build_swin(101, 0, drop_path_rate=0.0)
        m.cls_head.dropout = None
        return m

    lns = [m for _, m in choose_layers(build(), [nn.LayerNorm])][1:]
    g = torch.Generator().manual_seed(3)
    mp, vp = H.write_stat_files(str(tmp_path), [torch.randn(m.normalized_shape[0], generator=g).numpy() * 0.1 for m in lns],
                                [torch.rand(m.normalized_shape[0], generator=g).numpy() + 0.5 for m in lns])
    args = scripts.swin_ucf101_args([])
    args.datatype, args.num_classes = "synthetic", 101
    args.n_augmented_views = views
    args.input_size, args.scale_size, args.workers, args.verbose, args.result_dir = size, size, 0, False, str(tmp_path)
    args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
    args.update_only_bn_affine, args.lr = True, 1e-5
    x = data.SyntheticVideoDataset(1, views, T, size, 101, "swin", seed0=40)[0][0].unsqueeze(0)
    xe = data.SyntheticVideoDataset(1, 1, T, size, 101, "swin", seed0=40)[0][0].unsqueeze(0)
    res = {}
    for dev, backend in ((torch.device("cpu"), OracleBackend()), (_dev(), None)):
        adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(build()).to(dev), args, engine_backend=backend)
        adapter.set_adapt_mode()
        _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x.to(dev)))
        grads = {k: v.grad.detach().cpu().clone() for k, v in adapter.model.named_parameters() if v.requires_grad and v.grad is not None}
        adapter.close_hooks()
        logits = adapter.evaluate(adapter.shape_eval_input(xe.to(dev))).detach().cpu().clone()
        res[dev.type] = dict(loss_reg=float(loss_reg), loss_consis=float(loss_consis), grads=grads, logits=logits)
    sampled = ["module.backbone.layers.2.blocks.4.norm1.weight", "module.backbone.norm.bias",
               "module.backbone.layers.3.blocks.1.norm2.weight", "module.backbone.layers.2.blocks.17.norm2.bias"]
    _compare(res, [k for k in sampled if k in res["cpu"]["grads"]], loss_reg_rel=1e-4)
