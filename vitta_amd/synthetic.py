"""Seed-deterministic synthetic models / statistics (no dataset or checkpoint is reachable offline).

Used by bench.py, __graft_entry__.smoke() and the tests: random-init TANet of the real architecture
with BN running statistics calibrated so activations are O(1), and source statistics taken from a
second calibration pass.
"""
import os

import numpy as np
import torch
import torch.nn as nn


def seeded_randn(shape, seed, device=None):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(seed))
    return x.to(device) if device is not None else x


@torch.no_grad()
def calibrate_bn(model, x):
    """One pass in train mode with momentum 1: running stats := batch stats, so activations of the
    randomly initialised network are O(1) afterwards."""
    bns = [m for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    saved = [(m.momentum, m.training) for m in bns]
    was_training = model.training
    model.eval()
    for m in bns:
        m.momentum = 1.0
        m.train()
    model(x)
    for m, (mom, tr) in zip(bns, saved):
        m.momentum = mom
        m.train(tr)
    model.train(was_training)


@torch.no_grad()
def perturb_affine(model, seed, scale=0.1):
    """Seeded non-trivial norm affine parameters (default init is weight 1 / bias 0)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.LayerNorm)) and m.weight is not None:
            m.weight.add_(torch.randn(m.weight.shape, generator=g) * scale)
            m.bias.add_(torch.randn(m.bias.shape, generator=g) * scale)


def build_tanet(num_class, num_segments, seed, calib_size=64, calib_clips=8, var_floor=0.05):
    """Seeded TANet (this repo's TSN) with calibrated BN statistics and perturbed affine parameters.
    `var_floor` keeps 1/sqrt(running_var) bounded: with only calib_clips*T*h*w samples per channel a
    few calibrated variances come out ~1e-4 and would amplify gradients by 100x per layer."""
    from vitta_amd.tanet import TSN
    torch.manual_seed(seed)
    model = TSN(num_class, num_segments, "RGB", base_model="resnet50", consensus_type="avg", tam=True,
                partial_bn=False)  # get_model passes args.partial_bn (default False), basics.py:1473
    with torch.no_grad():
        model.new_fc.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(seed + 1))
    perturb_affine(model, seed + 2)
    x = seeded_randn((calib_clips, num_segments, 3, calib_size, calib_size), seed + 3)
    calibrate_bn(model, x)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                m.running_var.clamp_(min=var_floor)
    model.eval()
    return model


def write_stat_files(dirname, means, vars_, tag="golden"):
    """Two object-array .npy files like the ones compute_statistics writes."""
    mo = np.empty(len(means), dtype=object)
    vo = np.empty(len(vars_), dtype=object)
    for i, (m, v) in enumerate(zip(means, vars_)):
        mo[i], vo[i] = np.asarray(m, dtype=np.float32), np.asarray(v, dtype=np.float32)
    mp = os.path.join(dirname, f"list_spatiotemp_mean_{tag}.npy")
    vp = os.path.join(dirname, f"list_spatiotemp_var_{tag}.npy")
    np.save(mp, mo, allow_pickle=True)
    np.save(vp, vo, allow_pickle=True)
    return mp, vp




def build_swin(num_class, seed, patch_size=(2, 4, 4), window_size=(8, 7, 7), drop_path_rate=0.2, **kw):
    """Seeded Video Swin-B recognizer (this repo's Recognizer3D) with non-trivial LN affine / biases."""
    from vitta_amd.swin import Recognizer3D
    torch.manual_seed(seed)
    model = Recognizer3D(num_classes=num_class, patch_size=patch_size, window_size=window_size,
                         drop_path_rate=drop_path_rate, **kw)
    perturb_affine(model, seed + 2)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.add_(torch.randn(m.bias.shape, generator=g) * 0.02)
        model.cls_head.fc_cls.weight.normal_(0, 0.05, generator=g)
    model.eval()
    return model
