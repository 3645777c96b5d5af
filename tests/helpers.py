"""Shared, seed-deterministic builders used by the tests, the golden generator and bench.py.

Nothing here touches /root/reference; everything is reproducible from seeds (same torch build on the
GPU box and in the build container), so fixtures only need to store OUTPUTS of the reference.
"""
import os

import numpy as np
import torch
import torch.nn as nn

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=True)


def seeded_randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def channel_feature(shape, seed, chan_dim, offset_scale=3.0):
    """Noise with a per-channel offset and scale (non-zero means, unequal variances)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    c = shape[chan_dim]
    off = torch.randn(c, generator=g) * offset_scale
    sc = torch.rand(c, generator=g) * 2 + 0.25
    view = [1] * len(shape)
    view[chan_dim] = c
    return x * sc.view(view) + off.view(view)


def feature_module(kind, c):
    """The norm module a statistics hook is attached to in the op-level fixtures: default-initialised,
    eval mode.  The hooked feature is `feature_module(kind, c)(x)` (BN: x / sqrt(1 + eps); LN: the
    per-token normalisation), recomputed identically wherever the fixture is consumed."""
    if kind == "ln":
        return nn.LayerNorm(c).eval()
    return {"bn2d": nn.BatchNorm2d, "bn3d": nn.BatchNorm3d, "bn1d": nn.BatchNorm1d}[kind](c).eval()


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


class ReplayDropout(nn.Module):
    """Dropout that replays recorded keep-masks (one per training-mode call), scaled by 1/(1-p)."""

    def __init__(self, p, masks):
        super().__init__()
        self.p, self.masks, self.calls = p, masks, 0

    def forward(self, x):
        if not self.training:
            return x
        m = self.masks[self.calls].to(x.device, x.dtype).view_as(x)
        self.calls += 1
        return x * m / (1.0 - self.p)


def pack_mask(mask):
    m = np.asarray(mask, dtype=bool)
    return np.packbits(m.reshape(-1)), np.array(m.shape)


def unpack_mask(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(tuple(int(s) for s in shape)).astype(np.float32))


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


def tanet_args(tmpdir, num_classes_dataset="ucf101", **over):
    """get_opts() defaults + the overrides of a small synthetic TANet run."""
    from vitta_amd.opts import get_opts
    args = get_opts([])
    args.arch, args.dataset, args.datatype = "tanet", num_classes_dataset, "synthetic"
    args.clip_length, args.workers, args.verbose = 8, 0, False
    args.result_dir = str(tmpdir)
    args.gpus = [0]
    args.num_classes = 101
    for k, v in over.items():
        setattr(args, k, v)
    return args
