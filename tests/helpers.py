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


from vitta_amd.synthetic import build_swin, build_tanet, calibrate_bn, perturb_affine, write_stat_files  # noqa: E402,F401


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


class MaskTape:
    """Recorded per-sample keep flags of every stochastic-depth call of a run, in call order."""

    def __init__(self, masks):
        self.masks, self.pos = masks, 0

    def next(self):
        m = self.masks[self.pos]
        self.pos += 1
        return m


class ReplayDropPath(nn.Module):
    """DropPath replaying recorded keep flags (timm semantics: kept samples are scaled by 1/keep_prob)."""

    def __init__(self, drop_prob, tape):
        super().__init__()
        self.drop_prob, self.tape = drop_prob, tape

    def forward(self, x):
        if not self.training or self.drop_prob == 0.0:
            return x
        keep = torch.as_tensor(self.tape.next(), dtype=x.dtype, device=x.device)
        return x * (keep / (1.0 - self.drop_prob)).view((-1,) + (1,) * (x.dim() - 1))


def pack_mask(mask):
    m = np.asarray(mask, dtype=bool)
    return np.packbits(m.reshape(-1)), np.array(m.shape)


def unpack_mask(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(tuple(int(s) for s in shape)).astype(np.float32))


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


class FakeDecord:
    """Stand-in for the `decord` module in dataset tests (decord is not installed in the image): every video is a
    seeded stack of uint8 frames; `VideoReader(path)` / `len` / `get_batch(idx).asnumpy()` as the datasets use them."""

    def __init__(self, n_frames=40, w=320, h=240, seed=0):
        self.n_frames, self.w, self.h, self.seed = n_frames, w, h, seed
        self.opened = []

    def VideoReader(self, path):
        import zlib
        import numpy as np
        owner = self
        owner.opened.append(path)
        rng = np.random.RandomState((self.seed + zlib.crc32(path.encode())) % (2 ** 31))
        base = rng.randint(0, 256, size=(self.n_frames, self.h // 16 + 1, self.w // 16 + 1, 3)).astype(np.uint8)
        frames = np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)[:, :self.h, :self.w]
        frames = (frames.astype(np.int32) + rng.randint(-8, 9, size=frames.shape)).clip(0, 255).astype(np.uint8)

        class Batch:
            def __init__(self, arr):
                self.arr = arr

            def asnumpy(self):
                return self.arr

        class Reader:
            def __len__(self):
                return owner.n_frames

            def get_batch(self, idx):
                return Batch(frames[np.asarray(idx)])

        return Reader()
