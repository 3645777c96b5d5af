"""Oracle-backed stand-in for vitta_amd.norm_stats.HipBackend -- TEST INFRASTRUCTURE ONLY.

Lets the CPU test-suite drive the product's host logic (hook protocol, engine bookkeeping, the
tta_standard loop, the data-parallel exchanges over gloo) without a GPU, by computing what the HIP
launches would compute with the CPU oracle.  Never imported by vitta_amd/ (only tests/, smoke() and bench.py's cpu_baseline leg).
"""
import torch

from . import vitta_oracle as O
from vitta_amd._lib import LAYOUT_NCHW, LAYOUT_NHWC


def _kind_of(feature, layout):
    if layout == LAYOUT_NHWC:
        return "ln"
    return "bn2d" if feature.dim() == 4 else "bn3d"


def _layout(feature, kind):
    if kind in ("rows", "nct"):
        return _bns_kind_layout(feature, kind)
    if kind == "bn2d":
        nt, c, h, w = feature.shape
        return nt, c, h * w, LAYOUT_NCHW
    if kind == "bn3d":
        n, c, t, h, w = feature.shape
        return n, c, t * h * w, LAYOUT_NCHW
    c = feature.shape[-1]
    return feature.numel() // c, c, 1, LAYOUT_NHWC


def _moments(feature, kind):
    clip = feature.shape[0] if kind == "bn2d" else None
    return O.moments(feature, kind, clip)


def _bns_kind_layout(feature, kind):
    if kind == "rows":
        return feature.shape[0], feature.shape[1], 1, LAYOUT_NHWC
    n, c, t = feature.shape
    return n, c, t, LAYOUT_NCHW


class OraclePlan:
    def __init__(self, shapes, device):
        self.shapes = [tuple(int(v) for v in s) for s in shapes]
        self.device = device
        n = self.n_layers = len(self.shapes)
        self.offsets, off = [], 0
        for _, c, _, _ in self.shapes:
            self.offsets.append(off)
            off += c
        tc = self.total_channels = off
        self.stats = torch.zeros(n + 2 * tc, device=device)
        self.cnt, self.s1, self.s2 = self.stats[:n], self.stats[n:n + tc], self.stats[n + tc:]
        self.mu, self.coef_a, self.coef_b = (torch.zeros(tc, device=device) for _ in range(3))
        self.layer_loss = torch.zeros(n, device=device)
        self.total_loss = torch.zeros(1, device=device)

    def channel_slice(self, layer):
        o = self.offsets[layer]
        return slice(o, o + self.shapes[layer][1])

    def moments(self, feats, shift=None):
        for i, f in enumerate(feats):
            outer, c, inner, layout = self.shapes[i]
            mean, var = _moments(f.double(), _kind_of(f, layout))
            n = outer * inner
            sl = self.channel_slice(i)
            k = shift[sl].double() if shift is not None else torch.zeros(c, dtype=torch.float64)
            self.cnt[i] = n
            self.s1[sl] = (n * (mean - k)).float()
            self.s2[sl] = (n * (var + (mean - k) ** 2)).float()
        return self.cnt, self.s1, self.s2

    def align(self, shift, ema_mean, ema_var, src_mean, src_var, momentum, reg_type):
        total = torch.zeros((), dtype=torch.float32)
        for i in range(self.n_layers):
            sl = self.channel_slice(i)
            n = self.cnt[i].double()
            k = shift[sl].double() if shift is not None else 0.0
            m1, m2 = self.s1[sl].double() / n, self.s2[sl].double() / n
            mean = (k + m1).float()
            var = (m2 - m1 * m1).clamp_min(0).float()
            ema_mean[sl] = momentum * mean + (1.0 - momentum) * ema_mean[sl]
            ema_var[sl] = momentum * var + (1.0 - momentum) * ema_var[sl]
            loss = O.compute_regularization(src_mean[sl], ema_mean[sl], src_var[sl], ema_var[sl], reg_type)
            a, b = O.align_coefficients(ema_mean[sl], ema_var[sl], src_mean[sl], src_var[sl], momentum, reg_type, float(n))
            self.mu[sl], self.coef_a[sl], self.coef_b[sl] = mean, a, b
            self.layer_loss[i] = loss
            total = total + loss
        self.total_loss[0] = total
        return self.total_loss, self.layer_loss


class OracleBackend:
    def moments(self, feature, kind):
        return _moments(feature, kind)

    def feature_moments(self, feature, kind):
        return _moments(feature, kind)  # plain torch ops: autograd supplies the backward

    def make_plan(self, shapes, device):
        return OraclePlan(shapes, device)

    def layout(self, feature, kind):
        return _layout(feature, kind)

    def inject(self, x, gout, kind, mu, a, b, gscale):
        view = [1] * x.dim()
        cdim = x.dim() - 1 if kind == "ln" else 1
        view[cdim] = -1
        return gout + gscale * (a.view(view) + b.view(view) * (x - mu.view(view)))
