"""CPU ORACLE -- test infrastructure, NOT product code.

A restatement, in stock PyTorch CPU ops and in the reference's own op order, of the ViTTA
online-adaptation operators.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (vitta_amd/) never does and fails loudly when libvitta_hip.so is missing.

Pinning: every function below is checked in tests/test_oracle_golden.py against golden
vectors captured by importing the reference implementation (/root/reference) in the build
container with tools/refgen/gen_golden.py (fixtures under tests/golden/).  Citations are
file:line of the reference checkout.
"""
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# A1 / A2 -- moments of a hooked feature
# ---------------------------------------------------------------------------------------------
def to_ncthw(feature, kind, clip_len=None):
    """Bring a hooked feature to (N, C, T, H, W) the way hook_fn does.

    bn2d: utils/norm_stats_utils.py:189-193  view(bz*m, t, c, h, w).permute(0,2,1,3,4).contiguous()
    bn3d: utils/norm_stats_utils.py:195-199  already (N, C, T, H, W)
    ln  : utils/norm_stats_utils.py:224-227  permute(0,4,1,2,3).contiguous()
    """
    if kind == "bn2d":
        nmt, c, h, w = feature.shape
        return feature.view(nmt // clip_len, clip_len, c, h, w).permute(0, 2, 1, 3, 4).contiguous()
    if kind == "bn3d":
        return feature
    if kind == "ln":
        return feature.permute(0, 4, 1, 2, 3).contiguous()
    raise ValueError(kind)


def moments_ncthw(output):
    """utils/norm_stats_utils.py:242-243 (== :93-95 of ComputeNormStatsHook)."""
    c = output.shape[1]
    batch_mean = output.mean((0, 2, 3, 4))
    batch_var = output.permute(1, 0, 2, 3, 4).contiguous().view([c, -1]).var(1, unbiased=False)
    return batch_mean, batch_var


def moments(feature, kind, clip_len=None):
    if kind == "rows":  # utils/BNS_utils.py:43-45, BatchNorm1d input (N*C, T)
        nch = feature.shape[1]
        return feature.mean([0]), feature.permute(1, 0).contiguous().view([nch, -1]).var(1, unbiased=False)
    if kind == "nct":  # utils/BNS_utils.py:46-48, BatchNorm1d input (N, C, T)
        nch = feature.shape[1]
        return feature.mean([0, 2]), feature.permute(1, 0, 2).contiguous().view([nch, -1]).var(1, unbiased=False)
    return moments_ncthw(to_ncthw(feature, kind, clip_len))


# ---------------------------------------------------------------------------------------------
# A3 -- zero-initialised EMA without bias correction
# ---------------------------------------------------------------------------------------------
class MovingAverage:
    """utils/utils_.py:204-211: avg0 = scalar 0; avg <- m*val + (1-m)*avg.detach()."""

    def __init__(self, momentum=0.1):
        self.momentum = momentum
        self.avg = torch.tensor(0.0)

    def update(self, val):
        self.avg = self.momentum * val + (1.0 - self.momentum) * self.avg.detach()
        return self.avg


# ---------------------------------------------------------------------------------------------
# A4 -- alignment loss
# ---------------------------------------------------------------------------------------------
def compute_kld(mean_true, mean_pred, var_true, var_pred):
    """utils/norm_stats_utils.py:8-16."""
    kld = 0.5 * torch.log(torch.div(var_pred, var_true)) + (var_true + (mean_true - mean_pred) ** 2) / (2 * var_pred) - 0.5
    return torch.sum(kld)


def compute_regularization(mean_true, mean_pred, var_true, var_pred, reg_type):
    """utils/norm_stats_utils.py:531-542 (L1Loss / MSELoss with reduction='mean', :5-6)."""
    if reg_type == "mse_loss":
        return F.mse_loss(var_true, var_pred) + F.mse_loss(mean_true, mean_pred)
    if reg_type == "l1_loss":
        return F.l1_loss(var_true, var_pred) + F.l1_loss(mean_true, mean_pred)
    if reg_type == "kld":
        return compute_kld(mean_true, mean_pred, var_true, var_pred)
    raise ValueError(reg_type)


class StatHookOracle:
    """One CombineNormStatsRegHook_onereg (utils/norm_stats_utils.py:103-258) without the module."""

    def __init__(self, src_mean, src_var, reg_type="l1_loss", momentum=0.1, kind="bn2d", clip_len=None):
        self.src_mean, self.src_var = src_mean, src_var
        self.reg_type, self.kind, self.clip_len = reg_type, kind, clip_len
        self.mean_avg, self.var_avg = MovingAverage(momentum), MovingAverage(momentum)

    def __call__(self, feature):
        m, v = moments(feature, self.kind, self.clip_len)
        self.mean_avg.update(m)
        self.var_avg.update(v)
        self.batch_mean, self.batch_var = m.detach(), v.detach()
        return compute_regularization(self.src_mean, self.mean_avg.avg, self.src_var, self.var_avg.avg, self.reg_type)


# ---------------------------------------------------------------------------------------------
# A5 -- prediction consistency
# ---------------------------------------------------------------------------------------------
def compute_pred_consis(preds):
    """utils/pred_consistency_utils.py:15-31 (L1Loss(reduction='sum'), :8)."""
    bz, n_views, n_class = preds.size()
    softmaxs = [F.softmax(preds[:, v, :], dim=1) for v in range(n_views)]
    avg_softmax = torch.stack(softmaxs, dim=0).mean(0)
    loss = [F.l1_loss(softmaxs[v], avg_softmax, reduction="sum") for v in range(n_views)]
    return sum(loss) / n_views


# ---------------------------------------------------------------------------------------------
# A6 -- closed form of the stat-loss gradient (what the HIP backward implements); the tests
# compare it with autograd through the ops above
# ---------------------------------------------------------------------------------------------
def align_coefficients(ema_mean, ema_var, src_mean, src_var, momentum, reg_type, n):
    c = ema_mean.numel()
    if reg_type == "l1_loss":
        gm = torch.sign(ema_mean - src_mean) / c
        gv = torch.sign(ema_var - src_var) / c
    elif reg_type == "mse_loss":
        gm = 2 * (ema_mean - src_mean) / c
        gv = 2 * (ema_var - src_var) / c
    else:
        dm = src_mean - ema_mean
        gm = -dm / ema_var
        gv = 0.5 / ema_var - (src_var + dm * dm) / (2 * ema_var * ema_var)
    return momentum * gm / n, 2 * momentum * gv / n


# ---------------------------------------------------------------------------------------------
# A9 -- TAM tail in the reference's op order
# ---------------------------------------------------------------------------------------------
def tam_pool(x, n_segment):
    """models/tanet_models/temporal_module.py:45-52 -> pooled (N, C, T)."""
    nt, c, h, w = x.size()
    n = nt // n_segment
    new_x = x.view(n, n_segment, c, h, w).permute(0, 2, 1, 3, 4).contiguous()
    out = F.adaptive_avg_pool2d(new_x.view(n * c, n_segment, h, w), (1, 1))
    return out.view(n, c, n_segment)


def tam_aggregate(x, gate, kern, n_segment):
    """models/tanet_models/temporal_module.py:47-63 given local_activation `gate` (N,C,T) and the
    adaptive kernel `kern` (N*C, 3)."""
    nt, c, h, w = x.size()
    n = nt // n_segment
    new_x = x.view(n, n_segment, c, h, w).permute(0, 2, 1, 3, 4).contiguous()
    new_x = new_x * gate.view(n, c, n_segment, 1, 1)
    out = F.conv2d(new_x.view(1, n * c, n_segment, h * w), kern.view(n * c, 1, 3, 1), bias=None, stride=(1, 1),
                   padding=(1, 0), groups=n * c)
    out = out.view(n, c, n_segment, h, w)
    return out.permute(0, 2, 1, 3, 4).contiguous().view(nt, c, h, w)
