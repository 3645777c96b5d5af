"""Feature-statistics hooks of ViTTA on the HIP kernels.

Interface mirror of utils/norm_stats_utils.py:
    compute_kld :8-16, ComputeNormStatsHook :18-101, CombineNormStatsRegHook_onereg :103-258,
    compute_regularization :531-542
Two execution modes behind the same hook protocol (constructed on a module, `.r_feature`,
`.close()`, `.add_hook_back(module)`):

* stand-alone (engine=None) -- one hook == one reference hook: on every forward it runs the
  single-layer HIP moment reduction with an analytic autograd backward (ops.FeatureMoments), then the
  [C]-sized EMA / loss arithmetic in torch.  Works on any torch model.

* batched (engine=StatAlignEngine) -- the MI355X-first restructuring: the loss is
  separable, nothing downstream in the forward consumes the global moments, so hooks only RECORD
  their feature and splice a gradient-injection node into the graph.  After the forward
  `engine.finish()` does: ONE batched moments launch over all hooked layers -> (ONE all-reduce of
  [cnt|s1|s2] when data-parallel) -> ONE EMA/loss/coefficient launch.  During backward every
  injection node adds a_c + b_c (x - mu_c) to the incoming gradient.  29-42 sync points become 1.
"""
import torch
import torch.nn as nn

from . import _lib
from .utils_ import AverageMeterTensor, MovingAverageTensor


def compute_kld(mean_true, mean_pred, var_true, var_pred):
    """KL( N(mean_true,var_true) || N(mean_pred,var_pred) ) summed over channels."""
    kld = 0.5 * torch.log(var_pred / var_true) + (var_true + (mean_true - mean_pred) ** 2) / (2 * var_pred) - 0.5
    return kld.sum()


def compute_regularization(mean_true, mean_pred, var_true, var_pred, reg_type):
    mean_pred = mean_pred.to(mean_true.device)
    var_pred = var_pred.to(var_true.device)
    if reg_type == "mse_loss":
        return torch.mean((var_true - var_pred) ** 2) + torch.mean((mean_true - mean_pred) ** 2)
    if reg_type == "l1_loss":
        return torch.mean(torch.abs(var_true - var_pred)) + torch.mean(torch.abs(mean_true - mean_pred))
    if reg_type == "kld":
        return compute_kld(mean_true, mean_pred, var_true, var_pred)
    return None  # the reference falls through for any other name (e.g. BNFeatureHook's default 'l2norm')


def feature_kind(module):
    if isinstance(module, nn.BatchNorm1d):
        return "bn1d"
    if isinstance(module, nn.BatchNorm2d):
        return "bn2d"
    if isinstance(module, nn.BatchNorm3d):
        return "bn3d"
    if isinstance(module, nn.LayerNorm):
        return "ln"
    raise Exception(f"undefined module {module}")


def _check_feature(feature, kind):
    if kind == "ln" and feature.dim() != 5:
        raise AssertionError("LayerNorm features must be (B, T, H, W, C)")


class ComputeNormStatsHook:
    """Source-statistics producer (stat_type 'spatiotemp'): stores batch_mean / batch_var over
    (N, T, H, W) of the hooked feature after every forward."""

    def __init__(self, module, clip_len=None, stat_type=None, before_norm=None, batch_size=None, backend=None):
        if stat_type != "spatiotemp":
            raise NotImplementedError("only stat_type 'spatiotemp' is on the ViTTA path")
        self.backend = backend or HipBackend()
        self.hook = module.register_forward_hook(self.hook_fn)
        self.clip_len, self.stat_type, self.before_norm, self.batch_size = clip_len, stat_type, before_norm, batch_size

    def hook_fn(self, module, input, output):
        feature = input[0] if self.before_norm else output
        kind = feature_kind(module)
        if kind == "bn1d":
            raise AssertionError("BatchNorm1d has temporal statistics only")
        _check_feature(feature, kind)
        self.batch_mean, self.batch_var = self.backend.moments(feature.detach(), kind)

    def close(self):
        self.hook.remove()


# --------------------------------------------------------------------------------------------------
# batched engine
# --------------------------------------------------------------------------------------------------
class HipBackend:
    """The product backend: every heavy step is a libvitta_hip launch (CPU tensors raise; there is
    no eager fallback).  tests/ substitute an oracle-backed object with the same five methods to
    exercise the host logic (hook protocol, EMA bookkeeping, data-parallel exchanges) without a GPU."""

    def moments(self, feature, kind):
        from . import ops
        return ops.moments(feature, kind)

    def feature_moments(self, feature, kind):
        from . import ops
        if not feature.is_cuda:
            raise _lib.VittaHipError("feature statistics run on the HIP kernels only; move the model to the GPU")
        return ops.FeatureMoments.apply(feature, kind)

    def make_plan(self, shapes, device):
        from . import ops
        return ops.StatPlan(shapes, device)

    def make_fused_plan(self, shapes, device):
        """Plan for steps whose per-layer triples come from fused BatchNorm passes: every NCHW layer is split over the
        frames like an un-hooked fused BN launch would be (ops._bn_nsplit: >= 1024 workgroups per layer if the frame
        count allows)."""
        from . import ops
        ns = [ops._bn_nsplit(outer, c, inner) if layout == _lib.LAYOUT_NCHW else 0 for outer, c, inner, layout in shapes]
        return ops.StatPlan(shapes, device, nsplit=ns)

    def layout(self, feature, kind):
        from . import ops
        return ops.feature_layout(feature, kind)

    def inject(self, x, gout, kind, mu, a, b, gscale):
        from . import ops
        return ops.stat_align_bwd(x, gout, kind, mu, a, b, gscale)


class FusedSite:
    """Handle a fused BN pass (ops.FusedBNAct) uses to talk to the engine for one hooked layer: where to put
    the partial moments of this step, and -- at backward time -- which coefficient slices to inject."""

    def __init__(self, engine, index):
        self.engine, self.index = engine, index

    def begin(self, x):
        e = self.engine
        plan = e.plan
        outer, c, inner, _ = plan.shapes[self.index]
        if (x.shape[0], x.shape[1], x.shape[2] * x.shape[3]) != (outer, c, inner):
            raise RuntimeError("fused statistics pass: feature shape differs from the planned one")
        e._fused_seen.add(self.index)
        return plan.layer_geometry(self.index)[0], plan.triples_ptr(self.index)

    def coefficients(self):
        e = self.engine
        if not e._gscale_set:
            # loss_reg did not take part in this backward: inject nothing (gscale is 0 on the device)
            pass
        sl = e.plan.channel_slice(self.index)
        return e.plan.mu[sl], e.plan.coef_a[sl], e.plan.coef_b[sl], e.gscale


class FusedLNSite:
    """The same for a hooked LayerNorm on the fused pass (ops.FusedLayerNorm): the kernel's channel sums go straight
    into the engine's [cnt | s1 | s2] statistics (no partial triples, no finalize launch)."""

    def __init__(self, engine, index):
        self.engine, self.index = engine, index

    def begin(self, rows, c):
        e = self.engine
        plan = e.plan
        outer, pc, inner, _ = plan.shapes[self.index]
        if (rows, c) != (outer * inner, pc):
            raise RuntimeError("fused statistics pass: feature shape differs from the planned one")
        if not e._fused_seen:  # first fused layer of the step: the column sums ADD into [s1 | s2]
            e._zero_stats(plan)
            plan.cnt_src = None
            if e._colsums is not None:
                e._colsums.clear()  # (a forward that raised leaves nothing behind)
        e._fused_seen.add(self.index)
        e._fused_direct = True
        sl = plan.channel_slice(self.index)
        return e.src_mean[sl], plan.s1[sl], plan.s2[sl], plan.cnt[self.index:self.index + 1]

    def colsum_queue(self):
        """ops.ColsumQueue the pass leaves its column sums in (the engine issues them in reduce_local: one launch for all hooked
        layers), or None: the pass issues its own."""
        e = self.engine
        if e._colsums is None:
            from . import ops
            e._colsums = ops.ColsumQueue()
        return e._colsums

    coefficients = FusedSite.coefficients


class _Inject(torch.autograd.Function):
    """Identity in the forward; adds the stat-loss gradient of its layer in the backward."""

    @staticmethod
    def forward(ctx, feature, engine, index, kind):
        ctx.engine, ctx.index, ctx.kind = engine, index, kind
        ctx.save_for_backward(feature)
        return feature.view_as(feature)

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        return ctx.engine._inject(ctx.index, ctx.kind, x, gout.contiguous()), None, None, None


class _LossReg(torch.autograd.Function):
    """Carries the value of loss_reg; its backward publishes the upstream gradient (e.g.
    lambda_feature_reg) to the device scalar the injection kernels scale with.  It is created after
    every injection node of the step, so the autograd engine (highest sequence number first) runs it
    before any of them."""

    @staticmethod
    def forward(ctx, anchor, engine, value, tie=None, fresh=False):
        ctx.engine = engine
        ctx.tie = None if tie is None else (tie.shape, tie.dtype, tie.device)
        # (no copy: the HIP plan hands out a tensor of its own per alignment launch -- ops.Plan.align; other backends' totals are
        # cloned, they may be views of state the next step overwrites)
        return value.view_as(value) if fresh else value.clone()

    @staticmethod
    def backward(ctx, g):
        if g.data_ptr() != ctx.engine.gscale.data_ptr():  # (ops.WeightedLoss writes the upstream gradient into gscale itself)
            ctx.engine.gscale.copy_(g.reshape(1))
        ctx.engine._gscale_set = True
        # `tie` (the model output) receives a zero gradient: its only purpose is to make this backward walk the model's
        # graph -- the injection nodes hang off it -- when loss_reg is the WHOLE loss (no consistency term)
        gt = None if ctx.tie is None else torch.zeros(ctx.tie[0], dtype=ctx.tie[1], device=ctx.tie[2])
        return None, None, None, gt, None


class StatAlignEngine:
    """Owns the packed EMA state, the source statistics and the launch plans of all hooked layers."""

    def __init__(self, reg_type="l1_loss", momentum=0.1, backend=None, process_group=None, distributed=None):
        if reg_type not in _lib.REG_TYPES:
            raise ValueError(f"undefined reg_type {reg_type}")
        self.reg_type, self.momentum = reg_type, float(momentum)
        self.backend = backend or HipBackend()
        self.process_group = process_group
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size(process_group) > 1
        self.distributed = distributed
        self.zero_pool = None  # tta.FlatArena (or anything with reserve_zeroed): its per-step fill then covers the statistics
        self.hooks = []
        self._src = []
        self._plans = {}
        self._feats = {}
        self._kinds = {}
        self._built = False
        self.plan = None
        self.gscale = None
        self._gscale_set = False
        self._fused_seen = set()
        self._fused_direct = False
        self._colsums = None  # ops.ColsumQueue of the fused LayerNorm passes' column sums (FusedLNSite.colsum_queue)
        self._fused_plans = {}
        self.timing_events = None  # bench.py: callable returning (start, stop) events per step

    # -- registration -------------------------------------------------------------------------
    def register(self, hook, src_mean, src_var):
        if self._built:
            raise RuntimeError("cannot register hooks after the first step")
        self.hooks.append(hook)
        self._src.append((torch.as_tensor(src_mean, dtype=torch.float32).reshape(-1),
                          torch.as_tensor(src_var, dtype=torch.float32).reshape(-1)))
        return len(self.hooks) - 1

    def _build(self, device):
        self.device = device
        self.src_mean = torch.cat([s[0] for s in self._src]).to(device).contiguous()
        self.src_var = torch.cat([s[1] for s in self._src]).to(device).contiguous()
        self.ema_mean = torch.zeros_like(self.src_mean)  # avg0 = 0 (utils/utils_.py:208)
        self.ema_var = torch.zeros_like(self.src_var)
        self.gscale = torch.zeros(1, dtype=torch.float32, device=device)
        self._anchor = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)
        self._built = True

    # -- per-step protocol --------------------------------------------------------------------
    def fused_site(self, index, x):
        """A FusedSite if this step can take the fused BN path for hooked layer `index` (a plan for exactly
        these shapes exists, i.e. not the very first step and not a ragged batch); else None."""
        if not self._built or self.plan is None or not torch.is_grad_enabled() or not hasattr(self.plan, "triples_ptr"):
            return None
        outer, c, inner, _ = self.plan.shapes[index]
        if (x.shape[0], x.shape[1], x.shape[2] * x.shape[3]) != (outer, c, inner) or self._feats:
            return None
        if not self._fused_seen and hasattr(self.backend, "make_fused_plan"):
            # first fused layer of the step: switch to the plan whose frame splits suit one launch PER layer (the
            # batched-kernel plan walks all 16 frames in one workgroup: 49 workgroups for a 256 x 14 x 14 layer)
            key = tuple(self.plan.shapes)
            fused = self._fused_plans.get(key)
            if fused is None:
                fused = self._fused_plans[key] = self.backend.make_fused_plan(self.plan.shapes, self.device)
            self.plan = fused
        return FusedSite(self, index)

    def begin_direct(self, shapes, device):
        """Every hooked layer of this step deposits its additive statistics straight into [s1 | s2]: the convolution
        that produces the layer adds the shifted sums of its tile in the epilogue (vitta_conv_f32, VITTA_CONV_STATS).
        `shapes`: (frames, C, H*W, NCHW) per hooked layer in hook order.  Returns the plan whose buffers take part."""
        if self._feats or self._fused_seen:
            raise RuntimeError("a step must be either all-fused or all-recorded")
        if not self._built:
            self._build(device)
        shapes = tuple(tuple(int(v) for v in s) for s in shapes)
        if len(shapes) != len(self.hooks):
            raise RuntimeError("one shape per hooked layer expected")
        for (_, c, _, _), (sm, _) in zip(shapes, self._src):
            if c != sm.numel():
                raise RuntimeError(f"source statistics have {sm.numel()} channels, feature has {c}")
        plan = self._plans.get(shapes)
        if plan is None:
            plan = self._plans[shapes] = self.backend.make_plan(shapes, self.device)
        if getattr(plan, "_direct_cnt", None) is None:
            plan._direct_cnt = torch.tensor([float(o * i) for o, _, i, _ in shapes], dtype=torch.float32, device=self.device)
        self.plan = plan
        self._zero_stats(plan)
        if self.distributed or not hasattr(plan, "cnt_src"):
            plan.cnt.copy_(plan._direct_cnt)  # (the counts take part in the ranks' sum)
        else:
            plan.cnt_src = plan._direct_cnt   # one rank: constants of the plan, read in place by the alignment launch
        self._fused_seen = set(range(len(self.hooks)))
        self._fused_direct = True
        return plan

    def fused_ln_site(self, index, x):
        """A FusedLNSite if this step can take the fused LayerNorm path for hooked layer `index`: a plan for exactly
        these shapes exists and EVERY hooked layer of the plan is a channels-last LayerNorm the kernel covers (a step is
        all-fused or all-recorded)."""
        if not self._built or self.plan is None or not torch.is_grad_enabled() or self._feats \
                or not hasattr(self.plan, "s1"):
            return None
        ok = getattr(self.plan, "_ln_direct_ok", None)
        if ok is None:
            from . import ops
            ok = self.plan._ln_direct_ok = all(layout == _lib.LAYOUT_NHWC and inner == 1 and ops.ln_supported(c)
                                               for _, c, inner, layout in self.plan.shapes)
        outer, c, inner, _ = self.plan.shapes[index]
        if not ok or (x.numel() // x.shape[-1], x.shape[-1]) != (outer * inner, c):
            return None
        return FusedLNSite(self, index)

    def collect(self, index, feature, kind):
        """Called by hook `index` during the forward; returns the tensor that replaces the output."""
        if not self._built:
            self._build(feature.device)
        if index in self._feats:
            raise RuntimeError("hooked layer ran twice in one step; call engine.finish() between forwards")
        if not feature.is_contiguous():
            feature = feature.contiguous()
        out = _Inject.apply(feature, self, index, kind) if torch.is_grad_enabled() else feature
        self._feats[index] = feature.detach()
        self._kinds[index] = kind
        return out

    def finish(self, tie=None):
        """Batched moments -> (all-reduce) -> EMA + loss + coefficients.  Returns loss_reg."""
        self.reduce_local()
        self.exchange()
        return self.finish_global(tie)

    def exchange(self):
        """The moments all-reduce (data-parallel only): one SUM over the packed [cnt | s1 | s2] buffer."""
        if self.distributed:
            from . import exchange_timing as XT
            with XT.timed("moments", self.plan.stats.numel() * 4):
                torch.distributed.all_reduce(self.plan.stats, op=torch.distributed.ReduceOp.SUM, group=self.process_group)

    def finish_global(self, tie=None):
        """EMA update + loss + backward coefficients from the (reduced) statistics.  `tie`: the model output of this
        step when loss_reg alone is backpropagated (prediction consistency off): the statistics gradient is injected by
        nodes of the MODEL's graph, which a backward from loss_reg reaches only through that output."""
        return self._align_and_wrap(self.plan, tie)

    def _zero_stats(self, plan):
        """[cnt | s1 | s2] of the step to zero: inside the step's ONE fill where an arena offers its zeroed tail (zero_pool =
        tta.FlatArena: the plan's statistics move there on first use), by a launch of their own otherwise."""
        z = getattr(plan, "_zeroed", None)
        if z is None and self.zero_pool is not None and hasattr(plan, "rebind_stats") and not getattr(plan, "_zeroed_tried", False):
            plan._zeroed_tried = True
            z = self.zero_pool.reserve_zeroed(plan.stats.numel() * 4)
            if z is not None:  # (a fresh slice of the tail is zero: nothing has written there)
                plan.rebind_stats(z.tensor)
                plan._zeroed = z
        if z is not None and z.fresh():
            return
        plan.stats.zero_()

    def reduce_local(self):
        """This rank's additive statistics of the step into plan.stats (no communication)."""
        n = len(self.hooks)
        if self._fused_seen:
            # every hooked layer went through a fused BN pass: the partial triples are already in the plan's
            # workspace, only the tiny combine + align remain
            if len(self._fused_seen) != n or self._feats:
                raise RuntimeError("a step must be either all-fused or all-recorded")
            self._fused_seen = set()
            if self._fused_direct:  # fused LayerNorm passes wrote [cnt | s1 | s2] themselves
                self._fused_direct = False
                if self._colsums is not None:  # ... or left their column sums for this ONE launch
                    self._colsums.flush()
            else:
                self.plan.finalize(self.src_mean)
            return
        if len(self._feats) != n:
            missing = [i for i in range(n) if i not in self._feats]
            raise RuntimeError(f"hooks {missing} did not fire in this forward")
        feats = [self._feats[i] for i in range(n)]
        kinds = [self._kinds[i] for i in range(n)]
        shapes = tuple(self.backend.layout(f, k) for f, k in zip(feats, kinds))
        for (_, c, _, _), (sm, _) in zip(shapes, self._src):
            if c != sm.numel():
                raise RuntimeError(f"source statistics have {sm.numel()} channels, feature has {c}")
        plan = self._plans.get(shapes)
        if plan is None:
            plan = self._plans[shapes] = self.backend.make_plan(shapes, self.device)
        self.plan = plan
        plan.moments(feats, self.src_mean, **({"events": self.timing_events()} if self.timing_events else {}))
        self._feats, self._kinds = {}, {}

    def _align_and_wrap(self, plan, tie=None):
        in_launch = getattr(plan, "zeroes_in_align", False)  # (the HIP plan resets the gradient scale inside the alignment launch)
        total, layer = plan.align(self.src_mean, self.ema_mean, self.ema_var, self.src_mean, self.src_var,
                                  self.momentum, self.reg_type, **({"zero": self.gscale} if in_launch else {}))
        for i, h in enumerate(self.hooks):
            h.r_feature = layer[i]
        self._feats, self._kinds = {}, {}
        if not in_launch:
            self.gscale.zero_()
        self._gscale_set = False
        if tie is not None and not tie.requires_grad:
            tie = None
        return _LossReg.apply(self._anchor, self, total[0], tie, bool(getattr(total, "_vitta_fresh", False)))

    def finish_empty(self):
        """Ragged tail of a data-parallel run: this rank has no video in the step but still joins the
        moments all-reduce (contributing n = 0) so its EMA state stays identical to the others'."""
        if self.plan is None or not self.distributed:
            raise RuntimeError("finish_empty needs a previous step of a distributed run")
        self.plan.stats.zero_()
        torch.distributed.all_reduce(self.plan.stats, op=torch.distributed.ReduceOp.SUM, group=self.process_group)
        total, layer = self.plan.align(self.src_mean, self.ema_mean, self.ema_var, self.src_mean, self.src_var,
                                       self.momentum, self.reg_type)
        for i, h in enumerate(self.hooks):
            h.r_feature = layer[i]
        self._feats, self._kinds = {}, {}
        return total[0].clone()

    def _inject(self, index, kind, x, gout):
        if not self._gscale_set:
            # loss_reg did not take part in this backward: nothing to add
            return gout
        sl = self.plan.channel_slice(index)
        return self.backend.inject(x, gout, kind, self.plan.mu[sl], self.plan.coef_a[sl], self.plan.coef_b[sl],
                                   self.gscale)

    def layer_stats(self, index):
        """(ema_mean, ema_var) slices of a hooked layer, for inspection/tests."""
        sl = self.plan.channel_slice(index)
        return self.ema_mean[sl], self.ema_var[sl]


# --------------------------------------------------------------------------------------------------
# the hook
# --------------------------------------------------------------------------------------------------
class CombineNormStatsRegHook_onereg:
    """One regularisation term per hooked norm layer, statistics pooled over all augmented views."""

    def __init__(self, module, clip_len=None, spatiotemp_stats_clean_tuple=None, reg_type="mse_loss",
                 moving_avg=None, momentum=0.1, stat_type_list=None, reduce_dim=True, before_norm=None,
                 if_sample_tta_aug_views=None, n_augmented_views=None, engine=None, backend=None):
        assert stat_type_list == ["spatiotemp"]
        self.backend = backend or (engine.backend if engine is not None else HipBackend())
        self.clip_len, self.reg_type, self.moving_avg, self.momentum = clip_len, reg_type, moving_avg, momentum
        self.stat_type_list, self.reduce_dim, self.before_norm = stat_type_list, reduce_dim, before_norm
        self.if_sample_tta_aug_views, self.n_augmented_views = if_sample_tta_aug_views, n_augmented_views
        self.source_mean_spatiotemp, self.source_var_spatiotemp = spatiotemp_stats_clean_tuple
        self.kind = feature_kind(module)
        self.engine, self.index = None, None
        # BatchNorm1d carries no spatio-temporal statistics: the term is identically 0 whether or not the hook
        # fires (a fused TAM-branch kernel bypasses the module call)
        self.r_feature = torch.zeros(()) if self.kind == "bn1d" else None
        if self.kind != "bn1d" and self.source_mean_spatiotemp is not None:
            self.source_mean_spatiotemp = torch.as_tensor(self.source_mean_spatiotemp, dtype=torch.float32)
            self.source_var_spatiotemp = torch.as_tensor(self.source_var_spatiotemp, dtype=torch.float32)
            if engine is not None:
                if not moving_avg:
                    raise NotImplementedError("the batched engine covers moving_avg=True")
                # before_norm (utils/norm_stats_utils.py:185: feature = input[0]): the engine only sees a feature tensor; the
                # fused module passes decline such a hook (it fires as a callback), the trunk reads the raw convolution output
                self.engine = engine
                self.index = engine.register(self, self.source_mean_spatiotemp, self.source_var_spatiotemp)
        if self.moving_avg:
            self.mean_avgmeter_spatiotemp = MovingAverageTensor(momentum=momentum)
            self.var_avgmeter_spatiotemp = MovingAverageTensor(momentum=momentum)
        else:
            self.mean_avgmeter_spatiotemp, self.var_avgmeter_spatiotemp = AverageMeterTensor(), AverageMeterTensor()
        self.pre_hook = None
        self.add_hook_back(module)

    def pre_hook_fn(self, module, input):
        """before_norm on the batched engine: the engine's injection node (identity forward, statistics-loss gradient added in the
        backward) must sit on the layer's INPUT -- a forward hook can only replace the output -- so the feature is collected by
        a forward PRE-hook that hands the module the wrapped input."""
        feature = input[0]
        _check_feature(feature, self.kind)
        if self.kind == "bn2d" and feature.shape[0] % self.clip_len != 0:
            raise ValueError(f"{feature.shape[0]} frames are not a multiple of clip_len {self.clip_len}")
        return (self.engine.collect(self.index, feature, self.kind),) + tuple(input[1:])

    def hook_fn(self, module, input, output):
        if self.pre_hook is not None:  # collected on the way in
            return None
        feature = input[0] if self.before_norm else output
        if self.kind == "bn1d":
            # BatchNorm1d carries temporal statistics only ('temp' not in ['spatiotemp']): contributes 0
            self.r_feature = torch.zeros((), dtype=torch.float32, device=feature.device)
            return None
        _check_feature(feature, self.kind)
        if self.kind == "bn2d" and feature.shape[0] % self.clip_len != 0:
            raise ValueError(f"{feature.shape[0]} frames are not a multiple of clip_len {self.clip_len}")
        if self.engine is not None:
            return self.engine.collect(self.index, feature, self.kind)
        if self.kind == "bn2d":
            nmt, c, h, w = feature.shape
            bz_m = nmt // self.clip_len
            self.feature_shape = (bz_m, c, self.clip_len, h, w)
        elif self.kind == "bn3d":
            bz_m = feature.shape[0]
            self.feature_shape = tuple(feature.shape)
        else:
            b, t, h, w, c = feature.shape
            bz_m = b
            self.feature_shape = (b, c, t, h, w)
        bz = bz_m // self.n_augmented_views if self.if_sample_tta_aug_views else bz_m
        batch_mean, batch_var = self.backend.feature_moments(feature, self.kind)
        if self.moving_avg:
            self.mean_avgmeter_spatiotemp.update(batch_mean)
            self.var_avgmeter_spatiotemp.update(batch_var)
        else:
            self.mean_avgmeter_spatiotemp.update(batch_mean, n=bz)
            self.var_avgmeter_spatiotemp.update(batch_var, n=bz)
        dev = feature.device
        self.r_feature = compute_regularization(self.source_mean_spatiotemp.to(dev), self.mean_avgmeter_spatiotemp.avg,
                                                self.source_var_spatiotemp.to(dev), self.var_avgmeter_spatiotemp.avg,
                                                self.reg_type)
        return None

    def add_hook_back(self, module):
        self.hook = module.register_forward_hook(self.hook_fn)
        if self.engine is not None and self.before_norm and self.kind != "bn1d":
            self.pre_hook = module.register_forward_pre_hook(self.pre_hook_fn)

    def close(self):
        self.hook.remove()
        if self.pre_hook is not None:
            self.pre_hook.remove()
            self.pre_hook = None
