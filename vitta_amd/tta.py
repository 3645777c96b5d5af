"""The ViTTA driver: online test-time adaptation loop, source-only validation, source-statistics
producer, model/dataset factories.

Interface mirror of corpus/basics.py:
    tta_standard :403-747   validate :96-217   compute_statistics :220-307
    get_model :1447-1493    get_dataset_tanet :1230-1291   get_dataset_videoswin :1191-1228
Same call signatures, same per-video protocol (adapt on video i -> evaluate video i -> next video
without resetting the model), same log-line formats.  MI355X-first differences:

* statistics hooks run in the batched engine (one moments launch + one align launch per step, gradient
  injected during backward) whenever the configuration allows it (stat_reg 'mean_var', moving_avg; before_norm hooks
  collect through a forward pre-hook); otherwise each hook falls back to the stand-alone HIP op;
* device-agnostic (no hard-coded .cuda()); one process per GPU.  Under torch.distributed (RCCL) the
  test videos are sharded round-robin over ranks, the packed moments [cnt|s1|s2] are all-reduced once
  per step before the EMA update and the gradients once before the optimizer step -- R ranks x 1
  video is exactly the reference run with batch_size = R;
* no per-video host synchronisation: per-video metrics are read back with a lag (DeferredLog).
"""
import contextlib
import copy as cp
import os
import os.path as osp
import time

import numpy as np
import torch
import torch.nn as nn

from .bns_utils import choose_layers, collect_bn_params, freeze_except_bn
from .norm_stats import CombineNormStatsRegHook_onereg, ComputeNormStatsHook, StatAlignEngine
from .pred_consistency import compute_pred_consis
from .utils_ import AverageMeter, accuracy

# tests may point this at a factory of oracle-backed backends to run the host logic without a GPU;
# the product default (None) is the HIP backend, which raises on CPU tensors.
BACKEND_FACTORY = None

# tta_standard runs this many videos eagerly, then captures the step into hipGraphs (None: never)
GRAPH_AFTER_STEPS = 3
# hipStreamCaptureModeThreadLocal: other threads of the process (the RCCL watchdog of a data-parallel run, data-loader
# pin-memory threads) may keep calling into the runtime while this thread captures
CAPTURE_MODE = "thread_local"
# the TANet head of the adaptation pass + the loss combination as our own launches (ops.TanetHead, ops.WeightedLoss); "0": module chain
FUSED_HEAD = os.environ.get("VITTA_FUSED_HEAD", "1") != "0"
# overlapped step with trainable convolution weights: pack them once per step, not once per pass (ViTTAAdapter._prepacked)
PREPACK = os.environ.get("VITTA_PREPACK", "1") != "0"
# Round 6: the overlapped schedule of ONE process as SEPARATE graphs on two streams -- [re-pack] | forward + backward | optimizer on the
# step's stream, the evaluation forward as a graph of its own on the side stream -- instead of one graph with a forked branch
# (VITTA_SPLIT_GRAPHS=0).  Measured (tools/debug/view_split_probe.py): the evaluation graph beside the adaptation graph costs the
# adaptation nothing (4.68 vs 4.67 ms), the forked single graph ~0.4 ms per step.
SPLIT_GRAPHS = os.environ.get("VITTA_SPLIT_GRAPHS", "1") != "0"
# forked forms (one graph / eager): the evaluation forks in front of trunk block EVAL_FORK_BLOCK of the adaptation forward instead of
# in front of the whole step (0) -- beside the latency-bound layer-3 / 4 launches rather than the bandwidth-bound layer-1 ones
EVAL_FORK_BLOCK = int(os.environ.get("VITTA_EVAL_FORK_BLOCK", "0"))

NUM_CLASSES = {"ucf101": 101, "hmdb51": 51, "kinetics": 400, "somethingv2": 174, "kth": 6, "u2h": 12, "h2u": 12}


class SingleDeviceParallel(nn.Module):
    """Stand-in for nn.DataParallel(model, device_ids=[0]) (corpus/main_eval.py:61,65): one process
    drives one GPU, but sub-module names keep the `module.` prefix the checkpoints and the Swin
    `chosen_blocks` ('module.backbone.layers.2', tta_swin_ucf101.py:40) rely on."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


# ------------------------------------------------------------------------------------------------
# factories
# ------------------------------------------------------------------------------------------------
def get_model(args, num_classes, logger=None):
    if args.arch == "tanet":
        from .tanet import TSN
        return TSN(num_classes, args.clip_length, args.modality, base_model="resnet50", consensus_type="avg",
                   img_feature_dim=args.img_feature_dim, tam=True, non_local=False, partial_bn=args.partial_bn)
    if args.arch == "videoswintransformer":
        from .swin import Recognizer3D
        return Recognizer3D(num_classes=num_classes, patch_size=args.patch_size, window_size=args.window_size,
                            drop_path_rate=args.drop_path_rate)
    raise Exception(f"{args.arch} is not a valid model!")


def get_dataset_tanet(args, split="train", dataset_type=None):
    from . import data
    return data.build_tanet_dataset(args, split, dataset_type)


def get_dataset_videoswin(args, split="train", dataset_type=None):
    from . import data
    return data.build_videoswin_dataset(args, split, dataset_type)


def _device_of(model):
    return next(model.parameters()).device


def _loader(dataset, args):
    workers = 0 if getattr(dataset, "on_device", False) else args.workers
    return torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, shuffle=False, num_workers=workers,
                                       pin_memory=workers > 0)


def _n_clips(args):
    return int(args.sample_style.split("-")[-1]) if args.arch == "tanet" else args.num_clips


# test / rehearsal switch: run the two data-parallel exchanges (and the segmented graphs around them) even in a process
# group of ONE rank -- the only way to drive the RCCL calls themselves on a single-GPU box (bench.py --force-exchanges)
FORCE_EXCHANGES = False


def _dist():
    d = torch.distributed
    if d.is_available() and d.is_initialized() and d.get_world_size() > 1:
        return d.get_rank(), d.get_world_size()
    return 0, 1


# ------------------------------------------------------------------------------------------------
# source statistics
# ------------------------------------------------------------------------------------------------
def load_source_statistics(args, chosen_layers):
    """np.load both object arrays and align them with `chosen_layers` by position
    (corpus/basics.py:480-509): TANet lists hold BN2d/3d entries only -> None at BatchNorm1d slots."""
    mean_list = list(np.load(args.spatiotemp_mean_clean_file, allow_pickle=True))
    var_list = list(np.load(args.spatiotemp_var_clean_file, allow_pickle=True))
    if args.arch == "tanet":
        means, vars_, k = [], [], 0
        for _, layer in chosen_layers:
            if isinstance(layer, nn.BatchNorm1d):
                means.append(None)
                vars_.append(None)
            else:
                means.append(mean_list[k])
                vars_.append(var_list[k])
                k += 1
    else:
        means, vars_ = mean_list, var_list
    assert len(means) == len(chosen_layers), (len(means), len(chosen_layers))
    return means, vars_


def candidate_layers_for(args, model):
    """All norm layers eligible for hooks, in named_modules() order (corpus/basics.py:486-505)."""
    if args.arch == "tanet":
        return choose_layers(model, [nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d])
    return choose_layers(model, [nn.LayerNorm])[1:]  # the first LN sees (B, L, C): excluded


def select_hooked(args, chosen_layers):
    """[(position, name, layer)] of the layers whose name contains a chosen block (basics.py:571-573)."""
    return [(i, nm, layer) for i, (nm, layer) in enumerate(chosen_layers)
            if any(block in nm for block in args.chosen_blocks)]


# ------------------------------------------------------------------------------------------------
# data-parallel gradient exchange
# ------------------------------------------------------------------------------------------------
class FlatArena:
    """Trainable parameters AND their gradients live in two flat fp32 buffers; every `p.data` / `p.grad` is a
    view.  The optimizer sees ONE tensor: its whole step is a handful of launches instead of a handful per
    parameter tensor (capturable Adam over 170 affine tensors issued ~290 three-microsecond kernels per step in
    the r1g profile), and the data-parallel gradient exchange is ONE all-reduce of `grad` with no flatten copies.
    Element-wise optimizers (SGD with momentum / weight decay, Adam) compute exactly what they compute per tensor.
    SUM, not mean, across ranks: loss_consis is a sum over videos and loss_reg is one global scalar whose per-rank
    partial derivatives add."""

    ALIGN = 64  # floats: every tensor starts on a 256-byte boundary (the HIP kernels use 16-byte loads on weights)
    # floats behind the gradients that the SAME fill zeroes: what a step needs zeroed before its forward besides the gradients (the
    # statistics engine's additive sums, the trunk's fixed-point pooling sums) lives there instead of being filled by launches of its
    # own (reserve_zeroed).  1 MiB: the fill of the Adam-affine arena (58 K floats) stays a few microseconds.
    ZERO_SLACK = 1 << 18

    class Zeroed:
        """A slice of the arena's zeroed tail.  fresh(): True once after every fill of the arena (zero_step) -- the slice may then be
        used as zeroed; False: no fill since the last use, the user zeroes it itself."""

        def __init__(self, arena, tensor):
            self.arena, self.tensor, self._seen = arena, tensor, -1

        def fresh(self):
            e = self.arena._fills
            ok = e != self._seen
            self._seen = e
            return ok

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        pad = lambda k: -(-k // self.ALIGN) * self.ALIGN
        n = sum(pad(p.numel()) for p in self.params)
        dev = self.params[0].device
        # the gaps stay zero in both buffers: a zero gradient on a zero weight is a fixed point of SGD and Adam
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._grad_all = torch.zeros(n + self.ZERO_SLACK, dtype=torch.float32, device=dev)
        self.grad = self._grad_all[:n]
        self._tail, self._fills = n, 0
        off = 0
        self._all_views, self._learnt = [], False
        self.ranges = {}  # id(param) -> (first float, end) in the arena
        self._pending = []  # async bucket reductions of the running step
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.ranges[id(p)] = (off, off + pad(k))
                flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = flat[off:off + k].view_as(p)
                p.grad = self.grad[off:off + k].view_as(p)
                p._vitta_arena_view = p.grad  # found by ops._grad_sink even while `.grad` is detached
                self._all_views.append(p.grad)
                off += pad(k)
        self.flat_param = nn.Parameter(flat)  # shares storage with every p.data view
        self.flat_param.grad = self.grad

    def zero_grad(self):
        """Gradients ONLY (what optimizer.zero_grad() means, corpus/basics.py:669-671: forward, optimizer.zero_grad(), backward): the
        tail behind them holds FORWARD state of the running step -- the trunk's pooled sums its TAM backward reads, the engine's
        [cnt | s1 | s2] -- and must survive a zero_grad() issued between a forward and its backward."""
        self.grad.zero_()

    def zero_step(self):
        """The START of a step: gradients + everything reserved behind them in ONE fill (the adapter's own step and its captures call
        this before the forward; nobody else does)."""
        self._grad_all.zero_()
        self._fills += 1

    def reserve_zeroed(self, nbytes):
        """A FlatArena.Zeroed over `nbytes` (256-byte aligned) of the tail that zero_step() fills, or None when the tail is full."""
        k = -(-int(nbytes) // 256) * 64
        if self._tail + k > self._grad_all.numel():
            return None
        z = FlatArena.Zeroed(self, self._grad_all[self._tail:self._tail + k])
        self._tail += k
        return z

    # Gradients our backward kernels do not write themselves (every weight in SGD-all mode, norm layers on the torch
    # fallback) reach a parameter through autograd's AccumulateGrad.  With a live `.grad` view that is one in-place add
    # launch per tensor; with `.grad = None` AccumulateGrad just keeps the incoming tensor, and ONE multi-tensor copy
    # moves them all into the arena afterwards.  Parameters whose kernels accumulate straight into the arena
    # (`ops._grad_sink` finds the view through `_vitta_arena_view` and marks them) keep their view attached.
    def before_backward(self):
        if not self._learnt:
            return
        for p in self.params:
            if not getattr(p, "_vitta_direct_grad", False):
                p.grad = None

    def after_backward(self):
        if not self._learnt:  # first backward: every view was attached (plain AccumulateGrad adds)
            self._learnt = True
            return
        src, dst = [], []
        for p, view in zip(self.params, self._all_views):
            g = p.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                src.append(g.reshape(view.shape))
                dst.append(view)
            p.grad = view
        if src:
            torch._foreach_copy_(dst, src)

    def all_reduce(self):
        from . import exchange_timing as XT
        with XT.timed("gradients", self.grad.numel() * 4):
            torch.distributed.all_reduce(self.grad, op=torch.distributed.ReduceOp.SUM)

    # -- bucketed exchange: slices of the arena reduced as soon as their gradients are final ----------------------------
    def span(self, params):
        """[lo, hi) of the arena covered by `params` (None if none of them is trainable); they must be contiguous."""
        r = [self.ranges[id(p)] for p in params if id(p) in self.ranges]
        if not r:
            return None
        lo, hi = min(a for a, _ in r), max(b for _, b in r)
        if sum(b - a for a, b in r) != hi - lo:
            raise ValueError("the parameters of a bucket must be contiguous in the arena")
        return lo, hi

    def all_direct(self, params):
        """Every trainable parameter of the group has its gradient written straight into the arena by our kernels (known
        after the first backward): only then is the slice final when the group's backward has run."""
        return self._learnt and all(getattr(p, "_vitta_direct_grad", False) for p in params if id(p) in self.ranges)

    def reduce_range(self, lo, hi, async_op=True):
        if hi <= lo:
            return
        from . import exchange_timing as XT
        if XT.active():  # (bench.py's decomposition steps: one blocking call per bucket, an event pair around it)
            with XT.timed("gradients", (hi - lo) * 4):
                torch.distributed.all_reduce(self.grad[lo:hi], op=torch.distributed.ReduceOp.SUM)
            return
        w = torch.distributed.all_reduce(self.grad[lo:hi], op=torch.distributed.ReduceOp.SUM, async_op=async_op)
        if async_op and w is not None:
            self._pending.append(w)

    def wait_pending(self):
        for w in self._pending:
            w.wait()
        self._pending = []


class DeferredLog:
    """Per-video metrics are written to a device row and fetched with a lag so the host never blocks
    on the stream inside the loop; the lines come out in order, formatted like corpus/basics.py:730-738."""

    FIELDS = 5  # loss_reg, loss_consis, loss_ce, prec1, prec5

    def __init__(self, device, logger, total, verbose, lag=8, epoch=1, print_freq=1):
        self.logger, self.total, self.verbose, self.lag = logger, total, verbose, lag
        self.device, self.epoch, self.print_freq = device, epoch, print_freq
        self.pending = []
        self.meters = dict(batch_time=AverageMeter(), loss_reg=AverageMeter(), loss_consis=AverageMeter(),
                           loss_ce=AverageMeter(), top1=AverageMeter(), top5=AverageMeter())

    def push(self, batch_id, row, bz, elapsed):
        host = torch.empty(self.FIELDS, dtype=torch.float32, pin_memory=row.is_cuda)
        host.copy_(row, non_blocking=True)
        ev = torch.cuda.Event() if row.is_cuda else None
        if ev is not None:
            ev.record()
        self.pending.append((batch_id, host, ev, bz, elapsed))
        while len(self.pending) > self.lag:
            self._emit(self.pending.pop(0), wait=True)
        while self.pending and (self.pending[0][2] is None or self.pending[0][2].query()):
            self._emit(self.pending.pop(0), wait=False)

    def flush(self):
        while self.pending:
            self._emit(self.pending.pop(0), wait=True)

    def _emit(self, item, wait):
        batch_id, host, ev, bz, elapsed = item
        if ev is not None and wait:
            ev.synchronize()
        reg, consis, ce, p1, p5 = host.tolist()
        m = self.meters
        m["batch_time"].update(elapsed)
        m["loss_reg"].update(reg, bz)
        m["loss_consis"].update(consis, bz)
        m["loss_ce"].update(ce, bz)
        m["top1"].update(p1, bz)
        m["top5"].update(p5, bz)
        if self.verbose and self.logger is not None and batch_id % self.print_freq == 0:
            self.logger.debug(("TTA Epoch{epoch}: [{0}/{1}]\t"
                               "Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                               "Loss reg {loss_reg.val:.4f} ({loss_reg.avg:.4f})\t"
                               "Loss consis {loss_consis.val:.4f} ({loss_consis.avg:.4f})\t"
                               "Prec@1 {top1.val:.3f} ({top1.avg:.3f})\t"
                               "Prec@5 {top5.val:.3f} ({top5.avg:.3f})").format(
                batch_id, self.total, epoch=self.epoch, batch_time=m["batch_time"], loss_reg=m["loss_reg"],
                loss_consis=m["loss_consis"], top1=m["top1"], top5=m["top5"]))


# ------------------------------------------------------------------------------------------------
# adapter: model + optimizer + hooks of one adaptation run
# ------------------------------------------------------------------------------------------------
class ViTTAAdapter:
    """Everything `tta_standard` sets up before the first video (corpus/basics.py:525-601), plus the
    adapt / evaluate steps of the loop body so bench.py and the tests can drive them directly."""

    def __init__(self, model_origin, args, engine_backend=None, use_engine=None, copy=True):
        self.args = args
        # tta_standard adapts a copy (basics.py:527); the epoch-style test_time_adapt adapts the caller's model
        self.model = cp.deepcopy(model_origin) if copy else model_origin
        model = self.model
        self.device = _device_of(model)
        self.rank, self.world = _dist()
        self.bn_types = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)
        self.chosen_layers = candidate_layers_for(args, model)

        if args.update_only_bn_affine:
            kinds = list(self.bn_types) if args.arch == "tanet" else [nn.LayerNorm]
            freeze_except_bn(model, bn_condidiate_layers=kinds)
            params, self.param_names = collect_bn_params(model, bn_candidate_layers=kinds)
            self.arena = FlatArena(params)
            if self.device.type == "cuda":  # one launch; the step counter is a device scalar (graph capturable)
                from .optim import FlatAdam
                self.optimizer = FlatAdam(self.arena, lr=args.lr, betas=(0.9, 0.999), weight_decay=0.0)
            else:  # host-side tests (oracle backend)
                self.optimizer = torch.optim.Adam([self.arena.flat_param], lr=args.lr, betas=(0.9, 0.999), weight_decay=0.0)
        else:
            params = list(model.parameters())
            self.arena = FlatArena(params)
            if self.device.type == "cuda":
                from .optim import FlatSGD
                self.optimizer = FlatSGD(self.arena, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
            else:
                self.optimizer = torch.optim.SGD(params=[self.arena.flat_param], lr=args.lr, momentum=args.momentum,
                                                 weight_decay=args.weight_decay)
        self.params = self.arena.params
        self.bucket = self.arena if (self.world > 1 or FORCE_EXCHANGES) else None
        # SGD over all parameters exchanges 100 MB (TANet) per step: the arena is cut at bottleneck-block boundaries into
        # `grad_buckets` slices, each all-reduced from INSIDE the backward as soon as its blocks are done (reverse layer
        # order), so the exchange runs beside the remaining data / weight gradient launches; the affine-only arena (0.2 MB)
        # stays one all-reduce.  VITTA_GRAD_BUCKETS=1: one monolithic all-reduce after the backward.
        self.grad_buckets = int(os.environ.get("VITTA_GRAD_BUCKETS", "4")) if not getattr(args, "update_only_bn_affine", False) else 1
        self._bucket_plan = None
        self._armed = None          # {release signal: arena range} while a backward may release buckets
        self._armed_runner = None   # the trunk runner whose after_block callback is set meanwhile
        self._launched = set()      # signals whose bucket has left during the running backward
        self.dp_graph = "eager"     # how the data-parallel step is replayed: "one" graph | "segments" | "eager"
        self.n_from_backward = 0  # bucket reductions launched from inside a backward so far (tests)

        self.n_clips = _n_clips(args)
        self.if_pred_consistency = args.if_pred_consistency if args.if_sample_tta_aug_views else False
        self.n_views = args.test_crops * (args.n_augmented_views if args.if_sample_tta_aug_views else self.n_clips)
        self._graph = None
        self._side_stream = None
        self.engine = None
        if args.stat_reg == "BNS":
            # regularise the BN INPUT statistics towards the layer's own running statistics (basics.py:588-599)
            from .bns_utils import BNFeatureHook
            self.chosen_layers = choose_layers(model, list(self.bn_types))
            self.hooked = select_hooked(args, self.chosen_layers)
            backend = BACKEND_FACTORY() if (engine_backend is None and BACKEND_FACTORY is not None) else engine_backend
            self.stat_reg_hooks = [BNFeatureHook(layer, reg_type=args.reg_type, running_manner=args.running_manner,
                                                 use_src_stat_in_reg=args.use_src_stat_in_reg, momentum=args.momentum_bns,
                                                 backend=backend) for _, _, layer in self.hooked]
            return
        if args.stat_reg != "mean_var":
            raise Exception(f"undefined regularization type {args.stat_reg}")
        if isinstance(args.stat_type, str):
            raise NotImplementedError("args.stat_type of str is deprecated, use list instead.")
        means, vars_ = load_source_statistics(args, self.chosen_layers)
        if use_engine is None:
            use_engine = bool(args.moving_avg)
        if engine_backend is None and BACKEND_FACTORY is not None:
            engine_backend = BACKEND_FACTORY()
        self.backend = engine_backend
        self.engine = StatAlignEngine(args.reg_type, args.momentum_mvg, backend=engine_backend,
                                      distributed=True if FORCE_EXCHANGES else None) if use_engine else None
        if self.device.type == "cuda":  # what the step needs zeroed besides the gradients joins the arena's one fill per step
            if self.engine is not None:
                self.engine.zero_pool = self.arena
            base = getattr(self._net(), "base_model", None)
            if args.arch == "tanet" and base is not None:
                from . import trunk
                r = trunk.runner_of(base)
                r.zero_pool, r._pool_zeroed = self.arena, {}
        self.hooked = select_hooked(args, self.chosen_layers)
        self.stat_reg_hooks = [
            CombineNormStatsRegHook_onereg(layer, clip_len=args.clip_length,
                                           spatiotemp_stats_clean_tuple=(means[i], vars_[i]), reg_type=args.reg_type,
                                           moving_avg=args.moving_avg, momentum=args.momentum_mvg,
                                           stat_type_list=args.stat_type, reduce_dim=args.reduce_dim,
                                           before_norm=args.before_norm,
                                           if_sample_tta_aug_views=args.if_sample_tta_aug_views,
                                           n_augmented_views=args.n_augmented_views, engine=self.engine,
                                           backend=engine_backend)
            for i, _, layer in self.hooked]
        self._graph = None  # (adapt graph, eval graph, static buffers) once capture_graphs() ran
        self.n_clips = _n_clips(args)
        self.if_pred_consistency = args.if_pred_consistency if args.if_sample_tta_aug_views else False
        self.n_views = args.test_crops * (args.n_augmented_views if args.if_sample_tta_aug_views else self.n_clips)

    # -- modes --------------------------------------------------------------------------------
    def set_adapt_mode(self):
        """model.train() with every BatchNorm back in eval() when fix_BNS (basics.py:606-611): dropout
        and drop-path stay ACTIVE during the adaptation forward."""
        self.model.train()
        if self.args.fix_BNS:
            for m in self.model.modules():
                if isinstance(m, self.bn_types):
                    m.eval()

    def close_hooks(self):
        for h in self.stat_reg_hooks:
            h.close()

    def add_hooks_back(self):
        for h, (_, _, layer) in zip(self.stat_reg_hooks, self.hooked):
            h.add_hook_back(layer)

    # -- steps ---------------------------------------------------------------------------------
    def shape_tta_input(self, input):
        a = self.args
        if a.arch == "tanet":
            bz = input.shape[0]
            input = input.view(-1, 3, input.size(2), input.size(3))
            return input.view(bz * self.n_views, a.clip_length, 3, input.size(2), input.size(3))
        return input

    def shape_eval_input(self, input):
        a = self.args
        if a.arch == "tanet":
            bz = input.shape[0]
            input = input.view(-1, 3, input.size(2), input.size(3))
            return input.view(bz * a.test_crops * self.n_clips, a.clip_length, 3, input.size(2), input.size(3))
        return input

    def forward_losses(self, input, actual_bz):
        """Adaptation forward: (video logits, loss_reg, loss_consis or None)."""
        output, loss_consis = self.forward_local(input, actual_bz)
        if self.engine is not None:
            self.engine.exchange()
            loss_reg = self.engine.finish_global(tie=None if self.if_pred_consistency else output)
        else:
            loss_reg = torch.zeros((), dtype=torch.float32, device=output.device)
            for h in self.stat_reg_hooks:
                loss_reg = loss_reg + h.r_feature.to(output.device)
        return output, loss_reg, loss_consis

    def forward_local(self, input, actual_bz):
        """Everything of the adaptation forward that needs no communication: model forward, view
        consistency, this rank's additive moments."""
        a = self.args
        loss_consis = None
        if a.arch == "tanet":
            fused = self._fused_head(input, actual_bz)
            if fused is not None:
                output, loss_consis = fused
                if not self.if_pred_consistency:
                    loss_consis = None
            else:
                output = self.model(input).reshape(actual_bz, self.n_views, -1)
                if self.if_pred_consistency:
                    loss_consis = compute_pred_consis(output)
                output = output.mean(1)
        else:
            output, view_cls_score = self.model(input)
            if self.if_pred_consistency:
                loss_consis = compute_pred_consis(view_cls_score)
        if self.engine is not None:
            self.engine.reduce_local()
        return output, loss_consis

    def _fused_head(self, input, actual_bz):
        """(video logits, loss_consis) through ops.TanetHead when the model is our TSN on the hand-written trunk with its stock head
        (tanet.TSN.fused_head_ok): the head of the adaptation pass as dropout + ONE launch forward and ONE backward instead of
        fourteen (VITTA_FUSED_HEAD=0: the module chain).  None: not applicable, the caller takes the module chain."""
        if not FUSED_HEAD and self.device.type == "cuda":
            from ._lib import loud_once
            loud_once("fused_head_off", "VITTA_FUSED_HEAD=0 (an A/B switch): the adaptation head runs as the module chain (ATen / library launches)")
        if not FUSED_HEAD or self.device.type != "cuda" or not torch.is_grad_enabled():
            return None
        net = self._net()
        if not (hasattr(net, "fused_head_ok") and net.fused_head_ok()) or self.model is not net and (
                self.model._forward_hooks or self.model._forward_pre_hooks):
            return None
        from . import fused_bn, ops
        if not fused_bn.ENABLED:
            return None
        T = net.num_segments
        frames = input.view((-1, 3 * net.new_length) + input.size()[-2:]).shape[0]
        if frames != actual_bz * self.n_views * T or not ops.tanet_head_supported(
                torch.empty(0, net.new_fc.in_features, dtype=torch.float32, device=self.device), net.new_fc, actual_bz, self.n_views):
            return None
        feat = net.trunk_features(input)
        if feat is None:
            return None
        fc = net.base_model.fc
        return ops.TanetHead.apply(feat, net.new_fc.weight, net.new_fc.bias, float(fc.p), bool(fc.training), T, self.n_views)

    def _fused_eval_head(self, input, bz, nv):
        """Video logits [bz, K] of an EVALUATION pass through the same forward launch as the adaptation head (new_fc -> segment consensus
        -> mean over the nv crops x clips; no dropout in eval()): global pool + ONE launch instead of pool + linear + two reductions.
        None: not applicable (the caller takes the module chain)."""
        if not FUSED_HEAD or self.device.type != "cuda" or torch.is_grad_enabled():
            return None
        net = self._net()
        if not (hasattr(net, "fused_head_ok") and net.fused_head_ok()) or net.training or self.model is not net and (
                self.model._forward_hooks or self.model._forward_pre_hooks):
            return None
        from . import fused_bn, ops
        if not fused_bn.ENABLED:
            return None
        T = net.num_segments
        frames = input.view((-1, 3 * net.new_length) + input.size()[-2:]).shape[0]
        if frames != bz * nv * T or not ops.tanet_head_supported(
                torch.empty(0, net.new_fc.in_features, dtype=torch.float32, device=self.device), net.new_fc, bz, nv):
            return None
        feat = net.trunk_features(input)
        if feat is None:
            return None
        return ops.tanet_head_eval(feat, net.new_fc.weight, net.new_fc.bias, bz, nv, T)

    @staticmethod
    def _backward(loss):
        """loss.backward() from a cached unit gradient (no ones_like launch; ops.WeightedLoss.backward recognises it)."""
        if loss.is_cuda and loss.dtype == torch.float32 and loss.dim() == 0:
            from . import ops
            # (the LayerNorm passes' d gamma / d beta column sums: queued, ONE launch when the backward has been issued)
            with ops.deferred_grad_colsums():
                loss.backward(gradient=ops.unit_gradient(loss.device))
        else:
            loss.backward()

    def total_loss(self, loss_reg, loss_consis):
        a = self.args
        if self.if_pred_consistency:
            if (FUSED_HEAD and self.engine is not None and loss_reg.is_cuda and loss_reg.requires_grad and loss_consis.requires_grad
                    and loss_reg.dtype == torch.float32):
                from . import ops  # one launch forward, one backward (which leaves lambda_feature_reg * g in the engine's gscale)
                return ops.WeightedLoss.apply(loss_reg, loss_consis, a.lambda_feature_reg, a.lambda_pred_consis, self.engine.gscale)
            return a.lambda_feature_reg * loss_reg + a.lambda_pred_consis * loss_consis
        return loss_reg

    # -- overlapped schedule -----------------------------------------------------------------------
    def step(self, tta_input, eval_input, has_video=True):
        """Adaptation step on `tta_input` (video i) WHILE `eval_input` (video i-1) is evaluated on a second
        stream.  Both forward passes read the weights left by step i-1 -- exactly what the sequential order
        `adapt(i-1); eval(i-1); adapt(i)` gives them -- and the optimizer update of step i waits for the
        evaluation to finish.  Same results, but the small-grid kernels of the two passes (layer3/4 of an
        8-frame clip fill well under half of 256 CUs) share the GPU instead of queueing.
        Returns ((logits, loss_reg, loss_consis) of the adaptation, logits of the evaluation or None)."""
        if eval_input is None:
            return self.adapt_step(tta_input, has_video), None
        g = self._graph
        if (g is not None and "step" in g and has_video and tta_input.shape == g["tta_in"].shape
                and eval_input.shape == g["eval_in"].shape):
            g["tta_in"].copy_(tta_input)
            g["eval_in"].copy_(eval_input)
            self.engine.plan = g["plan"]  # the buffers the captured launches (and the eager exchanges between them) use
            if g["step"] == "split":
                cur, side = torch.cuda.current_stream(), self._side_stream
                if "pre" in g:
                    g["pre"].replay()  # trainable convolutions: ONE re-pack that both passes read
                side.wait_stream(cur)  # the previous optimizer step, the clips' copies, the re-pack
                with torch.cuda.stream(side):
                    g["eval_side"].replay()
                g["fb"].replay()
                cur.wait_stream(side)  # the evaluation still reads the weights the update overwrites
                g["opt"].replay()
            elif g["step"] is not None:
                g["step"].replay()
            else:
                g["seg_fwd"].replay()
                self.engine.exchange()
                g["seg_bwd"].replay()
                self._disarm()
                self._exchange_end(ran_backward=False)
                g["seg_opt"].replay()
            return g["adapt_out"], g["eval_out_overlapped"]
        return self._step_eager(tta_input, eval_input, has_video)

    @contextlib.contextmanager
    def _prepacked(self):
        """Trainable trunk convolutions (SGD over all parameters): rebuild their packed copies ONCE, on the current stream,
        before the evaluation forks -- both passes of the step read the same weights -- instead of once per pass."""
        runner = None
        if PREPACK and self.args.arch == "tanet" and self.device.type == "cuda":
            from . import trunk
            net = self.model.module if isinstance(self.model, SingleDeviceParallel) else self.model
            base = getattr(net, "base_model", None)
            if base is not None and trunk.ENABLED:
                runner = trunk.runner_of(base)
        if runner is None:
            yield
            return
        runner.prepacked = False
        runner.refresh_packs(self.device, adapt=True)
        runner.prepacked = runner._repack.get(True) is not None
        try:
            yield
        finally:
            runner.prepacked = False

    def _fork_eval(self, eval_input):
        """Issue the evaluation forward on the side stream (hooks closed, model.eval()); the caller joins."""
        cur = torch.cuda.current_stream()
        if self._side_stream is None:
            from . import streams
            self._side_stream = streams.role(self.device, "eval")
        side = self._side_stream
        capturing = torch.cuda.is_current_stream_capturing()
        side.wait_stream(cur)  # the previous optimizer step and the copy of the clip
        self.close_hooks()
        with torch.cuda.stream(side):
            out = self._evaluate_eager(eval_input)
        if not capturing:
            eval_input.record_stream(side)
            out.record_stream(cur)
        self.add_hooks_back()
        self.set_adapt_mode()
        return out, side

    def _forked_step(self, tta_input, eval_input, has_video=True):
        """(eval logits, adaptation outputs) with the evaluation forked on the side stream -- in front of the step, or (EVAL_FORK_BLOCK >
        0, TANet trunk) in front of that block of the adaptation forward."""
        runner = None
        if EVAL_FORK_BLOCK > 0 and has_video and self.args.arch == "tanet":
            from . import trunk
            net = self.model.module if isinstance(self.model, SingleDeviceParallel) else self.model
            base = getattr(net, "base_model", None)
            if base is not None and trunk.ENABLED and EVAL_FORK_BLOCK < len(trunk.runner_of(base).blocks()):
                runner = trunk.runner_of(base)
        if runner is None:
            ev, side = self._fork_eval(eval_input)
            return ev, self._adapt_step_eager(tta_input, has_video, join=side)
        box = {}

        def fork(i):
            if i == EVAL_FORK_BLOCK and "side" not in box:
                box["ev"], box["side"] = self._fork_eval(eval_input)

        runner.before_block = fork
        try:
            out = self._adapt_step_eager(tta_input, has_video, join=lambda: box["side"])
        finally:
            runner.before_block = None
        return box["ev"], out

    def _step_eager(self, tta_input, eval_input, has_video=True):
        if self.device.type != "cuda":  # one queue: the sequential order with the evaluation first
            self.close_hooks()
            ev = self._evaluate_eager(eval_input)
            self.add_hooks_back()
            self.set_adapt_mode()
            return self._adapt_step_eager(tta_input, has_video), ev
        with self._prepacked():
            ev, out = self._forked_step(tta_input, eval_input, has_video)
        return out, ev

    # -- gradient exchange --------------------------------------------------------------------------------------------
    def _net(self):
        return self.model.module if isinstance(self.model, SingleDeviceParallel) else self.model

    def _bucket_units(self):
        """[(module, release)] in FORWARD order: the modules at whose boundaries the gradient arena may be cut, and for each the
        index of the unit whose "backward done" signal makes its gradients final.  From the model's STRUCTURE only, so every
        rank builds the same plan whether or not it has run a step (a rank without a video must issue the same collectives).
        TANet: the bottleneck blocks of the hand-written trunk, signalled by the trunk node after each block's backward
        (release = itself).  Video Swin: every SwinTransformerBlock3D and PatchMerging, signalled by a tensor hook on its
        input (the autograd engine runs every node recorded after a tensor before the node that produced it); a block that is
        not the first of its stage has its norm1 differentiated inside the PREVIOUS block's closing residual + LayerNorm pass
        (swin.py, `next_norm`), so it is released one signal later."""
        net = self._net()
        if self.args.arch == "tanet":
            base = getattr(net, "base_model", None)
            if base is None or not hasattr(base, "layer1"):
                return None
            from . import trunk
            return [(b, i) for i, b in enumerate(trunk.runner_of(base).blocks())]
        layers = getattr(getattr(net, "backbone", None), "layers", None)
        if layers is None:
            return None
        units = []
        for layer in layers:
            for j, blk in enumerate(layer.blocks):
                units.append((blk, len(units) if j == 0 else len(units) - 1))
            if layer.downsample is not None:
                units.append((layer.downsample, len(units)))
        return units

    def bucket_plan(self):
        """buckets [(release signal, lo, hi)] in LAUNCH order (last units first) + the complement ranges reduced after the
        backward; None when the exchange stays monolithic (one rank, affine-only mode, a model without bucket units)."""
        if self.bucket is None or self.grad_buckets <= 1:
            return None
        if self._bucket_plan is not None:
            return self._bucket_plan or None
        units = self._bucket_units()
        self._bucket_plan = False  # (decided: do not walk the model again)
        if not units:
            return None
        spans = [self.arena.span(list(m.parameters())) for m, _ in units]
        if any(sp is None for sp in spans) or any(spans[i][1] != spans[i + 1][0] for i in range(len(spans) - 1)):
            return None
        total = spans[-1][1] - spans[0][0]
        plan, hi, acc = [], spans[-1][1], 0
        for i in range(len(units) - 1, -1, -1):  # walk the units the way the backward does
            acc += spans[i][1] - spans[i][0]
            if acc >= total / self.grad_buckets or i == 0:
                if plan and plan[-1][0] == units[i][1]:
                    # two cuts released by the SAME signal (a stage's second block leaves with its first block's signal) are
                    # one bucket: the armed table is keyed by signal, a second entry would overwrite the first and that range
                    # would never be reduced on an armed step -- while an un-armed rank still reduced it
                    plan[-1] = (plan[-1][0], spans[i][0], plan[-1][2])
                else:
                    plan.append((units[i][1], spans[i][0], hi))
                hi, acc = spans[i][0], 0
        n = self.arena.grad.numel()
        self._bucket_plan = dict(buckets=plan, rest=[(0, spans[0][0]), (spans[-1][1], n)], blocks=[m for m, _ in units])
        if self.args.arch != "tanet":  # tensor hooks on the inputs of the units whose signal releases a bucket
            for sig in sorted({sig for sig, _, _ in plan}):
                units[sig][0].register_forward_pre_hook(self._signal_hook(sig))
        return self._bucket_plan

    def _signal_hook(self, index):
        def pre_hook(module, args):
            x = args[0] if args else None
            if torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled():
                x.register_hook(lambda g: self._signal(index))
        return pre_hook

    def _signal(self, index):
        """The backward of unit `index` (and of everything after it) has been issued on the current stream."""
        if self._armed is None:
            return
        hit = self._armed.get(index)
        if hit is not None and index not in self._launched:
            self._launched.add(index)
            if self.device.type == "cuda":
                from . import ops
                ops.flush_grad_colsums()  # (queued d gamma / d beta sums of the units behind this cut join their bucket)
            self.arena.reduce_range(*hit, async_op=True)
            self.n_from_backward += 1

    def _disarm(self):
        """No bucket leaves from inside a backward any more; nothing of an abandoned step (a failed capture) stays pending."""
        self._armed = None
        if self._armed_runner is not None:
            self._armed_runner.after_block = None
            self._armed_runner = None

    def _exchange_begin(self):
        """Before the backward: arm the signals so that a bucket is reduced the moment its last unit's backward has been issued."""
        plan = self.bucket_plan()
        self._disarm()
        self._launched = set()
        if plan is None:
            return
        if not all(self.arena.all_direct(list(b.parameters())) for b in plan["blocks"]):
            return  # (first step, or a unit with an autograd-accumulated parameter): everything after the backward
        self._armed = {sig: (lo, hi) for sig, lo, hi in plan["buckets"]}
        if self.args.arch == "tanet":
            from . import trunk
            runner = trunk.runner_of(self._net().base_model)
            runner.after_block = self._signal
            self._armed_runner = runner

    def _exchange_end(self, ran_backward=True):
        """After the backward (+ the copy of autograd-accumulated gradients into the arena): whatever was not reduced from
        inside it, in the same order on every rank; then every pending reduction joins the stream."""
        if self.bucket is None:
            return
        plan = self.bucket_plan()
        if plan is None:
            self._disarm()
            self.bucket.all_reduce()
            return
        launched = self._launched if (self._armed is not None and ran_backward) else set()
        self._disarm()
        for sig, lo, hi in plan["buckets"]:  # (those launched from inside the backward are a prefix of this order)
            if sig not in launched:
                self.arena.reduce_range(lo, hi, async_op=True)
        for lo, hi in plan["rest"]:
            self.arena.reduce_range(lo, hi, async_op=True)
        self.arena.wait_pending()
        self._launched = set()

    def adapt_step(self, input, has_video=True):
        """One gradient step on one (already device-resident, already reshaped) TTA input.
        `has_video=False`: ragged tail of a data-parallel run -- this rank only takes part in the two
        exchanges so that EMA state and weights stay identical everywhere.
        With captured graphs (capture_graphs) a step is: copy the clip into the static buffer, replay."""
        g = self._graph
        if g is not None and has_video and input.shape == g["tta_in"].shape and "step" not in g:
            g["tta_in"].copy_(input)
            self.engine.plan = g["plan"]
            if "adapt" in g:
                g["adapt"].replay()
            else:  # data-parallel: three graph segments with the two exchanges launched eagerly in between
                g["seg_fwd"].replay()
                self.engine.exchange()
                g["seg_bwd"].replay()
                self._disarm()
                self._exchange_end(ran_backward=False)
                g["seg_opt"].replay()
            return g["adapt_out"]
        return self._adapt_step_eager(input, has_video)

    def _adapt_step_eager(self, input, has_video=True, join=None, optimizer=True):
        a = self.args
        self.arena.zero_step()
        output = loss_reg = loss_consis = None
        if has_video:
            actual_bz = input.shape[0] // self.n_views if a.arch == "tanet" else input.shape[0]
            output, loss_reg, loss_consis = self.forward_losses(input, actual_bz)
            self.arena.before_backward()
            self._exchange_begin()
            self._backward(self.total_loss(loss_reg, loss_consis))
            self.arena.after_backward()
        else:
            if self.engine is None:
                raise RuntimeError("ragged data-parallel steps need the batched engine")
            loss_reg = self.engine.finish_empty()
            self._disarm()
        self._exchange_end(ran_backward=has_video)
        if join is not None:  # an evaluation on a side stream still reads the weights this update overwrites
            torch.cuda.current_stream().wait_stream(join() if callable(join) else join)
        if optimizer:  # (False: the split-graph capture steps in a graph of its own, behind the join with the evaluation stream)
            self.optimizer.step()
        # detached: nothing the caller holds may keep this step's autograd graph (and its AccumulateGrad
        # nodes, which remember the stream they were created on) alive into a later graph capture
        det = lambda t: None if t is None else t.detach()
        return det(output), det(loss_reg), det(loss_consis)

    def evaluate(self, input):
        g = self._graph
        if g is not None and input.shape == g["eval_in"].shape:
            g["eval_in"].copy_(input)
            g["eval"].replay()
            return g["eval_out"]
        return self._evaluate_eager(input)

    def _capture_segments(self, g, overlap_eval=False):
        """Data-parallel capture: no collective inside a graph.  The step is cut at its two exchanges into
        forward | backward | optimizer segments sharing one memory pool (the backward segment walks the
        autograd graph recorded while the forward segment was captured, like make_graphed_callables)."""
        a = self.args
        x = g["tta_in"]
        actual_bz = x.shape[0] // self.n_views if a.arch == "tanet" else x.shape[0]
        pool = torch.cuda.graph_pool_handle()
        g["seg_fwd"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["seg_fwd"], pool=pool, capture_error_mode=CAPTURE_MODE):
            with self._prepacked() if overlap_eval else contextlib.nullcontext():
                if overlap_eval:  # the evaluation of the previous video runs beside the adaptation forward
                    g["eval_out_overlapped"], side = self._fork_eval(g["eval_in"])
                self.arena.zero_step()
                output, loss_consis = self.forward_local(x, actual_bz)
            if overlap_eval:
                torch.cuda.current_stream().wait_stream(side)
        g["seg_bwd"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["seg_bwd"], pool=pool, capture_error_mode=CAPTURE_MODE):
            loss_reg = self.engine.finish_global(tie=None if self.if_pred_consistency else output)
            self.arena.before_backward()
            self._backward(self.total_loss(loss_reg, loss_consis))
            self.arena.after_backward()
        g["seg_opt"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["seg_opt"], pool=pool, capture_error_mode=CAPTURE_MODE):
            self.optimizer.step()
        g["adapt_out"] = (output.detach(), loss_reg.detach(), None if loss_consis is None else loss_consis.detach())

    def capture_graphs(self, tta_input, eval_input, segmented=False, overlap_eval=False, collectives_in_graph=None, split=None):
        """Capture the adaptation step (forward, hooks, both losses, backward, optimizer) and the
        evaluation forward into two hipGraphs.  The per-video iteration is ~1500 short kernels; eagerly
        the host launch rate, not the GPU, sets the pace (r1a profile: 16 ms of kernels in a 29 ms
        step).  Requirements: a few eager steps ran before (launch plans, optimizer state and MIOpen
        solutions exist), fixed input shapes, single process (no collective inside the capture).
        Capturing records launches without executing them: model, EMA and optimizer state are untouched."""
        if self.device.type != "cuda":
            raise RuntimeError("graph capture needs a CUDA(HIP) device")
        if self.engine is None:
            raise RuntimeError("graph capture needs the batched engine (the stand-alone hooks keep host-side "
                               "EMA scalars that cannot be captured)")
        if self.engine.plan is None:
            raise RuntimeError("run at least one eager step before capturing")
        if collectives_in_graph is None:
            # Default: three segments with the two exchanges launched eagerly between them -- no collective inside a capture.
            # ONE graph with both all-reduces (and the bucketed gradient exchange inside the backward) captured measured the
            # same on a one-rank RCCL group (173.2 vs 174.1 videos/s) and has a failure mode no `except` can catch: torch's
            # process-group watchdog polls the end events of the eager collectives it still lists, and an event query on a
            # stream that is capturing is hipErrorCapturedEvent, which takes the PROCESS down (seen in about one run of four
            # before the quiesce below).  Until an N > 1 RCCL run has passed with it, the one-graph form is opt-in:
            # VITTA_GRAPH_COLLECTIVES=1 / bench.py --graph-collectives.
            collectives_in_graph = (self.bucket is not None and not segmented and torch.distributed.is_initialized()
                                    and torch.distributed.get_backend() == "nccl" and os.environ.get("VITTA_GRAPH_COLLECTIVES", "0") == "1")
        if collectives_in_graph:
            self._quiesce_collectives()
            try:
                self._capture(tta_input, eval_input, segmented, overlap_eval, True, split=False)
                self.dp_graph = "one"
                return
            except Exception as e:  # noqa: BLE001  (a capture the collectives library refuses must not cost the run)
                import warnings
                warnings.warn(f"data-parallel step not captured as one graph ({e!r}); using three segments")
                torch.cuda.synchronize()
        self._capture(tta_input, eval_input, segmented, overlap_eval, False, split=split)
        self.dp_graph = "segments" if "seg_fwd" in self._graph else ("split" if self._graph.get("step") == "split" else "one")

    def _quiesce_collectives(self):
        """Before a capture that the collectives' own stream joins: nothing of the eager steps may still be listed by the
        process group's watchdog.  Every pending Work of the arena is waited for, the device drained, the ranks meet (so no
        peer is still issuing eager collectives into a communicator whose stream is about to capture), and the watchdog is
        given three of its 100 ms polling periods to retire the completed entries (it exposes no "list empty" query)."""
        self.arena.wait_pending()
        torch.cuda.synchronize()
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()
        time.sleep(0.35)

    def _abandon_step(self):
        """Forget everything a step that did not finish (a capture that raised inside the backward) left armed or pending:
        the trunk's after_block callback, the armed bucket signals, the Work objects of reductions issued into the aborted
        capture.  Without this the fallback capture would reduce from inside its backward segment AND again eagerly."""
        self._disarm()
        self._launched = set()
        self.arena._pending = []
        self._graph = None

    def _capture_split(self, g):
        """The overlapped step of one process as separate graphs (SPLIT_GRAPHS): g["pre"] (only with trainable trunk convolutions: their
        re-pack, once for both passes), g["fb"] = fill + forward + both losses + backward, g["opt"] = the optimizer update, on the step's
        stream; g["eval_side"] = the evaluation forward, CAPTURED ON THE SIDE STREAM (everything keyed per stream -- split-K workspaces,
        arrival tickets, on-demand weight packs -- is then the side stream's own, as in the forked single graph) and replayed there.
        step() orders them: pre -> {eval_side || fb} -> join -> opt.  Same launches, same results as the forked graph; the two passes
        are independent graph launches instead of branches the graph executor schedules."""
        if self._side_stream is None:
            from . import streams
            self._side_stream = streams.role(self.device, "eval")
        side = self._side_stream
        # (the adaptation graphs are captured on torch's default capture stream -- one pool stream for every capture of the process, whose
        # per-stream buffers therefore exist after the first capture --; vitta_amd/streams.py keeps the side stream off it)
        assert side.cuda_stream != torch.cuda.graphs.graph.default_capture_stream.cuda_stream
        pre_cm = None
        runner = None
        if PREPACK and self.args.arch == "tanet":
            from . import trunk
            net = self.model.module if isinstance(self.model, SingleDeviceParallel) else self.model
            base = getattr(net, "base_model", None)
            if base is not None and trunk.ENABLED:
                runner = trunk.runner_of(base)
        try:
            if runner is not None and any(p.requires_grad for p in base.parameters() if p.dim() == 4):
                g["pre"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g["pre"], capture_error_mode=CAPTURE_MODE):
                    pre_cm = self._prepacked()
                    pre_cm.__enter__()  # (the re-pack launches land in g["pre"]; runner.prepacked stays set for the captures below)
                if not runner.prepacked:  # nothing was rebuilt (stem only): no such graph
                    del g["pre"]
            g["fb"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["fb"], capture_error_mode=CAPTURE_MODE):
                g["adapt_out"] = self._adapt_step_eager(g["tta_in"], True, optimizer=False)
            g["opt"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["opt"], capture_error_mode=CAPTURE_MODE):
                self.optimizer.step()
            self.close_hooks()
            with torch.cuda.stream(side):
                self._evaluate_eager(g["eval_in"])  # (eagerly once on the side stream: its per-stream tables exist before the capture)
            torch.cuda.synchronize()
            g["eval_side"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["eval_side"], stream=side, capture_error_mode=CAPTURE_MODE):
                g["eval_out_overlapped"] = self._evaluate_eager(g["eval_in"])
            self.add_hooks_back()
            self.set_adapt_mode()
        finally:
            if pre_cm is not None:
                pre_cm.__exit__(None, None, None)
        g["step"] = "split"

    def _capture(self, tta_input, eval_input, segmented, overlap_eval, collectives_in_graph, split=None):
        if split is None:
            split = SPLIT_GRAPHS
        self._abandon_step()
        g = {"tta_in": tta_input.clone(), "eval_in": eval_input.clone()}
        torch.cuda.synchronize()
        self.set_adapt_mode()
        # collectives_in_graph (opt-in, bench.py --graph-collectives): the two RCCL all-reduces are captured like any other
        # launch and the data-parallel step is ONE graph (no host round trip between segments); the default keeps the
        # collectives outside captures
        if (self.world > 1 or segmented or self.bucket is not None) and not collectives_in_graph:
            self._capture_segments(g, overlap_eval)
            if overlap_eval:
                g["step"] = None  # step() replays the three segments
        elif overlap_eval and split:
            self._capture_split(g)
        elif overlap_eval:
            g["step"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["step"], capture_error_mode=CAPTURE_MODE):
                with self._prepacked():
                    g["eval_out_overlapped"], g["adapt_out"] = self._forked_step(g["tta_in"], g["eval_in"], True)
        else:
            g["adapt"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["adapt"], capture_error_mode=CAPTURE_MODE):
                g["adapt_out"] = self._adapt_step_eager(g["tta_in"], True)
        self.close_hooks()
        self._evaluate_eager(g["eval_in"])  # eagerly once: what the stand-alone evaluation builds on first use (weight tables of
        torch.cuda.synchronize()            # its own pass) must exist before the capture -- the steps so far may all have ridden along
        g["eval"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["eval"], capture_error_mode=CAPTURE_MODE):
            g["eval_out"] = self._evaluate_eager(g["eval_in"])
        self.add_hooks_back()
        torch.cuda.synchronize()
        g["plan"] = self.engine.plan  # an eager step on other shapes in between may switch the engine to another plan
        self._graph = g

    @torch.no_grad()
    def _evaluate_eager(self, input):
        self.model.eval()
        a = self.args
        if a.arch == "tanet":
            nv = a.test_crops * self.n_clips
            bz = input.shape[0] // nv
            out = self._fused_eval_head(input, bz, nv)
            if out is not None:
                return out
            out = self.model(input).reshape(bz, nv, -1)
            return out[:, 0] if nv == 1 else out.mean(1)  # (a mean over one element is still a launch)
        output, _ = self.model(input)
        return output


# ------------------------------------------------------------------------------------------------
# the online TTA loop
# ------------------------------------------------------------------------------------------------
def tta_standard(model_origin, criterion, args=None, logger=None, writer=None):
    """'tta_online': one gradient step per video, evaluate that video right after, keep the model.
    'tta_standard': re-initialise model/optimizer/hooks for every video (momentum_mvg must be 1)."""
    if args.if_tta_standard == "tta_standard":
        assert args.momentum_mvg == 1.0
        assert args.n_epoch_adapat == 1
    elif args.if_tta_standard == "tta_online":
        assert args.momentum_mvg != 1.0
        assert args.n_gradient_steps == 1
        assert args.n_epoch_adapat == 1
    if not hasattr(args, "moving_avg"):
        args.moving_avg = False
    if not hasattr(args, "momentum_mvg"):
        args.momentum_mvg = 0.1

    device = _device_of(model_origin)
    rank, world = _dist()
    if args.arch == "tanet":
        tta_set = get_dataset_tanet(args, split="val", dataset_type="tta")
        eval_set = get_dataset_tanet(args, split="val", dataset_type="eval")
    elif args.arch == "videoswintransformer":
        tta_set = get_dataset_videoswin(args, split="val", dataset_type="tta")
        eval_set = get_dataset_videoswin(args, split="val", dataset_type="eval")
    else:
        raise NotImplementedError(f"Incorrect model type {args.arch}")
    n_total = len(tta_set)
    if world > 1:
        # rank r adapts videos R*k + r; batch_size videos per rank and step
        mine = list(range(rank, n_total, world))
        tta_set = torch.utils.data.Subset(tta_set, mine)
        eval_set = torch.utils.data.Subset(eval_set, mine)
        per_rank = -(-n_total // world)  # ceil
        n_steps = -(-per_rank // args.batch_size)
    tta_loader, eval_loader = _loader(tta_set, args), _loader(eval_set, args)
    if world == 1:
        n_steps = len(tta_loader)
    # (extension, --prefetch_input, default on: the next video's host -> device copies run on a copy stream beside the current step;
    # the reference's loop uploads with a blocking .cuda() in front of each step, basics.py:612-623)
    from .prefetch import DevicePrefetcher
    pf = device.type == "cuda" and bool(getattr(args, "prefetch_input", True))
    tta_iter = DevicePrefetcher(tta_loader, device) if pf else iter(tta_loader)
    eval_iter = DevicePrefetcher(eval_loader, device) if pf else iter(eval_loader)
    ahead = (lambda it: it.ahead()) if pf else (lambda it: None)

    log = DeferredLog(device, logger, n_steps, args.verbose)
    adapter = None
    # overlapped schedule (extension, --overlap_eval): the evaluation of video i runs beside the adaptation forward
    # of video i+1 on a second stream -- same weights, same numbers, one iteration later (ViTTAAdapter.step)
    # Only with frozen BatchNorm buffers: under --fix_BNS False the adaptation forward of video i+1 UPDATES running_mean /
    # running_var while the evaluation of video i would read them on the other stream (a race, and a different protocol:
    # the reference evaluates video i before adapt(i+1) touches the buffers).  LayerNorm models have no such buffers.
    bn_buffers_frozen = bool(args.fix_BNS) or not any(isinstance(m, nn.modules.batchnorm._BatchNorm)
                                                      for m in model_origin.modules())
    overlap = (bool(getattr(args, "overlap_eval", True)) and device.type == "cuda" and bn_buffers_frozen
               and args.if_tta_standard == "tta_online" and args.n_gradient_steps == 1)
    waiting = None  # (batch_id, row, actual_bz, ev_input, ev_target): adapted, not evaluated yet
    end = time.time()

    def finish(item, ev_output, now):
        batch_id_, row_, bz_, _, ev_target_ = item
        prec1, prec5 = accuracy(ev_output.data, ev_target_, topk=(1, 5))
        row_[3], row_[4] = prec1, prec5
        log.push(batch_id_, row_, bz_, now - end)

    for batch_id in range(n_steps):
        try:
            input, target = next(tta_iter)
            has_video = True
        except StopIteration:
            input = target = None
            has_video = False
        if adapter is None or args.if_tta_standard == "tta_standard":
            print(f"Batch {batch_id}, initialize the model, update chosen layers, initialize hooks, intialize average meter")
            adapter = ViTTAAdapter(model_origin, args)
        if (adapter._graph is None and GRAPH_AFTER_STEPS is not None and batch_id == GRAPH_AFTER_STEPS and has_video
                and getattr(args, "hip_graph", True) and device.type == "cuda" and adapter.engine is not None
                and args.if_tta_standard == "tta_online" and args.n_gradient_steps == 1):
            ev0 = eval_set[0][0].unsqueeze(0).expand(input.shape[0], *eval_set[0][0].shape)
            adapter.capture_graphs(adapter.shape_tta_input(input.to(device)), adapter.shape_eval_input(ev0.to(device)),
                                   overlap_eval=overlap)
        adapter.set_adapt_mode()
        row = torch.zeros(DeferredLog.FIELDS, dtype=torch.float32, device=device)
        actual_bz = 0
        if has_video:
            actual_bz = input.shape[0]
            input = adapter.shape_tta_input(input.to(device, non_blocking=True))
            target = target.to(device, non_blocking=True)
        ev_output = None
        if overlap:
            (output, loss_reg, loss_consis), ev_output = adapter.step(input, waiting[3] if waiting else None, has_video)
        else:
            for _ in range(args.n_gradient_steps):
                output, loss_reg, loss_consis = adapter.adapt_step(input, has_video)
        if overlap:
            ahead(tta_iter)  # the step is issued: the next video's upload goes out beside it
        if has_video:
            row[0] = loss_reg.detach()
            if loss_consis is not None:
                row[1] = loss_consis.detach()
            row[2] = criterion(output.detach(), target)  # logging only, never part of the loss (basics.py:657)
        now = time.time()
        if overlap:
            if waiting is not None:
                finish(waiting, ev_output, now)
                waiting = None
            if has_video:
                ev_input, ev_target = next(eval_iter)
                waiting = (batch_id, row, actual_bz, adapter.shape_eval_input(ev_input.to(device, non_blocking=True)),
                           ev_target.to(device, non_blocking=True))
                ahead(eval_iter)
            end = now
            continue
        adapter.close_hooks()
        if has_video:
            ev_input, ev_target = next(eval_iter)
            ev_input = adapter.shape_eval_input(ev_input.to(device, non_blocking=True))
            ev_target = ev_target.to(device, non_blocking=True)
            output = adapter.evaluate(ev_input)
            # sequential schedule: the host pulls the next video (loader decode, pinning: both block the host) only now, with this
            # video's adaptation AND evaluation already issued -- a loader-bound run keeps the GPU busy meanwhile
            ahead(tta_iter)
            ahead(eval_iter)
            prec1, prec5 = accuracy(output.data, ev_target, topk=(1, 5))
            row[3], row[4] = prec1, prec5
        else:
            ahead(tta_iter)
        if args.if_tta_standard == "tta_online":
            adapter.add_hooks_back()
        now = time.time()
        if has_video:
            log.push(batch_id, row, actual_bz, now - end)
        end = now
    if waiting is not None:  # drain: the last video's evaluation has nothing to overlap with
        adapter.close_hooks()
        finish(waiting, adapter.evaluate(waiting[3]), time.time())
        adapter.add_hooks_back()
    log.flush()
    top1 = log.meters["top1"]
    if world > 1:
        t = torch.tensor([top1.sum, top1.count], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t)
        return [float(t[0] / t[1])]
    return [top1.avg]


# ------------------------------------------------------------------------------------------------
# the epoch-style variant (if_tta_standard falsy): adapt over the whole list, then evaluate the whole list
# ------------------------------------------------------------------------------------------------
def _shard(dataset, n_total, rank, world, batch_size):
    """Rank r of R takes videos r, R+r, ...; returns (dataset, steps every rank must take)."""
    if world == 1:
        return dataset, -(-n_total // batch_size)
    mine = list(range(rank, n_total, world))
    per_rank = -(-n_total // world)
    return torch.utils.data.Subset(dataset, mine), -(-per_rank // batch_size)


def test_time_adapt(model, criterion, args=None, logger=None, writer=None):
    """corpus/basics.py:760-1084: one pass of gradient steps over the test list (`batch_size` videos per step, the
    caller's model adapted in place, statistics hooks closed afterwards), then `validate_brief` over the same list
    with the adapted weights.  Returns ([top-1 after the epoch], model).

    Same per-step arithmetic as the online loop (the adapter's adapt step); what differs is the protocol: no per-video
    evaluation, accuracy meters of the adaptation pass use the TRAIN-mode view-averaged logits (basics.py:1036), the
    eval loader batches `batch_size_eval` videos.  The reference's other branches of this function are not reachable
    as shipped: `stat_reg='BNS'` fails on an undefined name at basics.py:1015, a second epoch backpropagates through
    the closed hooks' stale `r_feature` graphs (basics.py:1013-1028 after :1065-1067), `cossim` is outside the ViTTA
    path."""
    if args.stat_reg != "mean_var":
        raise NotImplementedError(f"test_time_adapt: stat_reg={args.stat_reg!r} is not a working branch of the reference")
    if int(args.n_epoch_adapat) != 1:
        raise NotImplementedError("test_time_adapt: the reference cannot run a second epoch (hooks are closed after the first)")
    if not hasattr(args, "moving_avg"):
        args.moving_avg = False
    if not hasattr(args, "momentum_mvg"):
        args.momentum_mvg = 0.1
    device = _device_of(model)
    rank, world = _dist()
    if args.arch == "tanet":
        tta_set = get_dataset_tanet(args, split="val", dataset_type="tta")
        eval_set = get_dataset_tanet(args, split="val", dataset_type="eval")
    elif args.arch == "videoswintransformer":
        tta_set = get_dataset_videoswin(args, split="val", dataset_type="tta")
        eval_set = get_dataset_videoswin(args, split="val", dataset_type="eval")
    else:
        raise NotImplementedError(f"Incorrect model type {args.arch}")
    tta_set, n_steps = _shard(tta_set, len(tta_set), rank, world, args.batch_size)
    adapter = ViTTAAdapter(model, args, copy=False)
    if_sample = args.if_sample_tta_aug_views
    if if_sample:
        assert adapter.n_clips == 1

    epoch = 0
    log = DeferredLog(device, logger, n_steps, args.verbose, epoch=epoch, print_freq=args.print_freq)
    tta_iter = iter(_loader(tta_set, args))
    end = time.time()
    for i in range(n_steps):
        try:
            input, target = next(tta_iter)
            has_video = True
        except StopIteration:  # ragged tail of a data-parallel run
            input = target = None
            has_video = False
        adapter.set_adapt_mode()
        row = torch.zeros(DeferredLog.FIELDS, dtype=torch.float32, device=device)
        actual_bz = 0
        if has_video:
            actual_bz = input.shape[0]
            input = adapter.shape_tta_input(input.to(device, non_blocking=True))
            target = target.to(device, non_blocking=True)
        output, loss_reg, loss_consis = adapter.adapt_step(input, has_video)
        now = time.time()
        if has_video:
            row[0] = loss_reg
            if loss_consis is not None:
                row[1] = loss_consis
            row[2] = criterion(output, target)
            prec1, prec5 = accuracy(output.data, target, topk=(1, 5))
            row[3], row[4] = prec1, prec5
            log.push(i, row, actual_bz, now - end)
            if writer is not None:
                writer.add_scalars("loss", {"loss_reg": float(loss_reg)}, global_step=i + 1)
                if loss_consis is not None:
                    writer.add_scalars("loss", {"loss_consis": float(loss_consis)}, global_step=i + 1)
                writer.add_scalars("loss", {"loss_ce": float(row[2])}, global_step=i + 1)
        end = now
    log.flush()
    adapter.close_hooks()
    top1_acc = validate_brief(eval_set, adapter, global_iter=n_steps, epoch=epoch, args=args, logger=logger, writer=writer)
    return [top1_acc], adapter.model


def validate_brief(eval_set, adapter, global_iter, epoch=None, args=None, logger=None, writer=None):
    """corpus/basics.py:1105-1186: top-1 of the (adapted) model over the evaluation views of the whole list,
    `batch_size_eval` videos at a time; data-parallel runs split the list and add up the counts."""
    device = adapter.device
    rank, world = _dist()
    bz_eval = getattr(args, "batch_size_eval", args.batch_size)
    eval_set, _ = _shard(eval_set, len(eval_set), rank, world, bz_eval)
    workers = 0 if getattr(eval_set, "on_device", False) else args.workers
    loader = torch.utils.data.DataLoader(eval_set, batch_size=bz_eval, shuffle=False, num_workers=workers,
                                         pin_memory=workers > 0)
    batch_time, top1, top5 = AverageMeter(), AverageMeter(), AverageMeter()
    counts = torch.zeros(3, dtype=torch.float64, device=device)  # sum prec1*bz, sum prec5*bz, videos
    with torch.no_grad():
        end = time.time()
        for i, (input, target) in enumerate(loader):
            actual_bz = input.shape[0]
            input = adapter.shape_eval_input(input.to(device, non_blocking=True))
            target = target.to(device, non_blocking=True)
            output = adapter._evaluate_eager(input)
            prec1, prec5 = accuracy(output.data, target, topk=(1, 5))
            counts += torch.stack([prec1.reshape(()) * actual_bz, prec5.reshape(()) * actual_bz,
                                   torch.full_like(prec1.reshape(()), actual_bz)]).double()
            if args.verbose and i % args.print_freq == 0:  # the only place the host waits for the stream
                top1.update(prec1.item(), actual_bz)
                top5.update(prec5.item(), actual_bz)
                batch_time.update(time.time() - end)
                logger.debug(("  \tTest Epoch {epoch}: [{0}/{1}]\t"
                              "Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                              "Prec@1 {top1.val:.3f} ({top1.avg:.3f})\t"
                              "Prec@5 {top5.val:.3f} ({top5.avg:.3f})").format(
                    i, len(loader), epoch=epoch, batch_time=batch_time, top1=top1, top5=top5))
            end = time.time()
    if world > 1:
        torch.distributed.all_reduce(counts)
    s1, s5, n = counts.tolist()
    acc1, acc5 = s1 / max(n, 1.0), s5 / max(n, 1.0)
    if logger is not None:
        logger.debug("  \tTesting Results Epoch {epoch}: Prec@1 {0:.3f} Prec@5 {1:.3f}".format(acc1, acc5, epoch=epoch))
        logger.debug(f"  \tTest Epoch {epoch} acc {acc1} ")
    if writer is not None:
        writer.add_scalars("acc", {"test_acc": acc1}, global_step=global_iter)
    return acc1


# ------------------------------------------------------------------------------------------------
# source-only validation (BASELINE config 0: runs on the host CPU as well)
# ------------------------------------------------------------------------------------------------
def validate(val_loader, model, criterion, iter, epoch=None, args=None, logger=None, writer=None, optimizer=None):
    batch_time, losses, top1, top5 = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    n_clips = _n_clips(args)
    device = _device_of(model)
    if getattr(args, "evaluate_baselines", False) and getattr(args, "baseline", None) == "source":
        logger.debug(f"Starting ---- {getattr(args, 'corruptions', None)} ---- evaluation for Source...")
    with torch.no_grad():
        end = time.time()
        for i, (input, target) in enumerate(val_loader):
            model.eval()
            actual_bz = input.shape[0]
            input, target = input.to(device), target.to(device)
            if args.arch == "tanet":
                input = input.view(-1, 3, input.size(2), input.size(3))
                input = input.view(actual_bz * args.test_crops * n_clips, args.clip_length, 3, input.size(2), input.size(3))
                output = model(input).reshape(actual_bz, args.test_crops * n_clips, -1).mean(1)
            elif args.arch == "videoswintransformer":
                output, _ = model(input)
            else:
                raise NotImplementedError(f"Incorrect model type {args.arch}")
            loss = criterion(output, target)
            prec1, prec5 = accuracy(output.data, target, topk=(1, 5))
            losses.update(loss.item(), actual_bz)
            top1.update(prec1.item(), actual_bz)
            top5.update(prec5.item(), actual_bz)
            batch_time.update(time.time() - end)
            end = time.time()
            if args.verbose and i % args.print_freq == 0:
                logger.debug(("Test: [{0}/{1}]\t"
                              "Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                              "Loss {loss.val:.4f} ({loss.avg:.4f})\t"
                              "Prec@1 {top1.val:.3f} ({top1.avg:.3f})\t"
                              "Prec@5 {top5.val:.3f} ({top5.avg:.3f})").format(
                    i, len(val_loader), batch_time=batch_time, loss=losses, top1=top1, top5=top5))
    logger.debug("Testing Results: Prec@1 {top1.avg:.3f} Prec@5 {top5.avg:.3f} Loss {loss.avg:.5f}".format(
        top1=top1, top5=top5, loss=losses))
    logger.debug(f"Validation acc {top1.avg} ")
    return top1.avg


# ------------------------------------------------------------------------------------------------
# source-statistics producer
# ------------------------------------------------------------------------------------------------
def compute_statistics(model=None, args=None, logger=None, log_time=None):
    """Batch-size-weighted average of per-batch means and per-batch biased variances over the clean
    training videos, one entry per BN2d/3d (TANet) or LayerNorm[1:] (Swin); written as two object
    arrays list_<stat_type>_{mean,var}_<log_time>.npy (corpus/basics.py:220-307)."""
    if args.stat_type != "spatiotemp":
        raise NotImplementedError("only stat_type 'spatiotemp' is on the ViTTA path")
    if args.arch == "tanet":
        chosen_layers = choose_layers(model, [nn.BatchNorm2d, nn.BatchNorm3d])
        dataset = get_dataset_tanet(args, split="val", dataset_type="eval")
    elif args.arch == "videoswintransformer":
        chosen_layers = choose_layers(model, [nn.LayerNorm])[1:]
        dataset = get_dataset_videoswin(args, split="val", dataset_type="eval")
    else:
        raise NotImplementedError(f"Incorrect model type {args.arch}")
    backend = BACKEND_FACTORY() if BACKEND_FACTORY is not None else None
    hooks = [ComputeNormStatsHook(layer, clip_len=args.clip_length, stat_type=args.stat_type,
                                  before_norm=args.before_norm, batch_size=args.batch_size, backend=backend)
             for _, layer in chosen_layers]
    device = _device_of(model)
    n_clips = _n_clips(args)
    loader = _loader(dataset, args)
    sum_mean = [None] * len(hooks)
    sum_var = [None] * len(hooks)
    count = 0
    model.eval()
    with torch.no_grad():
        for batch_id, (input, _) in enumerate(loader):
            actual_bz = input.shape[0]
            input = input.to(device)
            if args.arch == "tanet":
                input = input.view(-1, 3, input.size(2), input.size(3))
                input = input.view(actual_bz * args.test_crops * n_clips, args.clip_length, 3, input.size(2), input.size(3))
            model(input)
            if batch_id % 1000 == 0:
                print(f"{batch_id}/{len(loader)} batches completed ...")
            for k, h in enumerate(hooks):
                sum_mean[k] = h.batch_mean * actual_bz if sum_mean[k] is None else sum_mean[k] + h.batch_mean * actual_bz
                sum_var[k] = h.batch_var * actual_bz if sum_var[k] is None else sum_var[k] + h.batch_var * actual_bz
            count += actual_bz
    for h in hooks:
        h.close()
    means = np.empty(len(hooks), dtype=object)
    vars_ = np.empty(len(hooks), dtype=object)
    for k in range(len(hooks)):
        means[k] = (sum_mean[k] / count).cpu().numpy()
        vars_[k] = (sum_var[k] / count).cpu().numpy()
    np.save(osp.join(args.result_dir, f"list_{args.stat_type}_mean_{log_time}.npy"), means, allow_pickle=True)
    np.save(osp.join(args.result_dir, f"list_{args.stat_type}_var_{log_time}.npy"), vars_, allow_pickle=True)
    return means, vars_
