"""The TANet trunk (ResNet-50 + TAM, models/tanet_models/tanet.py:125-150, temporal_module.py:12-140) on the hand-written
convolutions: every activation lives as channel-major planes [C, N*H*W] from the stem's max-pool to the global average
pooling, every convolution is `vitta_conv_f32` (vitta_amd/csrc/conv.hip), and BatchNorm / residual / ReLU / the hooked
layers' moments ride in the convolutions' epilogues and prologues.

One bottleneck (temporal_module.py:85-106), forward:
    conv1 (1x1)            -> x1 RAW                          [+ moments of bn1(x1) when hooked]
    TAM                    reads x1, applies relu(bn1(.)) on load -> a1                         (tam_cm.hip)
    conv2 (3x3, stride s)  -> x2 RAW, a2 = relu(bn2(x2))      [+ moments of bn2(x2)]
    downsample (block 0)   -> xd RAW, zd = bn_d(xd)           [+ moments of zd]
    conv3 (1x1)            reads a2 -> x3 RAW, out = relu(bn3(x3) + identity)   [+ moments of bn3(x3)]
backward (G = gradient w.r.t. out, all consumers summed):
    bn_bwd  : dz3 = G [out > 0] + inj3 -> dx3, g_id = G [out > 0], d gamma3 / d beta3
    dgrad conv3, epilogue = BatchNorm(+ReLU) backward of bn2 -> dx2, d gamma2 / d beta2
    dgrad conv2 -> d a1 ; TAM backward ; bn_bwd of bn1 (+ the pooling gradient) -> dx1
    (block 0) bn_bwd of bn_d on g_id -> dxd ; dgrad downsample (half resolution when strided) -> gd
    dgrad conv1, epilogue adds g_id (or gd at the even positions) -> G of the previous block
`inj` = the statistics-loss gradient gscale (a_c + b_c (z - mu_c)) of a hooked layer.

The whole trunk is ONE autograd node (TrunkFunction): parameter gradients go straight into their `.grad` storage
(ops._grad_sink), nothing in between is visible to autograd.  Eligibility (`TrunkRunner.eligible`): CUDA fp32 input, every
block a TemporalBottleneck, every BatchNorm in eval mode, and no forward hook other than the engine-bound statistics hooks
-- anything else takes the module-by-module path of resnet.py / tanet.py.  Convolution weights may be frozen (the packed
copies are cached per weight version) or TRAINABLE (SGD over all parameters, the reference's default optimizer,
corpus/basics.py:547-560): then they are re-packed in every forward (an optimizer that updates weights through its own
kernel does not bump tensor versions), the backward adds their gradients with `vitta_conv_wgrad_f32` /
`vitta_stem_conv7_wgrad_f32` (the stem falls back to torch modules in front of the node only for output widths the tiled
stem pass does not cover).
"""
import ctypes as C
import logging
import os
import sys

import torch
import torch.nn as nn

from . import _lib, conv as CV
from ._lib import check, lib

ENABLED = True


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# (Round 3 measured the identity / downsample path of a stage's first bottleneck on a helper stream beside the main chain: 6.00 ms
# per step against 5.88 -- with launches that fill the chip the event edges cost more than the overlap returns; removed in round 4.)
# weight gradients (SGD over all parameters) on a helper stream beside the data-gradient chain: nothing in the backward waits
# for them, so a block's three or four wgrad + reduce launches run while the main stream is already in the next block
WGRAD_SIDE = os.environ.get("VITTA_TRUNK_WGRAD_STREAM", "1") != "0"
# TAM's adaptive average pooling of relu(bn1(conv1(x))) taken from conv1's accumulators (VITTA_CONV_POOL: frame-major [N][T][C]
# fixed-point sums -- order-independent --, which the branch kernels read with pooled_tc = 1) instead of a pass over x1: 16 launches fewer per pass; "0": the
# stand-alone pooling launch (tests hold the two forms together)
POOL_FOLD = os.environ.get("VITTA_TRUNK_POOL_FOLD", "1") != "0"
# bn3 (+ identity add + ReLU) backward of block i inside the epilogue of block i+1's conv1 data gradient -- the launch that produces
# the gradient w.r.t. block i's output -- instead of its own pass (VITTA_CONV_BWD_BN | VITTA_CONV_BWD_RELU | VITTA_CONV_RES with
# bwd_x = x3, bwd_mask = out, y_raw = the masked gradient that feeds the identity path): 15 launches and two streams of a
# 4p-channel tensor fewer per backward.  (Measured slower on the round-2 fp32 kernels, 79.8 vs 57.7 us + the pass; re-measured on
# conv_b3.hip in round 5.)  "0": the stand-alone pass.
BN3_FOLD = os.environ.get("VITTA_TRUNK_BN3_FOLD", "1") != "0"
_side_streams = {}
_side_pool = {}


class _Fork:
    """A helper stream that waits for the work queued so far on the current stream, and that the current stream waits for at
    `join()` (the weight-gradient launches of SGD over all parameters beside the data-gradient chain, WGRAD_SIDE).  Works the
    same eagerly and under hipGraph capture (the helper stream is pulled into the capture by the first wait and leaves it
    at the join).  Tensors the helper's kernels touch are allocated by the caller on the main stream and outlive the join."""

    def __init__(self, device, role=0):
        self.main = torch.cuda.current_stream(device)
        key = (device.index, self.main.cuda_stream, role)
        self.side = _side_streams.get(key)
        if self.side is None:  # helper streams are created ahead, outside any capture; here one is only assigned
            pool = _side_pool.get(device.index)
            if pool is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("vitta_amd.trunk: run one eager step before capturing (helper streams are created eagerly)")
                from . import streams
                pool = _side_pool[device.index] = streams.roles(device, "trunk_helper", 8)  # distinct from every other role's
            used = sum(1 for k in _side_streams if k[0] == device.index)
            self.side = _side_streams[key] = pool[used % len(pool)]
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)

    def __enter__(self):
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)

    def join(self):
        ev = torch.cuda.Event()
        ev.record(self.side)
        self.main.wait_event(ev)


_sync_bufs = {}


def _sync(device):
    """Meeting counters of the fused TAM branch launches on the CURRENT stream (zero at rest; one buffer per stream: launches
    on different streams may overlap)."""
    return CV.zeroed_per_stream(_sync_bufs, device, 1024, spares=4)  # (never zero-filled inside a graph capture)


def _bn_ptrs(bn):
    arr = (C.c_void_p * 4)()
    for i, t in enumerate((bn.weight, bn.bias, bn.running_mean, bn.running_var)):
        arr[i] = t.data_ptr()
    return arr


def _bn_t(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var)


def _producer_hooks(bn):
    """The ComputeNormStatsHook objects (source-statistics producer, utils/norm_stats_utils.py:18-101) on a BatchNorm2d whose
    moments the node can supply from a convolution epilogue: product backend, spatio-temporal statistics."""
    from .norm_stats import ComputeNormStatsHook, HipBackend
    out = []
    for fn in bn._forward_hooks.values():
        owner = getattr(fn, "__self__", None)
        if isinstance(owner, ComputeNormStatsHook) and isinstance(owner.backend, HipBackend) and owner.stat_type == "spatiotemp":
            out.append(owner)
    return out


def _engine_hook(bn):
    """(ok, hook): the engine-bound statistics hook of a BatchNorm2d, or None; ok False if the module carries any forward
    hook besides that one and the producer hooks of _producer_hooks (those need the module-by-module path)."""
    hook = None
    producers = _producer_hooks(bn)
    for fn in bn._forward_hooks.values():
        owner = getattr(fn, "__self__", None)
        if owner is not None and any(owner is p for p in producers):
            continue
        if (owner is not None and getattr(owner, "engine", None) is not None and hasattr(owner, "index")
                and getattr(owner, "kind", None) == "bn2d" and hook is None):
            hook = owner
        else:
            return False, None
    # (a before_norm hook of the batched engine collects through its own forward PRE-hook: the node reads the raw output instead)
    pre_ok = all(hook is not None and getattr(fn, "__self__", None) is hook for fn in bn._forward_pre_hooks.values())
    return pre_ok, hook


def _noop_hooks_only(module):
    for fn in module._forward_hooks.values():
        owner = getattr(fn, "__self__", None)
        if owner is None or getattr(owner, "kind", None) != "bn1d":
            return False
    return not module._forward_pre_hooks


def _sflags(site):
    """Statistics flags of a convolution launch that feeds `site` (None: no statistics)."""
    if site is None:
        return 0
    return CV.CONV_STATS | (CV.CONV_STATS_RAW if site.raw else 0)


class Site:
    """A hooked BatchNorm2d of this step: where the convolution epilogue deposits the additive statistics and which
    coefficient slices its backward injects.  raw (before_norm hooks, utils/norm_stats_utils.py:185: the hooked feature is the
    BatchNorm's INPUT): the statistics are those of the raw convolution output (VITTA_CONV_STATS_RAW) and the statistics-loss
    gradient joins the gradient w.r.t. that output (VITTA_CONV_INJ_RAW / VITTA_BN_BWD_INJ_RAW)."""

    def __init__(self, engine, plan, index, raw=False):
        self.raw = bool(raw)
        sl = plan.channel_slice(index)
        self.stats = (engine.src_mean[sl], plan.s1[sl], plan.s2[sl])
        self.inj = (plan.mu[sl], plan.coef_a[sl], plan.coef_b[sl], engine.gscale)


class ProducerSite:
    """A BatchNorm2d carrying ComputeNormStatsHook objects: the convolution epilogue deposits sum(f - k), sum (f - k)^2 of the
    hooked feature f and finish() turns them into the hooks' batch_mean / batch_var.  f = z = bn(conv output) with k = the
    BN's bias, or -- before_norm hooks, utils/norm_stats_utils.py:52-53 -- f = the RAW convolution output (the BN's input)
    with k = its running mean (VITTA_CONV_STATS_RAW): taken directly as the reference does, so a zero / tiny gamma is just
    another channel (round 3 inverted the affine map and divided by gamma)."""
    inj = None

    def __init__(self, bn, hooks, device):
        c = bn.num_features
        self.bn, self.hooks = bn, hooks
        self.raw = bool(hooks[0].before_norm)
        if any(bool(h.before_norm) != self.raw for h in hooks):
            raise RuntimeError("vitta_amd.trunk: one BatchNorm2d carries before_norm and after-norm producer hooks")
        self.s = torch.zeros(2, c, dtype=torch.float32, device=device)
        self.shift = (bn.running_mean if self.raw else bn.bias).detach()
        self.stats = (self.shift, self.s[0], self.s[1])

    def finish(self, count):
        d1 = self.s[0].double() / count
        mean = (self.shift.double() + d1).float()
        var = (self.s[1].double() / count - d1 * d1).clamp_(min=0.0).float()
        for h in self.hooks:
            h.batch_mean, h.batch_var = mean, var


class TrunkRunner:
    def __init__(self, resnet):
        self.net = resnet
        self._packed = {}
        self._step_packs = {}
        self._repack = {}
        self._geo = {}
        self._wgrad_pending = []
        # tta.py, for the span of one overlapped step: the adaptation set of packs was rebuilt on the main stream BEFORE the
        # evaluation stream forked and the weights do not change until both passes are done -- neither pass re-packs, the
        # evaluation reads the adaptation set's forward packs
        self.prepacked = False
        self.after_block = None  # callable(block index) run after each block's backward (tta: bucketed gradient exchange)
        self.before_block = None  # callable(block index) run before each block's forward of an adaptation pass (tta: delayed evaluation fork)
        # tta.FlatArena of the adapter driving this trunk (or None): the adaptation pass's fixed-point pooling sums then live in the
        # arena's zeroed tail -- the step's ONE fill covers them (FlatArena.Zeroed.fresh() tells whether it has, else zeroed here)
        self.zero_pool = None
        self._pool_zeroed = {}

    # -- structure ---------------------------------------------------------------------------------------------
    def blocks(self):
        out = []
        for name in ("layer1", "layer2", "layer3", "layer4"):
            out.extend(getattr(self.net, name).children())
        return out

    def bn2d_modules(self):
        """Every BatchNorm2d of the trunk the runner evaluates, stem first."""
        mods = [self.net.bn1]
        for b in self.blocks():
            mods += [b.net.bn1, b.net.bn2, b.net.bn3]
            if b.net.downsample is not None:
                mods.append(b.net.downsample[1])
        return mods

    def why_not(self, x):
        """None when the hand-written trunk takes this pass, else the reason it declines (logged by trunk.run)."""
        from . import fused_bn, ops
        from .tanet import TemporalBottleneck
        net = self.net
        if not (ENABLED and fused_bn.ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            return "trunk / fused BN passes disabled, or the clip is not a CUDA float32 [N, 3, H, W] tensor"
        if x.requires_grad:  # the node hands no gradient back to the clip
            return "the input clip requires a gradient (the trunk node hands none back)"
        # pixel counts of every stage must be multiples of four (16-byte epilogue accesses): frames x H x W after the stem,
        # after each stride-2 stage
        hh, ww = CV.out_size(CV.out_size(x.shape[2], 7, 2, 3), 3, 2, 1), CV.out_size(CV.out_size(x.shape[3], 7, 2, 3), 3, 2, 1)
        for _ in range(4):
            if (x.shape[0] * hh * ww) % 4:
                return "frames x H x W of a stage is not a multiple of four (16-byte epilogue accesses)"
            hh, ww = CV.out_size(hh, 3, 2, 1), CV.out_size(ww, 3, 2, 1)
        grad = torch.is_grad_enabled()
        mp, c1 = net.maxpool, net.conv1
        if net._forward_hooks or net._forward_pre_hooks or not isinstance(mp, nn.MaxPool2d) or mp._forward_hooks \
                or (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) != (3, 2, 1, 1, False) \
                or net.relu._forward_hooks or c1._forward_hooks or c1.bias is not None \
                or (tuple(c1.weight.shape), c1.stride, c1.padding, c1.dilation, c1.groups) != ((64, 3, 7, 7), (2, 2), (3, 3), (1, 1), 1) \
                or x.shape[3] % 4 \
                or not isinstance(net.bn1, nn.BatchNorm2d) or not net.bn1.affine \
                or x.shape[0] * 64 > 65535:
            return "the stem is not the stock hook-free 7x7/2 convolution + eval-mode affine BatchNorm2d + 3/2/1 max-pool (or the width is not a multiple of four / too many frames)"
        if net.bn1.training:
            return "a BatchNorm2d is in TRAIN mode (--fix_BNS False, or a calibration pass) or has no affine parameters"
        blocks = self.blocks()
        if not blocks:
            return "no residual blocks"
        t = blocks[0].n_segment if isinstance(blocks[0], TemporalBottleneck) else 0
        if t <= 0 or x.shape[0] % t:
            return "the blocks are not TemporalBottlenecks or the frame count is not a multiple of the segment count"
        engines = set()
        for b in blocks:
            if not isinstance(b, TemporalBottleneck) or b.n_segment != t or b._forward_hooks or b.net._forward_hooks or b.tam._forward_hooks:
                return "a block is not a hook-free TemporalBottleneck with the common segment count"
            n = b.net
            convs = [n.conv1, n.conv2, n.conv3]
            bns = [n.bn1, n.bn2, n.bn3]
            if n.downsample is not None:
                ds = n.downsample
                if not (isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d)
                        and isinstance(ds[1], nn.BatchNorm2d) and not ds._forward_hooks
                        and ds[0].kernel_size == (1, 1) and ds[0].padding == (0, 0) and ds[0].stride == n.conv2.stride):
                    return "a downsample branch is not Conv2d 1x1 + BatchNorm2d with conv2's stride"
                convs.append(ds[0])
                bns.append(ds[1])
            for cv in convs:
                if cv.bias is not None or cv._forward_hooks or cv._forward_pre_hooks \
                        or cv.groups != 1 or cv.dilation != (1, 1):
                    return "a convolution has a bias, a hook, groups or dilation"
            if n.conv1.padding != (0, 0) or n.conv3.padding != (0, 0) or n.conv1.in_channels % 16 or n.conv1.out_channels % 32 \
                    or (n.conv1.kernel_size, n.conv1.stride) != ((1, 1), (1, 1)) or (n.conv3.kernel_size, n.conv3.stride) != ((1, 1), (1, 1)) \
                    or n.conv2.kernel_size != (3, 3) or n.conv2.padding != (1, 1) or n.conv2.stride[0] != n.conv2.stride[1] \
                    or n.conv2.stride[0] not in (1, 2) or n.relu._forward_hooks:
                return "a bottleneck's convolution geometry is not 1x1 / 3x3 (stride 1 | 2) / 1x1 with 16- / 32-aligned channels"
            for bn in bns:
                if not isinstance(bn, nn.BatchNorm2d) or bn.training or not bn.affine:
                    return "a BatchNorm2d is in TRAIN mode (--fix_BNS False, or a calibration pass) or has no affine parameters"
                ok, hook = _engine_hook(bn)
                if not ok or (hook is not None and not grad):
                    return "a BatchNorm2d carries a foreign forward hook (e.g. stat_reg='BNS' BNFeatureHook), or an engine hook under no_grad"
                if hook is not None:
                    engines.add(id(hook.engine))
            tam = b.tam
            bg, bl = tam.G[1], tam.L[1]
            if bg.training or bl.training or not _noop_hooks_only(bg) or not _noop_hooks_only(bl) \
                    or not ops.tam_branch_supported(n.conv1.out_channels, t) or x.shape[0] // t > 32:
                return "a TAM BatchNorm1d is in train mode or hooked, the TAM width / segment count is unsupported, or more than 32 clips per pass"
            for m in (tam.G[0], tam.G[3], tam.L[0], tam.L[3]):
                if m._forward_hooks:
                    return "a TAM convolution / linear layer carries a hook"
        ok, hook = _engine_hook(net.bn1)
        if not ok or hook is not None:  # an ENGINE hook on the stem BN takes the module path (the shipped configuration hooks
            return "the stem BatchNorm2d carries a foreign or engine hook"                # layer3/4); producer hooks get the stem's moments from its raw output
        for bn in [net.bn1] + [m for b in blocks for m in ([b.net.bn1, b.net.bn2, b.net.bn3] + ([b.net.downsample[1]] if b.net.downsample is not None else []))]:
            if len({bool(h.before_norm) for h in _producer_hooks(bn)}) > 1:
                return "a BatchNorm2d carries statistics-producer hooks on both its input and its output"  # one statistics site per BatchNorm2d: its input OR its output
        return None if len(engines) <= 1 else "the hooked layers belong to more than one statistics engine"

    def eligible(self, x):
        return self.why_not(x) is None

    # -- caches ------------------------------------------------------------------------------------------------
    def packed(self, conv, kind, adapt=None):
        """Packed weight of `conv` (kind "f" forward, "b" data gradient: conv.Pack with the split-bf16 image where the shape
        qualifies; "s": the stem's [148][64] array).
        adapt: which re-packed set of a trainable weight -- True: the adaptation pass's (forward AND backward name it: the
        grad mode is off inside TrunkFunction.forward and inside autograd's backward, so it cannot tell the passes apart),
        False / None: the evaluation pass's, which another stream may be re-packing and which holds no backward packs."""
        w = conv.weight
        if w.requires_grad and kind in ("f", "b"):  # trainable: this pass's copies, rebuilt by ONE launch per forward
            st = self._repack.get(bool(adapt))
            if st is not None and (id(w), kind) in st["packs"]:
                return st["packs"][(id(w), kind)]
        if w.requires_grad:  # (stem, or outside a refreshed forward): re-packed on demand
            key = (id(w), kind, torch.cuda.current_stream(w.device).cuda_stream)  # per stream: evaluation runs beside adaptation
            hit = self._step_packs.get(key)
            if hit is None:
                with torch.no_grad():
                    hit = self._step_packs[key] = CV.pack_stem(w) if kind == "s" else CV.make_pack(CV.pack_fwd(w) if kind == "f" else CV.pack_bwd(w))
            return hit
        key = (id(w), kind, CV.ARITH)
        tag = (w.data_ptr(), w._version, tuple(w.shape))
        hit = self._packed.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = (tag, CV.pack_stem(w) if kind == "s" else CV.make_pack(CV.pack_fwd(w) if kind == "f" else CV.pack_bwd(w)))
            self._packed[key] = hit
        return hit[1]

    def refresh_packs(self, device, adapt=False):
        """Trainable convolution weights: rebuild the packed copies (`vitta_conv_repack_f32`, one launch over all of them, and
        `vitta_conv_pack_b3_table`, one launch for their split-bf16 images).
        Two sets, created once (before any graph capture: the tables are host -> device copies): one for passes under
        autograd (adaptation: `adapt`, with the data-gradient packs), one for the evaluation pass, which may run beside it on
        a second stream.  The caller names the pass: TrunkFunction.forward runs with grad mode off like the evaluation."""
        convs = []
        for b in self.blocks():
            convs += [b.net.conv1, b.net.conv2, b.net.conv3] + ([b.net.downsample[0]] if b.net.downsample is not None else [])
        convs = [c for c in convs if c.weight.requires_grad]
        if not convs:
            return
        key = bool(adapt)
        if self.prepacked:
            if self._repack.get(True) is None:
                raise RuntimeError("vitta_amd.trunk: prepacked without a rebuilt adaptation set")
            self._repack[key] = self._repack[True]
            return
        st = self._repack.get((key, CV.ARITH))
        sig = tuple((c.weight.data_ptr(), tuple(c.weight.shape)) for c in convs)
        if st is None or st["sig"] != sig:
            import numpy as np
            dt = np.dtype([("src", "<u8"), ("fwd", "<u8"), ("bwd", "<u8"), ("first", "<i8"), ("K", "<i4"), ("C", "<i4"), ("taps", "<i4"),
                           ("pad", "<i4")])
            dt3 = np.dtype([("src", "<u8"), ("dst", "<u8"), ("first", "<i8"), ("taps", "<i4"), ("R", "<i4"), ("O", "<i4"), ("pad", "<i4")])
            tab = np.zeros(len(convs), dtype=dt)
            packs, first, b3rows, units = {}, 0, [], 0
            for i, c in enumerate(convs):
                w = c.weight
                k, ci, kh, kw = w.shape
                taps = kh * kw
                pf = torch.empty(taps, ci, k, dtype=torch.float32, device=device)
                # the data-gradient pack only where a backward can follow (the no_grad set of the evaluation pass skips it)
                pb = (torch.empty(taps, k, ci, dtype=torch.float32, device=device) if key else None) if taps > 1 \
                    else w.detach().view(1, k, ci)
                for kind, pk in (("f", pf), ("b", pb)):
                    if pk is None:
                        packs[(id(w), kind)] = None
                        continue
                    b3 = None
                    if CV.ARITH == "b3" and CV.b3_eligible(pk) and (kind == "f" or key):
                        b3 = torch.empty(pk.numel() * 6, dtype=torch.uint8, device=device)
                        b3rows.append((pk.data_ptr(), b3.data_ptr(), units, pk.shape[0], pk.shape[1], pk.shape[2], 0))
                        units += pk.numel() // 8
                    packs[(id(w), kind)] = CV.Pack(pk, b3)
                tab[i] = (w.data_ptr(), pf.data_ptr(), pb.data_ptr() if (taps > 1 and pb is not None) else 0, first, k, ci, taps, 0)
                first += k * ci * taps
            dtab = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
            dtab3 = torch.from_numpy(np.array(b3rows, dtype=dt3).view(np.uint8).copy()).to(device) if b3rows else None
            st = self._repack[(key, CV.ARITH)] = dict(sig=sig, packs=packs, table=dtab, n=len(convs), total=first, table3=dtab3, n3=len(b3rows),
                                                      units=units)
        self._repack[key] = st
        check(lib().vitta_conv_repack_f32(_p(st["table"]), st["n"], st["total"], _stream()), "vitta_conv_repack_f32")
        if st["table3"] is not None:
            check(lib().vitta_conv_pack_b3_table(_p(st["table3"]), st["n3"], st["units"], _stream()), "vitta_conv_pack_b3_table")

    def geo(self, kind, n, h, w, k=1, stride=1, pad=0):
        key = (kind, n, h, w, k, stride, pad)
        g = self._geo.get(key)
        if g is None:
            g = CV.Geometry.forward(n, h, w, k, stride, pad) if kind == "f" else \
                CV.Geometry.dgrad_merged(n, h, w, k, stride, pad) if kind == "bm" else CV.Geometry.dgrad(n, h, w, k, stride, pad)
            self._geo[key] = g
        return g

    def merged_ok(self, geom, pack, c):
        key = (id(geom), c, CV.ARITH, getattr(pack, "b3", None) is not None)
        hit = self._geo.get(key)
        if hit is None:
            hit = self._geo[key] = CV.merged_dgrad_supported(geom, pack, c, c)
        return hit

    # -- statistics sites --------------------------------------------------------------------------------------
    def open_producer_sites(self, x):
        """{id(bn): ProducerSite} for the BatchNorm2d modules of the blocks that carry source-statistics producer hooks."""
        out = {}
        for bn in self.bn2d_modules()[1:]:
            hooks = _producer_hooks(bn)
            if hooks:
                out[id(bn)] = ProducerSite(bn, hooks, x.device)
        return out

    def open_sites(self, x):
        """{bn module: Site} for this step; the engine is switched to direct deposit (conv epilogues add into [s1 | s2])."""
        hooked, engine = [], None
        for bn in self.bn2d_modules():
            _, hook = _engine_hook(bn)
            if hook is not None:
                hooked.append((bn, hook))
                engine = hook.engine
        if engine is None:
            return {}
        if len(hooked) != len(engine.hooks):
            raise RuntimeError("statistics hooks outside the trunk share its engine: use the module path")
        shapes = self.feature_shapes(x)
        by_index = sorted(hooked, key=lambda p: p[1].index)
        plan = engine.begin_direct([shapes[id(bn)] for bn, _ in by_index], x.device)
        return {id(bn): Site(engine, plan, hook.index, hook.before_norm) for bn, hook in hooked}

    def feature_shapes(self, x):
        """{id(bn): (frames, C, HW, NCHW)} of every BatchNorm2d output for input x (what a hook would see)."""
        n = x.shape[0]
        h, w = CV.out_size(x.shape[2], 7, 2, 3), CV.out_size(x.shape[3], 7, 2, 3)
        out = {id(self.net.bn1): (n, 64, h * w, _lib.LAYOUT_NCHW)}
        h, w = CV.out_size(h, 3, 2, 1), CV.out_size(w, 3, 2, 1)
        for b in self.blocks():
            net = b.net
            p, s = net.conv1.out_channels, net.conv2.stride[0]
            out[id(net.bn1)] = (n, p, h * w, _lib.LAYOUT_NCHW)
            ho, wo = CV.out_size(h, 3, s, 1), CV.out_size(w, 3, s, 1)
            out[id(net.bn2)] = (n, p, ho * wo, _lib.LAYOUT_NCHW)
            out[id(net.bn3)] = (n, 4 * p, ho * wo, _lib.LAYOUT_NCHW)
            if net.downsample is not None:
                out[id(net.downsample[1])] = (n, 4 * p, ho * wo, _lib.LAYOUT_NCHW)
            h, w = ho, wo
        return out

    # -- forward -----------------------------------------------------------------------------------------------
    def stem(self, x):
        """conv1 -> bn1 -> relu -> maxpool: (raw convolution output NCHW, pooled NCHW): stem_conv.hip, then the BN + ReLU +
        max-pool pass of stem.hip."""
        from .ops import _ptr4
        net, bn = self.net, self.net.bn1
        y = CV.stem_conv(x.contiguous(), self.packed(net.conv1, "s"))
        n, c, h, w = y.shape
        ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        if w % 4 == 0 and w <= 256:  # the tiled pass writes the trunk's channel-major planes directly (no transposing copy)
            cm = torch.empty(c, n * ph * pw, dtype=torch.float32, device=y.device)
            check(lib().vitta_stem_bn_relu_pool_fwd_cm_f32(_p(y), _ptr4(bn.weight, bn.bias, bn.running_mean, bn.running_var), float(bn.eps),
                                                           n, c, h, w, _p(cm), _stream()), "vitta_stem_bn_relu_pool_fwd_cm_f32")
            return y, cm, (n, ph, pw)
        pooled = torch.empty(n, c, ph, pw, dtype=torch.float32, device=y.device)
        check(lib().vitta_stem_bn_relu_pool_fwd_f32(_p(y), _ptr4(bn.weight, bn.bias, bn.running_mean, bn.running_var), float(bn.eps),
                                                    n, c, h, w, _p(pooled), _stream()), "vitta_stem_bn_relu_pool_fwd_f32")
        return y, CV.to_cm(pooled), (n, ph, pw)

    def stem_producer(self, y, hooks):
        from . import ops
        bn = self.net.bn1
        if y is None:
            raise RuntimeError("statistics-producer hooks on the stem BatchNorm need the stem inside the node")
        mean_x, var_x = ops.moments(y, "bn2d")
        scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
        for h in hooks:
            if h.before_norm:
                h.batch_mean, h.batch_var = mean_x, var_x
            else:
                h.batch_mean = (bn.bias.detach().double() + (mean_x.double() - bn.running_mean.double()) * scale).float()
                h.batch_var = (var_x.double() * scale * scale).float()

    def stem_backward(self, y, gcm, dims, sink, x=None):
        """gcm: the gradient w.r.t. the pooled stem output as channel-major planes [64, n * ph * pw] (dims = (n, ph, pw))."""
        from .ops import _ptr4
        bn = self.net.bn1
        dw, db = sink(bn.weight), sink(bn.bias)
        conv_w = self.net.conv1.weight
        trainable = conv_w.requires_grad and x is not None
        if y.shape[3] % 4 == 0 and y.shape[3] <= 256 and (trainable or dw is not None or db is not None):
            dw = dw if dw is not None else torch.zeros_like(bn.weight)
            db = db if db is not None else torch.zeros_like(bn.bias)
            n, c, h, w = y.shape
            dy = torch.zeros_like(y) if trainable else None
            check(lib().vitta_stem_bn_relu_pool_bwd_cm_f32(_p(y), _p(gcm), _ptr4(bn.weight, bn.bias, bn.running_mean, bn.running_var),
                                                           float(bn.eps), n, c, h, w, _p(dw), _p(db), _p(dy), _stream()),
                  "vitta_stem_bn_relu_pool_bwd_cm_f32")
            if trainable:
                CV.stem_wgrad(x, dy, sink(conv_w))
            return
        gpool = CV.from_cm(gcm, *dims)
        if conv_w.requires_grad and x is not None:
            # trainable stem convolution: the same pass also scatters the gradient w.r.t. the convolution output, then
            # vitta_stem_conv7_wgrad_f32 turns it into the weight gradient (the clip itself needs no gradient)
            dw = dw if dw is not None else torch.zeros_like(bn.weight)
            db = db if db is not None else torch.zeros_like(bn.bias)
            n, c, h, w = y.shape
            dy = torch.zeros_like(y)
            check(lib().vitta_stem_bn_relu_pool_bwd_f32(_p(y), _p(gpool.contiguous()), _ptr4(bn.weight, bn.bias, bn.running_mean, bn.running_var),
                                                        float(bn.eps), n, c, h, w, _p(dw), _p(db), _p(dy), _stream()),
                  "vitta_stem_bn_relu_pool_bwd_f32")
            CV.stem_wgrad(x, dy, sink(conv_w))
            return
        if dw is None and db is None:
            return
        if dw is None or db is None:  # the kernel writes both
            dw = dw if dw is not None else torch.zeros_like(bn.weight)
            db = db if db is not None else torch.zeros_like(bn.bias)
        n, c, h, w = y.shape
        check(lib().vitta_stem_bn_relu_pool_bwd_affine_f32(_p(y), _p(gpool.contiguous()),
                                                           _ptr4(bn.weight, bn.bias, bn.running_mean, bn.running_var), float(bn.eps),
                                                           n, c, h, w, _p(dw), _p(db), _stream()),
              "vitta_stem_bn_relu_pool_bwd_affine_f32")

    def block_forward(self, b, xin, n, h, w, keep, sites, pool=None):
        """pool: [zeroed int64 buffer, next offset] of the pass in progress (POOL_FOLD) -- a local of forward(), not runner state."""
        net, tam = b.net, b.tam
        t = b.n_segment
        nb = n // t
        cin, p, s = net.conv1.in_channels, net.conv1.out_channels, net.conv2.stride[0]
        dev = xin.device
        f = dict(dtype=torch.float32, device=dev)
        L = lib()
        st = _stream()
        P = n * h * w
        s1, s2, s3 = sites.get(id(net.bn1)), sites.get(id(net.bn2)), sites.get(id(net.bn3))
        # identity path (first bottleneck of a stage: 1x1 convolution + BN, beside the main chain)
        xd = None
        if net.downsample is not None:
            dconv, dbn = net.downsample[0], net.downsample[1]
            sd = sites.get(id(dbn))
            gdn = self.geo("f", n, h, w, 1, dconv.stride[0], 0)
            Pd = n * gdn.hy * gdn.wy
            ident = torch.empty(4 * p, Pd, **f)
            xd = torch.empty(4 * p, Pd, **f) if keep else None
            wd = self.packed(dconv, "f", keep)

            def identity_path():
                CV.launch(gdn, xin, wd, ident, cin, 4 * p, flags=CV.CONV_EPI_APPLY | _sflags(sd), y_raw=xd,
                          epi_bn=_bn_t(dbn), eps=dbn.eps, stats=sd.stats if sd else None)
            identity_path()
        else:
            ident = xin
        # conv1 -> x1 raw
        x1 = torch.empty(p, P, **f)
        pooled, ptc = None, 0
        if pool is not None and h * w >= 32 and CV.ARITH == "b3":  # this block's [frames, p] piece of the pass's zeroed pooling
            # buffer (the exact-fp32 kernels keep the pooling launch: only their tile kernel carries the epilogue)
            buf, off = pool
            if off + n * p > buf.numel():
                raise RuntimeError("vitta_amd.trunk: the pass's pooling buffer is smaller than its blocks need")
            pooled, ptc = buf[off:off + n * p].view(nb, t, p), 1
            pool[1] = off + n * p
        CV.launch(self.geo("f", n, h, w), xin, self.packed(net.conv1, "f", keep), x1, cin, p, flags=_sflags(s1),
                  epi_bn=_bn_t(net.bn1) if (s1 or pooled is not None) else None, eps=net.bn1.eps, stats=s1.stats if s1 else None,
                  pool=pooled)
        # TAM on relu(bn1(x1))
        bn1p = _bn_ptrs(net.bn1)
        if pooled is None:
            pooled = torch.empty(nb, p, t, **f)
            check(L.vitta_tam_pool_cm_f32(_p(x1), bn1p, float(net.bn1.eps), p, nb, t, h * w, _p(pooled), st), "vitta_tam_pool_cm_f32")
        bg, bl = tam.G[1], tam.L[1]
        kern, gate, hpre = torch.empty(nb * p, 3, **f), torch.empty(nb, p, t, **f), torch.empty(2, nb, p // 4, t, **f)
        from .ops import _ptr4
        from .ops import tam_branch_fused_supported
        if tam_branch_fused_supported(nb, p, t):  # every workgroup of the one-launch form resident (also beside the other stream's)
            check(L.vitta_tam_branch_fwd_fused_f32(_p(pooled), _p(tam.G[0].weight), _ptr4(bg.weight, bg.bias, bg.running_mean, bg.running_var),
                                                   float(bg.eps), _p(tam.G[3].weight), _p(tam.L[0].weight),
                                                   _ptr4(bl.weight, bl.bias, bl.running_mean, bl.running_var), float(bl.eps),
                                                   _p(tam.L[3].weight), nb, p, t, _p(kern), _p(gate), _p(hpre), _p(_sync(dev)), ptc, st),
                  "vitta_tam_branch_fwd_fused_f32")
        else:  # two launches, no device-side meeting point
            check(L.vitta_tam_branch_fwd_f32(_p(pooled), _p(tam.G[0].weight), _ptr4(bg.weight, bg.bias, bg.running_mean, bg.running_var),
                                             float(bg.eps), _p(tam.G[3].weight), _p(tam.L[0].weight),
                                             _ptr4(bl.weight, bl.bias, bl.running_mean, bl.running_var), float(bl.eps),
                                             _p(tam.L[3].weight), nb, p, t, _p(kern), _p(gate), _p(hpre), ptc, st), "vitta_tam_branch_fwd_f32")
        a1 = torch.empty(p, P, **f)
        check(L.vitta_tam_agg_fwd_cm_f32(_p(x1), bn1p, float(net.bn1.eps), _p(gate), _p(kern), p, nb, t, h * w, _p(a1), st),
              "vitta_tam_agg_fwd_cm_f32")
        # conv2 -> a2 = relu(bn2(x2)) (+ x2 raw for the backward): the activation is applied ONCE in this epilogue -- as a
        # prologue of conv3 it sits in the slab loop (two VALU instructions per staged element beside the MFMAs: the
        # 1024 -> 256 pointwise launch 43 us against 26 us plain, tools/debug/conv_epilogue_probe.py)
        g2 = self.geo("f", n, h, w, 3, s, 1)
        ho, wo = g2.hy, g2.wy
        Po = n * ho * wo
        a2 = torch.empty(p, Po, **f)
        x2 = torch.empty(p, Po, **f) if keep else None
        CV.launch(g2, a1, self.packed(net.conv2, "f", keep), a2, p, p,
                  flags=CV.CONV_EPI_APPLY | CV.CONV_EPI_RELU | _sflags(s2), y_raw=x2,
                  epi_bn=_bn_t(net.bn2), eps=net.bn2.eps, stats=s2.stats if s2 else None)
        # conv3 -> x3 raw, out
        out = torch.empty(4 * p, Po, **f)
        x3 = torch.empty(4 * p, Po, **f) if keep else None
        CV.launch(self.geo("f", n, ho, wo), a2, self.packed(net.conv3, "f", keep), out, p, 4 * p,
                  flags=CV.CONV_EPI_APPLY | CV.CONV_EPI_RELU | CV.CONV_RES | _sflags(s3),
                  y_raw=x3, res=ident, epi_bn=_bn_t(net.bn3), eps=net.bn3.eps, stats=s3.stats if s3 else None)
        saved = None
        if keep:
            saved = dict(xin=xin, x1=x1, pooled=pooled, ptc=ptc, kern=kern, gate=gate, hpre=hpre, x2=x2, x3=x3, out=out, xd=xd,
                         a1=a1 if net.conv2.weight.requires_grad else None,
                         a2=a2 if net.conv3.weight.requires_grad else None, dims=(n, h, w, ho, wo))
        return out, ho, wo, saved

    def forward(self, x, keep, pooled_in=None):
        """x [N, 3, H, W] -> (features [N, 2048], tape).  keep: save what the backward needs.  pooled_in: the stem's output
        [N, 64, h, w] computed outside (trainable stem convolution); the tape then ends at it."""
        cur_stream = torch.cuda.current_stream(x.device).cuda_stream
        self._step_packs = {k: v for k, v in self._step_packs.items() if k[2] != cur_stream}  # this stream's packs are stale
        self.refresh_packs(x.device, adapt=keep)
        self._adapt_pass = keep
        sites = self.open_sites(x) if keep else {}
        producers = self.open_producer_sites(x)
        if producers:
            if any(k in sites for k in producers):
                raise RuntimeError("a BatchNorm2d carries both an engine hook and a statistics-producer hook: use the module path")
            sites = {**sites, **producers}
        if pooled_in is None:
            y, cur, (n, h, w) = self.stem(x)  # raw 7x7 output; the max-pooled planes, channel-major
        else:
            y = None
            n, _, h, w = pooled_in.shape
            cur = CV.to_cm(pooled_in.contiguous())
        h0w0 = (h, w)
        tape = []
        blocks = self.blocks()
        # one zeroed buffer for the pooled means of every block of this pass (conv1's epilogue ADDS into it)
        pool = None
        if POOL_FOLD:
            npool, buf = n * sum(b.net.conv1.out_channels for b in blocks), None
            if keep and self.zero_pool is not None:  # the adaptation pass: a slice of the arena tail that the step's ONE fill zeroes
                if npool not in self._pool_zeroed:  # (a fresh slice of the tail is zero: nothing has written there)
                    self._pool_zeroed[npool] = self.zero_pool.reserve_zeroed(npool * 8)
                z = self._pool_zeroed[npool]
                if z is not None and z.tensor.device == x.device:
                    buf = z.tensor.view(torch.int64)[:npool]
                    if not z.fresh():
                        buf.zero_()
            if buf is None:
                buf = torch.zeros(npool, dtype=torch.int64, device=x.device)
            pool = [buf, 0]
        for i, b in enumerate(blocks):
            if keep and self.before_block is not None:
                self.before_block(i)
            cur, h, w, saved = self.block_forward(b, cur, n, h, w, keep, sites, pool)
            tape.append(saved)
        c = cur.shape[0]
        feat = torch.empty(n, c, dtype=torch.float32, device=x.device)
        check(lib().vitta_avgpool_cm_f32(_p(cur), c, n, h * w, _p(feat), _stream()), "vitta_avgpool_cm_f32")
        if producers:  # source-statistics producer: hand every hook its batch moments
            shapes = self.feature_shapes(x)
            for site in producers.values():
                frames, _, hw, _ = shapes[id(site.bn)]
                site.finish(frames * hw)
            stem_hooks = _producer_hooks(self.net.bn1)
            if stem_hooks:  # the stem's BN output is never materialised: moments of the raw 7x7 output, mapped through the BN
                self.stem_producer(y, stem_hooks)
            sites = {k: v for k, v in sites.items() if k not in producers}
        return feat, dict(tape=tape, sites=sites, stem=y if (keep and pooled_in is None) else None,
                          x=x if (keep and pooled_in is None and self.net.conv1.weight.requires_grad) else None, pooled_hw=h0w0,
                          last=(c, n, h, w))

    # -- backward ----------------------------------------------------------------------------------------------
    def block_backward(self, b, sv, G, sites, sink, prev=None):
        """G [4p, Po] = gradient w.r.t. the block output (all consumers) -> gradient w.r.t. the block input; or G = (dx3, g_id):
        the block's bn3 + add + ReLU backward already applied by the launch that produced the gradient (BN3_FOLD).
        prev = (block, tape entry) of the block in FRONT of this one: its bn3 backward rides in the epilogue of this block's last
        launch, and the pair (dx3, g_id) of THAT block is returned instead of the plain gradient."""
        net, tam = b.net, b.tam
        t = b.n_segment
        n, h, w, ho, wo = sv["dims"]
        nb = n // t
        wq = []  # this block's weight-gradient launches (argument tuples keep their tensors alive until the helper is joined)

        def wgrad(*a, **kw):
            if WGRAD_SIDE:
                wq.append((a, kw))
            else:
                CV.wgrad(*a, **kw)
        cin, p, s = net.conv1.in_channels, net.conv1.out_channels, net.conv2.stride[0]
        dev = (G[0] if isinstance(G, tuple) else G).device
        f = dict(dtype=torch.float32, device=dev)
        L = lib()
        st = _stream()
        P, Po = n * h * w, n * ho * wo
        s1, s2, s3 = sites.get(id(net.bn1)), sites.get(id(net.bn2)), sites.get(id(net.bn3))

        def bn_bwd(g, x, bn, site, relu, mask=None, gm=None, rowadd=None, c=None, hw=None, dx=None):
            dx = torch.empty_like(g) if dx is None else dx
            st = _stream()
            dg, db = sink(bn.weight), sink(bn.bias)
            inj = site.inj if site else (None, None, None, None)
            check(L.vitta_bn_bwd_cm_f32(_p(g), None, _p(x), _p(mask), _p(rowadd), (1.0 / hw) if rowadd is not None else 0.0,
                                        _bn_ptrs(bn), float(bn.eps), _p(inj[0]), _p(inj[1]), _p(inj[2]), _p(inj[3]),
                                        int(relu) | (_lib.BN_BWD_INJ_RAW if (site and site.raw) else 0),
                                        _p(dx), _p(gm), _p(dg), _p(db), c, nb, t, hw, st), "vitta_bn_bwd_cm_f32")
            return dx

        # bn3 (+ identity add + ReLU) backward -- unless the launch that produced G did it (BN3_FOLD)
        if isinstance(G, tuple):
            dx3, g_id = G
            G = g_id  # (device, shape)
        else:
            g_id = torch.empty_like(G)
            dx3 = bn_bwd(G, sv["x3"], net.bn3, s3, True, mask=sv["out"], gm=g_id, c=4 * p, hw=ho * wo)
        # identity / downsample path of a stage's first bottleneck, beside the main chain: bn_d backward, its data gradient
        if net.downsample is not None:
            dconv, dbn = net.downsample[0], net.downsample[1]
            sd = sites.get(id(dbn))
            ds = dconv.stride[0]
            dxd = torch.empty_like(G)
            gd = torch.empty(cin, Po if ds == 2 else P, **f)
            wdb = self.packed(dconv, "b", True)

            def identity_path():
                bn_bwd(g_id, sv["xd"], dbn, sd, False, c=4 * p, hw=ho * wo, dx=dxd)
                if dconv.weight.requires_grad:
                    wgrad(self.geo("f", n, h, w, 1, ds, 0), sv["xin"], dxd, sink(dconv.weight), cin, 4 * p)
                CV.launch(self.geo("b", n, h, w, 1, ds, 0)[0], dxd, wdb, gd, 4 * p, cin)
            identity_path()
        # conv3 data gradient, epilogue = bn2 (+ReLU) backward
        dx2 = torch.empty(p, Po, **f)
        i2 = s2.inj if s2 else None
        CV.launch(self.geo("b", n, ho, wo)[0], dx3, self.packed(net.conv3, "b", True), dx2, 4 * p, p,
                  flags=CV.CONV_BWD_BN | CV.CONV_BWD_RELU | (CV.CONV_INJ_RAW if (s2 and s2.raw) else 0), bwd_bn=_bn_t(net.bn2),
                  eps=net.bn2.eps, bwd_x=sv["x2"], inj=i2,
                  dgamma=sink(net.bn2.weight), dbeta=sink(net.bn2.bias))
        if net.conv3.weight.requires_grad:
            wgrad(self.geo("f", n, ho, wo), sv["a2"], dx3, sink(net.conv3.weight), p, 4 * p)
        del dx3
        if net.conv2.weight.requires_grad:
            wgrad(self.geo("f", n, h, w, 3, s, 1), sv["a1"], dx2, sink(net.conv2.weight), p, p)
        # conv2 data gradient -> d a1
        ga1 = torch.empty(p, P, **f)
        wb2 = self.packed(net.conv2, "b", True)
        gm = self.geo("bm", n, h, w, 3, s, 1) if s == 2 else None
        if gm is not None and self.merged_ok(gm, wb2, p):
            CV.launch(gm, dx2, wb2, ga1, p, p)  # stride 2: the four parity classes of the input pixel in one launch
        else:
            for g in self.geo("b", n, h, w, 3, s, 1):
                CV.launch(g, dx2, wb2, ga1, p, p)
        del dx2
        # TAM backward
        bn1p = _bn_ptrs(net.bn1)
        ga = torch.empty(p, P, **f)
        ggate = torch.empty(nb * p * t * 4, **f)
        gkern = torch.empty(nb * p, 3, **f)
        check(L.vitta_tam_agg_bwd_cm_f32(_p(sv["x1"]), bn1p, float(net.bn1.eps), _p(sv["gate"]), _p(sv["kern"]), _p(ga1), p, nb,
                                         t, h * w, _p(ga), _p(ggate), _p(gkern), st), "vitta_tam_agg_bwd_cm_f32")
        del ga1
        bg, bl = tam.G[1], tam.L[1]
        from .ops import _ptr4
        gbuf = torch.empty(nb * p * t + nb * (p // 4) * t, **f)  # d pooled | scratch
        from .ops import tam_branch_fused_supported
        bn_sinks = _ptr4(sink(bg.weight), sink(bg.bias), sink(bl.weight), sink(bl.bias))
        dw0, dw3 = sink(tam.L[0].weight), sink(tam.L[3].weight)
        tq = []  # launches for the helper stream besides the convolutions' weight gradients
        if WGRAD_SIDE and (dw0 is not None or dw3 is not None):
            # the L branch's weight gradients (C x C/4 x 4 accumulations per clip) leave the main chain: the fused backward runs
            # without them (17-21 us instead of 37) and vitta_tam_branch_wgrad_f32 follows on the weight-gradient helper stream
            w_sinks = _ptr4(sink(tam.G[0].weight), sink(tam.G[3].weight), None, None)
            pooled_sv, gate_sv, hpre_sv, ptc_sv = sv["pooled"], sv["gate"], sv["hpre"], sv["ptc"]
            tq.append(lambda: check(L.vitta_tam_branch_wgrad_f32(
                _p(pooled_sv), ptc_sv, _p(gate_sv), _p(ggate), C.c_void_p(hpre_sv.data_ptr() + 4 * nb * (p // 4) * t),
                C.c_void_p(gbuf.data_ptr() + 4 * nb * p * t), nb, p, t, _p(dw0), _p(dw3), _stream()), "vitta_tam_branch_wgrad_f32"))
        else:
            w_sinks = _ptr4(sink(tam.G[0].weight), sink(tam.G[3].weight), dw0, dw3)
        if tam_branch_fused_supported(nb, p, t):
            check(L.vitta_tam_branch_bwd_fused_f32(_p(sv["pooled"]), _p(tam.G[0].weight), _ptr4(bg.weight, bg.bias, bg.running_mean, bg.running_var),
                                                   float(bg.eps), _p(tam.G[3].weight), _p(tam.L[0].weight),
                                                   _ptr4(bl.weight, bl.bias, bl.running_mean, bl.running_var), float(bl.eps),
                                                   _p(tam.L[3].weight), nb, p, t, _p(sv["kern"]), _p(sv["gate"]), _p(sv["hpre"]), _p(gkern),
                                                   _p(ggate), _p(gbuf), bn_sinks, w_sinks, _p(_sync(dev)), sv["ptc"], st),
                  "vitta_tam_branch_bwd_fused_f32")
        else:
            check(L.vitta_tam_branch_bwd_f32(_p(sv["pooled"]), _p(tam.G[0].weight), _ptr4(bg.weight, bg.bias, bg.running_mean, bg.running_var),
                                             float(bg.eps), _p(tam.G[3].weight), _p(tam.L[0].weight),
                                             _ptr4(bl.weight, bl.bias, bl.running_mean, bl.running_var), float(bl.eps),
                                             _p(tam.L[3].weight), nb, p, t, _p(sv["kern"]), _p(sv["gate"]), _p(sv["hpre"]), _p(gkern),
                                             _p(ggate), _p(gbuf), bn_sinks, w_sinks, sv["ptc"], st), "vitta_tam_branch_bwd_f32")
        # bn1 (+ReLU) backward with the pooling gradient added per (n, c, t) row
        dx1 = bn_bwd(ga, sv["x1"], net.bn1, s1, True, rowadd=gbuf, c=p, hw=h * w)
        del ga
        if net.conv1.weight.requires_grad:
            wgrad(self.geo("f", n, h, w), sv["xin"], dx1, sink(net.conv1.weight), cin, p)
        # join the identity / downsample path
        gin = torch.empty(cin, P, **f)
        if net.downsample is not None:
            res, rflag = gd, (CV.CONV_RES_HALF if ds == 2 else CV.CONV_RES)
        else:
            res, rflag = g_id, CV.CONV_RES
        fold = None
        if BN3_FOLD and prev is not None:
            pnet, psv = prev[0].net, prev[1]
            ps3 = sites.get(id(pnet.bn3))
            g_idp = torch.empty(cin, P, **f)  # the masked gradient: the previous block's identity path reads it
            CV.launch(self.geo("b", n, h, w)[0], dx1, self.packed(net.conv1, "b", True), gin, p, cin,
                      flags=rflag | CV.CONV_BWD_BN | CV.CONV_BWD_RELU | (CV.CONV_INJ_RAW if (ps3 and ps3.raw) else 0), res=res,
                      bwd_bn=_bn_t(pnet.bn3), eps=pnet.bn3.eps, bwd_x=psv["x3"], bwd_mask=psv["out"], inj=ps3.inj if ps3 else None,
                      dgamma=sink(pnet.bn3.weight), dbeta=sink(pnet.bn3.bias), y_raw=g_idp)
            fold = (gin, g_idp)  # gin now holds dx3 of the previous block
        else:
            CV.launch(self.geo("b", n, h, w)[0], dx1, self.packed(net.conv1, "b", True), gin, p, cin, flags=rflag, res=res)
        if wq or tq:
            helper = _Fork(dev, role=1)  # waits for everything this block has queued
            with helper:  # four launches + ONE reduction of their partial tiles
                for fn in tq:
                    fn()
                CV.wgrad_reduce([CV.wgrad(*wa, defer=i, **wkw) for i, (wa, wkw) in enumerate(wq)])
            done = torch.cuda.Event()
            done.record(helper.side)
            self._wgrad_pending.append((helper, (wq, tq, ggate, gbuf), done))
            # the operands of a block's weight gradients (its saved activations and gradient tensors) stay alive until the
            # main stream has waited for THAT block's helper work; two blocks back it has long finished, so the wait is free
            # and the tensors of at most three blocks are held instead of all sixteen
            while len(self._wgrad_pending) > 2:
                h0, _, ev0 = self._wgrad_pending.pop(0)
                h0.main.wait_event(ev0)
        return fold if fold is not None else gin

    def join_wgrads(self):
        """The current stream waits for the weight gradients issued on the helper stream; their operands may be freed."""
        if self._wgrad_pending:
            self._wgrad_pending[-1][0].join()  # one helper stream, in order: its last launch covers the earlier ones
            self._wgrad_pending = []

    def backward(self, ctxd, gfeat, sink):
        c, n, h, w = ctxd["last"]
        G = torch.empty(c, n * h * w, dtype=torch.float32, device=gfeat.device)
        check(lib().vitta_avgpool_cm_bwd_f32(_p(gfeat.contiguous()), c, n, h * w, _p(G), _stream()), "vitta_avgpool_cm_bwd_f32")
        blocks = self.blocks()
        try:
            for i in range(len(blocks) - 1, -1, -1):
                sv = ctxd["tape"][i]
                G = self.block_backward(blocks[i], sv, G, ctxd["sites"], sink, prev=(blocks[i - 1], ctxd["tape"][i - 1]) if i > 0 else None)
                sv.clear()
                if self.after_block is not None:
                    self.join_wgrads()  # a gradient bucket may leave now
                    self.after_block(i)
            self.join_wgrads()
        finally:
            self._wgrad_pending = []  # (a backward that raised leaves nothing behind for the next step)
        h0, w0 = ctxd["pooled_hw"]
        if ctxd["stem"] is None:  # the stem ran outside (trainable 7x7 convolution): hand its output gradient back
            return CV.from_cm(G, n, h0, w0)
        # stem: bn1 affine gradients through the fused BN + ReLU + max-pool pass (the 7x7 convolution is frozen)
        x0 = ctxd.get("x")
        self.stem_backward(ctxd["stem"], G, (n, h0, w0), sink, x=x0)
        return None


class TrunkFunction(torch.autograd.Function):
    """features = trunk(x) as one autograd node.  Inputs after `runner`: the trainable parameters (so that autograd
    schedules this node); their gradients are written by the kernels straight into `.grad` storage where it exists."""

    @staticmethod
    def forward(ctx, x, runner, pooled, *params):
        ctx.set_materialize_grads(False)
        feat, tape = runner.forward(x, True, pooled_in=pooled)
        ctx.runner, ctx.tape, ctx.params = runner, tape, params
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        from . import ops
        runner, params = ctx.runner, ctx.params
        if gfeat is None:
            return (None, None, None) + tuple(None for _ in params)
        bufs = {}

        def sink(param):
            if not param.requires_grad:
                return None
            hit = bufs.get(id(param))
            if hit is None:
                buf, ret = ops._grad_sink(param, True, zero=True)
                hit = bufs[id(param)] = (buf, ret)
            return hit[0]

        gpooled = runner.backward(ctx.tape, gfeat, sink)
        ctx.tape = None
        grads = []
        for p in params:
            hit = bufs.get(id(p))
            grads.append(hit[1] if hit is not None else None)
        return (None, None, gpooled) + tuple(grads)


_DECLINED = set()


def _declined(x, why):
    """The module-by-module path (vendor-library convolutions + the fused BN passes) takes over: say so ONCE per reason, loudly --
    a production run must not lose the hand-written kernels silently.  VITTA_REQUIRE_TRUNK=1 raises instead.
    CPU tensors (BASELINE config 0, the host-logic tests) are not a decline of the HIP path: silent."""
    if not x.is_cuda:
        return
    if os.environ.get("VITTA_REQUIRE_TRUNK", "0") == "1":
        from ._lib import VittaHipError
        raise VittaHipError("the hand-written TANet trunk declined this pass: " + why)
    if why not in _DECLINED:
        _DECLINED.add(why)
        msg = "[vitta_amd] WARNING: the hand-written TANet trunk (trunk.py / conv_b3.hip) DECLINED this pass -> module-by-module " \
              "path with vendor-library convolutions: " + why
        logging.getLogger("vitta_amd").warning(msg)
        print(msg, file=sys.stderr, flush=True)


def runner_of(resnet):
    runner = getattr(resnet, "_vitta_trunk", None)
    if runner is None:
        runner = TrunkRunner(resnet)
        object.__setattr__(resnet, "_vitta_trunk", runner)
    return runner


def run(resnet, x):
    """features [N, 2048] of the trunk on the hand-written path, or None if the configuration needs the module path."""
    runner = runner_of(resnet)
    why = runner.why_not(x)
    if why is not None:
        _declined(x, why)
        return None
    params = [p for m in runner.bn2d_modules() for p in (m.weight, m.bias)]
    for b in runner.blocks():
        params += [b.tam.G[1].weight, b.tam.G[1].bias, b.tam.L[1].weight, b.tam.L[1].bias,
                   b.tam.G[0].weight, b.tam.G[3].weight, b.tam.L[0].weight, b.tam.L[3].weight,
                   b.net.conv1.weight, b.net.conv2.weight, b.net.conv3.weight]
        if b.net.downsample is not None:
            params.append(b.net.downsample[0].weight)
    # under autograd the node form runs whenever something trains OR statistics hooks are bound to the engine (their sites are
    # opened and deposited by the keep = True forward even if only parameters outside the trunk are adapted)
    hooked = torch.is_grad_enabled() and any(_engine_hook(m)[1] is not None for m in runner.bn2d_modules())
    if torch.is_grad_enabled() and (hooked or any(p.requires_grad for p in params + [resnet.conv1.weight])):
        pooled = None
        y_w = (x.shape[3] - 1) // 2 + 1
        if resnet.conv1.weight.requires_grad and (y_w % 4 or y_w > 256):  # stem gradient kernels: tiled path only
            from .fused_bn import bn_act
            pooled = resnet.maxpool(bn_act(resnet.bn1, resnet.conv1(x), relu=True, act=resnet.relu))
            stem_params = {id(resnet.bn1.weight), id(resnet.bn1.bias)}
            params = [p for p in params if id(p) not in stem_params]
        if pooled is None and resnet.conv1.weight.requires_grad:
            params = params + [resnet.conv1.weight]
        return TrunkFunction.apply(x, runner, pooled, *[p for p in params if p.requires_grad])
    with torch.no_grad():
        feat, _ = runner.forward(x, False)
    return feat
