"""Torch-facing wrappers of the C ABI (device memory + streams only; all arithmetic of the
hooked-statistics path runs in libvitta_hip.so).

Every function here requires CUDA(HIP) tensors and raises if the library is missing -- there
is no eager fallback for the HIP path.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import LAYOUT_NCHW, LAYOUT_NHWC, REG_TYPES, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_cuda_feature(t, name):
    """Hooked features may be fp32 or bfloat16 (widened to fp32 in registers by the moments kernels)."""
    if not t.is_cuda:
        raise _lib.VittaHipError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.VittaHipError(f"{name} must be float32 or bfloat16 (got {t.dtype})")


def _require_cuda_f32(t, name):
    if not t.is_cuda:
        raise _lib.VittaHipError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise _lib.VittaHipError(f"{name} must be float32 (got {t.dtype})")


def feature_layout(feature, kind):
    """(outer, C, inner, layout, flat_feature) of a hooked feature.

    kind 'bn2d': [N*T, C, H, W]  (norm_stats_utils.py:188-193)   -> NCHW, outer = N*T
    kind 'bn3d': [N, C, T, H, W] (norm_stats_utils.py:195-199)   -> NCHW with inner = T*H*W
    kind 'ln'  : [N, T, H, W, C] (norm_stats_utils.py:222-230)   -> NHWC, rows = N*T*H*W
    """
    if kind == "bn2d":
        nt, c, h, w = feature.shape
        return nt, c, h * w, LAYOUT_NCHW
    if kind == "bn3d":
        n, c, t, h, w = feature.shape
        return n, c, t * h * w, LAYOUT_NCHW
    if kind == "ln":
        c = feature.shape[-1]
        return feature.numel() // c, c, 1, LAYOUT_NHWC
    if kind == "rows":  # (R, C): BatchNorm1d input of TAM.G, statistics over the rows (BNS_utils.py:43-45)
        r, c = feature.shape
        return r, c, 1, LAYOUT_NHWC
    if kind == "nct":  # (N, C, T): BatchNorm1d input of TAM.L, statistics over (N, T) (BNS_utils.py:46-48)
        n, c, t = feature.shape
        return n, c, t, LAYOUT_NCHW
    raise ValueError(f"unknown feature kind {kind}")


# ------------------------------------------------------------------------------------------------
# single-layer moments (drop-in for one hook invocation)
# ------------------------------------------------------------------------------------------------
def moments(feature, kind):
    """Per-channel (mean, biased var) over (N, T, H, W) of one hooked feature (fp32 or bfloat16, fp32 accumulation and
    results); no autograd."""
    _require_cuda_feature(feature, "feature")
    x = feature if feature.is_contiguous() else feature.contiguous()
    outer, c, inner, layout = feature_layout(x, kind)
    L = lib()
    ws_bytes = L.vitta_moments_workspace_bytes(outer, c, inner, layout)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty(c, dtype=torch.float32, device=x.device)
    sfx = "bf16" if x.dtype == torch.bfloat16 else "f32"
    if layout == LAYOUT_NCHW:
        check(getattr(L, f"vitta_moments_nchw_{sfx}")(_p(x), outer, c, inner, _p(mean), _p(var), _p(ws), ws_bytes, _stream()),
              f"vitta_moments_nchw_{sfx}")
    else:
        check(getattr(L, f"vitta_moments_nhwc_{sfx}")(_p(x), outer, c, _p(mean), _p(var), _p(ws), ws_bytes, _stream()),
              f"vitta_moments_nhwc_{sfx}")
    return mean, var


def stat_align_bwd(x, gout, kind, mu, coef_a, coef_b, gscale=None, out=None):
    """gin = gout + gscale * (a_c + b_c (x - mu_c)); `out` may alias gout."""
    _require_cuda_f32(x, "x")
    outer, c, inner, layout = feature_layout(x, kind)
    if out is None:
        out = torch.empty_like(x)
    check(lib().vitta_stat_align_bwd_f32(_p(x), _p(gout), _p(out), outer, c, inner, layout, _p(mu), _p(coef_a),
                                         _p(coef_b), _p(gscale), _stream()), "vitta_stat_align_bwd_f32")
    return out


class FeatureMoments(torch.autograd.Function):
    """(mean, var) = moments(feature) with the analytic backward
    dx = gmean/n + gvar * 2 (x - mean)/n   (autograd of norm_stats_utils.py:242-243)."""

    @staticmethod
    def forward(ctx, feature, kind):
        x = feature if feature.is_contiguous() else feature.contiguous()
        mean, var = moments(x, kind)
        ctx.kind = kind
        ctx.save_for_backward(x, mean)
        return mean, var

    @staticmethod
    def backward(ctx, gmean, gvar):
        x, mean = ctx.saved_tensors
        _, c, _, _ = feature_layout(x, ctx.kind)
        n = x.numel() // c
        a = (gmean / n).contiguous()
        b = (gvar * (2.0 / n)).contiguous()
        return stat_align_bwd(x, None, ctx.kind, mean, a, b), None


# ------------------------------------------------------------------------------------------------
# prediction consistency
# ------------------------------------------------------------------------------------------------
class PredConsis(torch.autograd.Function):
    """compute_pred_consis (utils/pred_consistency_utils.py:15-31): loss and dloss/dlogits in one launch."""

    @staticmethod
    def forward(ctx, preds):
        _require_cuda_f32(preds, "preds")
        z = preds.contiguous()
        b, v, k = z.shape
        loss = torch.empty(1 + b, dtype=torch.float32, device=z.device)
        grad = torch.empty_like(z)
        check(lib().vitta_pred_consis_f32(_p(z), b, v, k, _p(loss), _p(grad), _stream()), "vitta_pred_consis_f32")
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g


def pred_consis(preds):
    return PredConsis.apply(preds)


# ------------------------------------------------------------------------------------------------
# batched plan: all hooked layers of one forward in single launches
# ------------------------------------------------------------------------------------------------
class StatPlan:
    """Owns a vitta_plan plus the packed per-channel device buffers of the batched path."""

    def __init__(self, shapes, device, target_blocks=0, nt_loads=None, nsplit=None):
        # shapes: list of (outer, C, inner, layout); nsplit: optional per-layer frame split (0 = library's choice)
        self.device = torch.device(device)
        self.shapes = [tuple(int(v) for v in s) for s in shapes]
        n = len(self.shapes)
        if n == 0 or n > _lib.MAX_LAYERS:
            raise ValueError(f"number of hooked layers must be in 1..{_lib.MAX_LAYERS}, got {n}")
        arr = (_lib.LayerShape * n)(*[_lib.LayerShape(*s) for s in self.shapes])
        handle = C.c_void_p()
        ns = (C.c_int32 * n)(*[int(v) for v in nsplit]) if nsplit is not None else None
        with torch.cuda.device(self.device):
            check(lib().vitta_plan_create_split(arr, n, target_blocks, ns, C.byref(handle)), "vitta_plan_create_split")
        self._h = handle
        L = lib()
        if nt_loads is not None:  # default: the library's (non-temporal loads: every feature is read once)
            check(L.vitta_plan_set_option(self._h, 1, int(bool(nt_loads))), "vitta_plan_set_option")
        # device tables of the plan: a torch-owned buffer (the library never allocates device memory)
        self.tables = torch.empty(int(L.vitta_plan_table_bytes(self._h)), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(L.vitta_plan_upload(self._h, _p(self.tables), self.tables.numel(), _stream()), "vitta_plan_upload")
        self.n_layers = n
        self.total_channels = int(L.vitta_plan_total_channels(self._h))
        self.offsets = [int(L.vitta_plan_channel_offset(self._h, i)) for i in range(n)]
        self.ws_bytes = int(L.vitta_plan_workspace_bytes(self._h))
        self.num_blocks = int(L.vitta_plan_num_blocks(self._h))
        f = dict(dtype=torch.float32, device=self.device)
        tc = self.total_channels
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        # additive statistics, packed [cnt(L) | s1(TC) | s2(TC)] so ONE all-reduce covers them
        self.stats = torch.zeros(n + 2 * tc, **f)
        self.cnt = self.stats[:n]
        self.s1 = self.stats[n:n + tc]
        self.s2 = self.stats[n + tc:]
        self.mu = torch.zeros(tc, **f)
        self.coef_a = torch.zeros(tc, **f)
        self.coef_b = torch.zeros(tc, **f)
        self.layer_loss = torch.zeros(n, **f)
        self.total_loss = torch.zeros(1, **f)
        self._ptr_arr = (C.c_void_p * n)()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().vitta_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def channel_slice(self, layer):
        o = self.offsets[layer]
        return slice(o, o + self.shapes[layer][1])

    def moments(self, feats, shift=None, events=None):
        """feats: list of contiguous CUDA tensors (one per layer), all fp32 or all bfloat16 -> fills cnt/s1/s2.
        `events`: a KernelEventPair attached to the dispatch of the streaming (partials) kernel -- bench.py's
        live kernel timing."""
        self.cnt_src = None  # (this launch writes the counts)
        if len(feats) != self.n_layers:
            raise ValueError("one feature per planned layer expected")
        bf16 = len(feats) > 0 and feats[0].dtype == torch.bfloat16
        for i, t in enumerate(feats):
            _require_cuda_feature(t, f"feature {i}")
            if (t.dtype == torch.bfloat16) != bf16:
                raise _lib.VittaHipError("one launch reads features of one element type (all fp32 or all bfloat16)")
            if not t.is_contiguous():
                raise _lib.VittaHipError(f"feature {i} must be contiguous")
            outer, c, inner, _ = self.shapes[i]
            if t.numel() != outer * c * inner:
                raise _lib.VittaHipError(f"feature {i} has {t.numel()} elements, plan expects {outer * c * inner}")
            self._ptr_arr[i] = t.data_ptr()
        if bf16:
            if events is not None:
                raise _lib.VittaHipError("dispatch-timed launches exist for fp32 features only")
            check(lib().vitta_moments_batched_bf16(self._h, self._ptr_arr, _p(shift), _p(self.cnt), _p(self.s1),
                                                   _p(self.s2), _p(self.ws), self.ws_bytes, _stream()),
                  "vitta_moments_batched_bf16")
        elif events is None:
            check(lib().vitta_moments_batched_f32(self._h, self._ptr_arr, _p(shift), _p(self.cnt), _p(self.s1),
                                                  _p(self.s2), _p(self.ws), self.ws_bytes, _stream()),
                  "vitta_moments_batched_f32")
        else:
            check(lib().vitta_moments_partials_timed_f32(self._h, self._ptr_arr, _p(self.ws), self.ws_bytes, _stream(),
                                                         events.start, events.stop), "vitta_moments_partials_timed_f32")
            check(lib().vitta_moments_finalize_f32(self._h, _p(shift), _p(self.cnt), _p(self.s1), _p(self.s2),
                                                   _p(self.ws), self.ws_bytes, _stream()),
                  "vitta_moments_finalize_f32")
        return self.cnt, self.s1, self.s2

    def partials(self, feats):
        """First stage only (the streaming kernel), for micro-benchmarks."""
        for i, t in enumerate(feats):
            self._ptr_arr[i] = t.data_ptr()
        name = "vitta_moments_partials_bf16" if feats[0].dtype == torch.bfloat16 else "vitta_moments_partials_f32"
        check(getattr(lib(), name)(self._h, self._ptr_arr, _p(self.ws), self.ws_bytes, _stream()), name)

    def layer_geometry(self, layer):
        """(nsplit, nchunks, slots, ws_off) of a layer: where a fused BN pass deposits its partial triples."""
        out = (C.c_int64 * 5)()
        check(lib().vitta_plan_layer_geometry(self._h, layer, out), "vitta_plan_layer_geometry")
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def triples_ptr(self, layer):
        return self.ws.data_ptr() + 12 * self.layer_geometry(layer)[3]

    def finalize(self, shift=None):
        """Second stage only: the partial triples were written by fused BN passes (vitta_bn_act_fwd_f32)."""
        self.cnt_src = None
        check(lib().vitta_moments_finalize_f32(self._h, _p(shift), _p(self.cnt), _p(self.s1), _p(self.s2), _p(self.ws),
                                               self.ws_bytes, _stream()), "vitta_moments_finalize_f32")
        return self.cnt, self.s1, self.s2

    def mean_var(self, shift=None):
        mean = torch.empty(self.total_channels, dtype=torch.float32, device=self.device)
        var = torch.empty_like(mean)
        check(lib().vitta_moments_to_meanvar_f32(self._h, _p(shift), _p(self.cnt), _p(self.s1), _p(self.s2),
                                                 _p(mean), _p(var), _stream()), "vitta_moments_to_meanvar_f32")
        return mean, var

    zeroes_in_align = True  # align(..., zero=t) resets the one-element tensor t inside its launch
    cnt_src = None          # per-layer counts for align() when they are constants of the plan (the direct path of a single rank)

    def rebind_stats(self, t):
        """Move [cnt | s1 | s2] into the caller's storage `t` (>= n_layers + 2 total_channels floats, e.g. a slice that the
        step's one fill zeroes: tta.FlatArena.reserve_zeroed)."""
        n, tc = self.n_layers, self.total_channels
        if t.dtype != torch.float32 or t.numel() < n + 2 * tc or not t.is_contiguous():
            raise VittaHipError("rebind_stats: a contiguous float32 tensor of n_layers + 2 * total_channels elements is needed")
        self.stats = t[:n + 2 * tc]
        self.cnt, self.s1, self.s2 = self.stats[:n], self.stats[n:n + tc], self.stats[n + tc:]

    def align(self, shift, ema_mean, ema_var, src_mean, src_var, momentum, reg_type, zero=None):
        # (a tensor of its own per launch: norm_stats._LossReg hands it on without a copy)
        self.total_loss = torch.empty(1, dtype=torch.float32, device=self.device)
        self.total_loss._vitta_fresh = True
        check(lib().vitta_stat_align_fwd_f32(self._h, _p(shift), _p(self.cnt if self.cnt_src is None else self.cnt_src), _p(self.s1), _p(self.s2),
                                             _p(ema_mean), _p(ema_var), _p(src_mean), _p(src_var),
                                             float(momentum), REG_TYPES[reg_type], _p(self.layer_loss),
                                             _p(self.total_loss), _p(self.mu), _p(self.coef_a), _p(self.coef_b),
                                             _p(zero), _stream()), "vitta_stat_align_fwd_f32")
        return self.total_loss, self.layer_loss


class KernelEventPair:
    """Two hipEvents the library attaches to ONE kernel dispatch (vitta_moments_partials_timed_f32): elapsed_ms()
    is that kernel's own duration, the figure rocprofv3 --kernel-trace reports for the launch."""

    def __init__(self):
        a, b = C.c_void_p(), C.c_void_p()
        check(lib().vitta_event_create(C.byref(a)), "vitta_event_create")
        check(lib().vitta_event_create(C.byref(b)), "vitta_event_create")
        self.start, self.stop = a, b

    def elapsed_ms(self):
        ms = C.c_float()
        check(lib().vitta_event_elapsed_ms(self.start, self.stop, C.byref(ms)), "vitta_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            lib().vitta_event_destroy(self.start)
            lib().vitta_event_destroy(self.stop)
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# parameter-gradient sinks
# ------------------------------------------------------------------------------------------------
DIRECT_PARAM_GRAD = True


def _grad_sink(param, needed, zero=True):
    """Where a backward kernel puts a parameter gradient: (buffer, value returned to autograd).

    A leaf whose `.grad` already exists (the views of tta.FlatArena, zeroed once per step) is accumulated into IN
    PLACE and autograd gets None -- exactly what its AccumulateGrad node would do with `grad += new`, minus one
    launch per tensor (170 five-microsecond adds per TANet step in the r1h profile).  Only under a plain
    `.backward()`: `torch.autograd.grad` callers must switch DIRECT_PARAM_GRAD off."""
    if not needed:
        return None, None
    g = param.grad
    if g is None:
        g = getattr(param, "_vitta_arena_view", None)  # tta.FlatArena detached `.grad` for this backward
    if (DIRECT_PARAM_GRAD and param.is_leaf and g is not None and g.dtype == torch.float32 and g.is_contiguous()
            and g.device == param.device and g.shape == param.shape):
        param._vitta_direct_grad = True  # tta.FlatArena keeps the live view attached for this parameter
        return g, None
    buf = torch.zeros_like(param, memory_format=torch.contiguous_format) if zero \
        else torch.empty_like(param, memory_format=torch.contiguous_format)
    return buf, buf


# ------------------------------------------------------------------------------------------------
# TAM
# ------------------------------------------------------------------------------------------------
class TamPool(torch.autograd.Function):
    """pooled[n,c,t] = mean_hw x[n*T+t, c, :]  (temporal_module.py:47-52 without the permute copy)."""

    @staticmethod
    def forward(ctx, x, n_segment):
        _require_cuda_f32(x, "x")
        x = x.contiguous()
        nt, c, h, w = x.shape
        n = nt // n_segment
        pool = torch.empty(n, c, n_segment, dtype=torch.float32, device=x.device)
        check(lib().vitta_tam_pool_f32(_p(x), n, n_segment, c, h * w, _p(pool), _stream()), "vitta_tam_pool_f32")
        ctx.dims = (n, n_segment, c, h, w)
        return pool

    @staticmethod
    def backward(ctx, gpool):
        n, t, c, h, w = ctx.dims
        gx = torch.zeros(n * t, c, h, w, dtype=torch.float32, device=gpool.device)
        check(lib().vitta_tam_pool_bwd_f32(_p(gpool.contiguous()), n, t, c, h * w, _p(gx), _stream()),
              "vitta_tam_pool_bwd_f32")
        return gx, None


class TamAggregate(torch.autograd.Function):
    """out[n,t,c] = sum_j K[n,c,j] gate[n,c,t+j-1] x[n,t+j-1,c]  (temporal_module.py:56-63, fused)."""

    @staticmethod
    def forward(ctx, x, gate, kern, n_segment):
        _require_cuda_f32(x, "x")
        x = x.contiguous()
        gate = gate.contiguous()
        kern = kern.contiguous()
        nt, c, h, w = x.shape
        n = nt // n_segment
        out = torch.empty_like(x)
        check(lib().vitta_tam_agg_fwd_f32(_p(x), _p(gate), _p(kern), n, n_segment, c, h * w, _p(out), _stream()),
              "vitta_tam_agg_fwd_f32")
        ctx.save_for_backward(x, gate, kern)
        ctx.dims = (n, n_segment, c, h * w)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, gate, kern = ctx.saved_tensors
        n, t, c, hw = ctx.dims
        gout = gout.contiguous()
        gx = torch.empty_like(x)
        ggate_buf = torch.empty(n * c * t * 4, dtype=torch.float32, device=x.device)
        gkern = torch.empty_like(kern)
        check(lib().vitta_tam_agg_bwd_f32(_p(x), _p(gate), _p(kern), _p(gout), n, t, c, hw, _p(gx), _p(ggate_buf),
                                          _p(gkern), _stream()), "vitta_tam_agg_bwd_f32")
        ggate = ggate_buf[: n * c * t].view_as(gate)
        return gx, ggate, gkern, None


# ------------------------------------------------------------------------------------------------
# fused window attention (Video Swin)
# ------------------------------------------------------------------------------------------------
def wmsa_supported(n_tokens, head_dim):
    return bool(lib().vitta_wmsa_supported(int(n_tokens), int(head_dim)))


def wmsa_rel_supported(n_tokens, head_dim, table_rows):
    """relative-position-table form: windows up to 800 tokens; tables up to 4096 rows (8192 beyond 400 tokens)"""
    ok = bool(lib().vitta_wmsa_rel_supported(int(n_tokens), int(head_dim)))
    return ok and table_rows <= (8192 if n_tokens > 400 else 4096)


class WindowAttention(torch.autograd.Function):
    """softmax(scale q k^T + bias (+ mask)) v per (window, head) in one launch; the N x N matrix never
    reaches HBM (swin_transformer.py:144-168).  qkv (B_, N, 3C) -> (B_, N, C)."""

    @staticmethod
    def forward(ctx, qkv, bias, mask, scale, num_heads):
        _require_cuda_f32(qkv, "qkv")
        qkv, bias = qkv.contiguous(), bias.contiguous()
        mask = mask.contiguous() if mask is not None else None
        b_, n, c3 = qkv.shape
        c = c3 // 3
        hd = c // num_heads
        out = torch.empty(b_, n, c, dtype=torch.float32, device=qkv.device)
        lse = torch.empty(b_, num_heads, n, dtype=torch.float32, device=qkv.device)
        nw = mask.shape[0] if mask is not None else 1
        check(lib().vitta_wmsa_fwd_f32(_p(qkv), _p(bias), _p(mask), nw, b_, n, num_heads, hd, float(scale), _p(out),
                                       _p(lse), _stream()), "vitta_wmsa_fwd_f32")
        ctx.save_for_backward(qkv, bias, mask, out, lse)
        ctx.meta = (float(scale), num_heads, hd, nw)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bias, mask, out, lse = ctx.saved_tensors
        scale, nh, hd, nw = ctx.meta
        b_, n, _ = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        dbias = torch.zeros_like(bias) if ctx.needs_input_grad[1] else None
        check(lib().vitta_wmsa_bwd_f32(_p(qkv), _p(bias), _p(mask), nw, b_, n, nh, hd, scale, _p(out), _p(dout), _p(lse),
                                       _p(delta), _p(dqkv), _p(dbias), _stream()), "vitta_wmsa_bwd_f32")
        return dqkv, dbias, None, None, None


# bfloat16-operand window attention (vitta_wmsa_rel_{fwd,bwd}_bf16; BASELINE config 5's recipe): opt-in, and only where the
# relative-position table is frozen (its gradient lives on the fp32 kernels)
WMSA_BF16 = False


_dtable_ws = {}  # (device index, stream) -> scratch of the bf16 attention's table-gradient columns


class WindowAttentionRel(torch.autograd.Function):
    """WindowAttention with the relative-position bias looked up from the [T, nH] table and the shift
    mask derived from region ids inside the kernel (nothing of size N x N in memory).

    `rowmap` None: qkv (B_, N, 3C) in the partitioned layout -> (B_, N, C).
    `rowmap` int32 (nW, N): qkv (B, L, 3C) in the NATURAL token order, L = nW * N; window w of sample s gathers its
    tokens at rows s*L + rowmap[w] and scatters its output to the same rows -> (B, L, C): the cyclic shift and the
    window partition / reverse never touch memory."""

    @staticmethod
    def forward(ctx, qkv, table, code, code_off, region, scale, num_heads, rowmap=None):
        """A bfloat16 qkv (the bf16 data flow: ops.wmsa_io16_ok said the bf16-operand kernels take this shape) is read as it is
        and the context comes back as bfloat16 -- the dense products on either side hand over 2-byte activations."""
        io16 = qkv.dtype == torch.bfloat16
        if not io16:
            _require_cuda_f32(qkv, "qkv")
        qkv, table = qkv.contiguous(), table.contiguous()
        c = qkv.shape[-1] // 3
        hd = c // num_heads
        if rowmap is None:
            b_, n, _ = qkv.shape
            nwm, tokens = 1, 0
            out = torch.empty(b_, n, c, dtype=qkv.dtype, device=qkv.device)
        else:
            bsz, tokens, _ = qkv.shape
            nwm, n = rowmap.shape
            if tokens != nwm * n or rowmap.dtype != torch.int32 or not rowmap.is_contiguous():
                raise ValueError("rowmap must be a contiguous int32 [nW, N] with nW * N tokens per sample")
            b_ = bsz * nwm
            out = torch.empty(bsz, tokens, c, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(b_, num_heads, n, dtype=torch.float32, device=qkv.device)
        nw = region.shape[0] if region is not None else 1
        # (a trainable table -- SGD over all parameters -- needs the one-pass backward, which bins its gradient in LDS)
        bf16 = bool(WMSA_BF16 and lib().vitta_wmsa_bf16_supported(n, hd, table.shape[0])
                    and (not table.requires_grad or lib().vitta_wmsa_bf16_dtable_supported(n, hd, table.shape[0])))
        if io16 and not bf16:
            raise _lib.VittaHipError("a bfloat16 qkv needs the bf16-operand attention kernels (ops.wmsa_io16_ok)")
        tm = KTIMING("wmsa_bf16" if bf16 else "wmsa_f32", 4.0 * n * n * hd * b_ * num_heads, 8.0 * n * n * b_ * num_heads) if KTIMING is not None else None
        if bf16:
            check(lib().vitta_wmsa_rel_fwd_bf16_io(_p(qkv), _p(table), table.shape[0], _p(code), int(code_off), _p(region), nw,
                                                   b_, n, num_heads, hd, float(scale), _p(rowmap), nwm, tokens, _p(out), _p(lse),
                                                   int(io16), _stream()), "vitta_wmsa_rel_fwd_bf16")
        else:
            check(lib().vitta_wmsa_rel_fwd_f32(_p(qkv), _p(table), table.shape[0], _p(code), int(code_off), _p(region), nw,
                                               b_, n, num_heads, hd, float(scale), _p(rowmap), nwm, tokens, _p(out), _p(lse),
                                               _stream()), "vitta_wmsa_rel_fwd_f32")
        if tm is not None:
            tm.stop()
        ctx.save_for_backward(qkv, table, code, region, rowmap, out, lse)
        ctx.meta = (int(code_off), float(scale), num_heads, hd, nw, b_, n, nwm, tokens, bf16)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, table, code, region, rowmap, out, lse = ctx.saved_tensors
        off, scale, nh, hd, nw, b_, n, nwm, tokens, bf16 = ctx.meta
        dout = dout.contiguous()
        if dout.dtype != qkv.dtype:
            dout = dout.to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        # algorithmic flops of the backward: S, dP, dV, dK, dQ once each = 10 N^2 d (until round 4 the line counted the 14 N^2 d the two-kernel
        # form executes: S and dP twice); vector lane-operations per score: ~8 forward, ~10 backward (bias / mask terms, exp, dS, packing)
        tm = KTIMING("wmsa_bf16" if bf16 else "wmsa_f32", 10.0 * n * n * hd * b_ * nh, 10.0 * n * n * b_ * nh) if KTIMING is not None else None
        dtable, r_table = _grad_sink(table, ctx.needs_input_grad[1])
        if bf16:
            ws = None
            if dtable is not None and os.environ.get("VITTA_DTABLE_WS", "1") != "0":
                # scratch for the pairs' table-gradient columns (plain stores + one reduce launch instead of atomics): one grow-only
                # buffer per stream, made outside any capture the first time (contents are undefined before and after a call)
                need = int(lib().vitta_wmsa_bf16_dtable_workspace_bytes(b_, nh, table.shape[0])) // 4
                key = (qkv.device.index, torch.cuda.current_stream(qkv.device).cuda_stream)
                ws = _dtable_ws.get(key)
                if ws is None or ws.numel() < need:
                    ws = _dtable_ws[key] = torch.empty(need, dtype=torch.float32, device=qkv.device)
            check(lib().vitta_wmsa_rel_bwd_bf16_io(_p(qkv), _p(table), table.shape[0], _p(code), off, _p(region), nw, b_, n, nh,
                                                   hd, scale, _p(rowmap), nwm, tokens, _p(out), _p(dout), _p(lse), _p(delta),
                                                   _p(dqkv), _p(dtable), _p(ws), ws.numel() * 4 if ws is not None else 0,
                                                   int(qkv.dtype == torch.bfloat16), _stream()), "vitta_wmsa_rel_bwd_bf16")
            if tm is not None:
                tm.stop()
            return dqkv, r_table, None, None, None, None, None, None
        check(lib().vitta_wmsa_rel_bwd_f32(_p(qkv), _p(table), table.shape[0], _p(code), off, _p(region), nw, b_, n, nh,
                                           hd, scale, _p(rowmap), nwm, tokens, _p(out), _p(dout), _p(lse), _p(delta),
                                           _p(dqkv), _p(dtable), _stream()), "vitta_wmsa_rel_bwd_f32")
        if tm is not None:
            tm.stop()
        return dqkv, r_table, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# fused eval-mode BatchNorm2d (+ residual) (+ ReLU) (+ ViTTA statistics)
# ------------------------------------------------------------------------------------------------
def bn_act_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and (x.shape[1] * x.shape[2] * x.shape[3]) % 4 == 0


def _bn_nsplit(outer, c, hw):
    nchunks = (c * hw + 1023) // 1024
    return max(1, min(outer, -(-1024 // nchunks)))


class FusedBNAct(torch.autograd.Function):
    """z = act(batch_norm_eval(x) + residual) in one pass; with `site` (a hooked layer of the batched
    engine) the per-channel moments of the BN output come out of the same pass and the backward adds the
    statistics-loss gradient; backward = ReLU mask + injection + BN backward (dx, dgamma, dbeta) in one pass."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, residual, relu, site, fork=False):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        outer, c, h, w = x.shape
        hw = h * w
        res = residual.contiguous() if residual is not None else None
        z = torch.empty_like(x)
        triples = C.c_void_p(0)
        if site is not None:
            nsplit, triples_ptr = site.begin(x)
            triples = C.c_void_p(triples_ptr)
        else:
            nsplit = _bn_nsplit(outer, c, hw)
        check(lib().vitta_bn_act_fwd_f32(_p(x), _p(res), _p(z), _p(weight), _p(bias), _p(running_mean), _p(running_var),
                                         float(eps), outer, c, hw, nsplit, int(bool(relu)), triples, _stream()),
              "vitta_bn_act_fwd_f32")
        ctx.save_for_backward(x, z if (relu and res is not None) else None, weight, bias, running_mean, running_var)
        ctx.meta = (float(eps), bool(relu), res is not None, site, nsplit)
        if fork:
            # a second handle on the same storage: the consumer of the identity path takes this one, so the two
            # gradients of z arrive separately and the backward kernel sums them while it reads them
            twin = torch.empty(0, dtype=z.dtype, device=z.device).set_(z.untyped_storage(), z.storage_offset(), z.shape,
                                                                       z.stride())
            return z, twin
        return z

    @staticmethod
    def backward(ctx, gz, gz2=None):
        x, z, weight, bias, running_mean, running_var = ctx.saved_tensors
        eps, relu, has_res, site, nsplit = ctx.meta
        outer, c, h, w = x.shape
        hw = h * w
        if gz is None:
            gz, gz2 = gz2, None
        if gz is None:
            gz = torch.zeros_like(x)
        gz = gz.contiguous()
        gz2 = gz2.contiguous() if gz2 is not None else None
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gres = torch.empty_like(x) if has_res else None
        dgamma, ret_gamma = _grad_sink(weight, ctx.needs_input_grad[1], zero=False)
        dbeta, ret_beta = _grad_sink(bias, ctx.needs_input_grad[2], zero=False)
        accumulate = ret_gamma is None and ret_beta is None and dgamma is not None and dbeta is not None
        if not accumulate:  # frozen affine, or only one of the two has a live .grad: plain outputs
            dgamma = ret_gamma = torch.empty_like(weight)
            dbeta = ret_beta = torch.empty_like(bias)
            if not ctx.needs_input_grad[1]:
                ret_gamma = None
            if not ctx.needs_input_grad[2]:
                ret_beta = None
        partial = None
        if not accumulate:
            nfl = lib().vitta_bn_act_partial_floats(outer, c, hw, nsplit)
            partial = torch.empty(nfl, dtype=torch.float32, device=x.device)
        mu = ca = cb = gs = None
        if site is not None:
            mu, ca, cb, gs = site.coefficients()
        check(lib().vitta_bn_act_bwd_f32(_p(x), _p(z), _p(gz), _p(gz2), _p(gx), _p(gres), _p(weight), _p(bias), _p(running_mean),
                                         _p(running_var), eps, _p(mu), _p(ca), _p(cb), _p(gs), outer, c, hw, nsplit,
                                         int(relu), _p(partial), _p(dgamma), _p(dbeta), int(accumulate), _stream()),
              "vitta_bn_act_bwd_f32")
        return gx, ret_gamma, ret_beta, None, None, None, gres, None, None, None


# ------------------------------------------------------------------------------------------------
# TAM branches (G and L) fused
# ------------------------------------------------------------------------------------------------
def tam_branch_supported(c, t):
    return bool(lib().vitta_tam_branch_supported(int(c), int(t)))


_fused_ok = {}


def tam_branch_fused_supported(n, c, t):
    """The one-launch TAM branch passes hold every workgroup resident for n clips (`vitta_tam_branch_fused_supported`)."""
    key = (int(n), int(c), int(t))
    hit = _fused_ok.get(key)
    if hit is None:
        hit = _fused_ok[key] = bool(lib().vitta_tam_branch_fused_supported(*key))
    return hit


def _ptr4(*tensors):
    arr = (C.c_void_p * 4)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


class TamFused(torch.autograd.Function):
    """The whole temporal adaptive module as ONE autograd node: pool -> G/L branches -> adaptive aggregation
    (temporal_module.py:43-65).  Same kernels as TamPool, the G/L branch launches and TamAggregate; what the single node
    saves is autograd's glue around them in the backward pass: x feeds both the pooling and the aggregation, so as
    separate nodes their two full-size input gradients meet in a zero-fill + an add per TAM (16 of each per step,
    on the largest tensors of the network); here vitta_tam_pool_bwd accumulates into the aggregation's dx."""

    @staticmethod
    def forward(ctx, x, n_segment, wg1, bng_w, bng_b, wg3, w0, bnl_w, bnl_b, w3, bng_rm, bng_rv, eps_g, bnl_rm, bnl_rv,
                eps_l):
        _require_cuda_f32(x, "x")
        x = x.contiguous()
        nt, c, h, w = x.shape
        t = int(n_segment)
        n, hw = nt // t, h * w
        if not (w0.is_contiguous() and w3.is_contiguous() and wg1.is_contiguous() and wg3.is_contiguous()):
            raise ValueError("TAM weights must be contiguous")
        f = dict(dtype=torch.float32, device=x.device)
        pooled = torch.empty(n, c, t, **f)
        kern = torch.empty(n * c, 3, **f)
        gate = torch.empty(n, c, t, **f)
        hpre = torch.empty(2, n, c // 4, t, **f)
        out = torch.empty_like(x)
        st = _stream()
        check(lib().vitta_tam_pool_f32(_p(x), n, t, c, hw, _p(pooled), st), "vitta_tam_pool_f32")
        check(lib().vitta_tam_branch_fwd_f32(_p(pooled), _p(wg1), _ptr4(bng_w, bng_b, bng_rm, bng_rv), float(eps_g),
                                             _p(wg3), _p(w0), _ptr4(bnl_w, bnl_b, bnl_rm, bnl_rv), float(eps_l), _p(w3),
                                             n, c, t, _p(kern), _p(gate), _p(hpre), 0, st), "vitta_tam_branch_fwd_f32")
        check(lib().vitta_tam_agg_fwd_f32(_p(x), _p(gate), _p(kern), n, t, c, hw, _p(out), st), "vitta_tam_agg_fwd_f32")
        ctx.save_for_backward(x, pooled, wg1, bng_w, bng_b, wg3, w0, bnl_w, bnl_b, w3, bng_rm, bng_rv, bnl_rm, bnl_rv,
                              kern, gate, hpre)
        ctx.meta = (n, t, c, hw, float(eps_g), float(eps_l))
        return out

    @staticmethod
    def backward(ctx, gout):
        (x, pooled, wg1, bng_w, bng_b, wg3, w0, bnl_w, bnl_b, w3, bng_rm, bng_rv, bnl_rm, bnl_rv, kern, gate,
         hpre) = ctx.saved_tensors
        n, t, c, hw, eps_g, eps_l = ctx.meta
        f = dict(dtype=torch.float32, device=x.device)
        st = _stream()
        gout = gout.contiguous()
        gx = torch.empty_like(x)
        ggate = torch.empty(n * c * t * 4, **f)  # the kernel's per-row partials, reduced in place into the head
        gkern = torch.empty_like(kern)
        check(lib().vitta_tam_agg_bwd_f32(_p(x), _p(gate), _p(kern), _p(gout), n, t, c, hw, _p(gx), _p(ggate), _p(gkern),
                                          st), "vitta_tam_agg_bwd_f32")
        need = ctx.needs_input_grad
        (dgw, r_gw), (dgb, r_gb), (dlw, r_lw), (dlb, r_lb) = (_grad_sink(v, True) for v in (bng_w, bng_b, bnl_w, bnl_b))
        r_gw, r_gb, r_lw, r_lb = (r if nd else None for r, nd in zip((r_gw, r_gb, r_lw, r_lb),
                                                                    (need[3], need[4], need[7], need[8])))
        (dwg1, r_wg1), (dwg3, r_wg3), (dw0, r_w0), (dw3, r_w3) = (
            _grad_sink(v, nd) for v, nd in zip((wg1, wg3, w0, w3), (need[2], need[5], need[6], need[9])))
        gbuf = torch.empty(n * c * t + n * (c // 4) * t, **f)  # d pooled | scratch: d(conv1 output)
        check(lib().vitta_tam_branch_bwd_f32(_p(pooled), _p(wg1), _ptr4(bng_w, bng_b, bng_rm, bng_rv), eps_g, _p(wg3),
                                             _p(w0), _ptr4(bnl_w, bnl_b, bnl_rm, bnl_rv), eps_l, _p(w3), n, c, t,
                                             _p(kern), _p(gate), _p(hpre), _p(gkern), _p(ggate), _p(gbuf),
                                             _ptr4(dgw, dgb, dlw, dlb), _ptr4(dwg1, dwg3, dw0, dw3), 0, st),
              "vitta_tam_branch_bwd_f32")
        check(lib().vitta_tam_pool_bwd_f32(_p(gbuf), n, t, c, hw, _p(gx), st), "vitta_tam_pool_bwd_f32")
        return (gx, None, r_wg1, r_gw, r_gb, r_wg3, r_w0, r_lw, r_lb, r_w3, None, None, None, None, None, None)


class ResidualDropPath(torch.autograd.Function):
    """out = x + scale_b * branch in one pass (scale: one value per sample on the device, bernoulli(keep)/keep of
    timm's DropPath; None = 1).  Backward: dx = g (the same tensor), dbranch = scale_b * g."""

    @staticmethod
    def forward(ctx, x, branch, scale):
        _require_cuda_f32(x, "x")
        x, branch = x.contiguous(), branch.contiguous()
        out = torch.empty_like(x)
        b = x.shape[0]
        check(lib().vitta_scale_add_f32(_p(x), _p(branch), _p(scale), b, x.numel() // b, _p(out), _stream()),
              "vitta_scale_add_f32")
        ctx.save_for_backward(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        if scale is None:
            return g, g, None
        g = g.contiguous()
        gb = torch.empty_like(g)
        b = g.shape[0]
        check(lib().vitta_scale_add_f32(None, _p(g), _p(scale), b, g.numel() // b, _p(gb), _stream()), "vitta_scale_add_f32")
        return g, gb, None


# ------------------------------------------------------------------------------------------------
# LayerNorm (+ residual / stochastic depth) (+ ViTTA statistics), channels-last rows
# ------------------------------------------------------------------------------------------------
def ln_supported(c):
    return bool(lib().vitta_ln_supported(int(c)))


class ColsumQueue:
    """Column sums of LayerNorm partial rows (vitta_colsum2_f32) put off to the point where their results are first read, then
    issued as ONE launch per 32 sites (vitta_colsum2_multi_f32): ~100 five-microsecond launches per Video Swin step leave the
    dependent chain.  The queue keeps the partial rows and the outputs alive until flush()."""

    def __init__(self):
        self.items = []

    def add(self, partial, nb, c, out_a, out_b, cnt=None, cnt_value=0.0):
        self.items.append((partial, int(nb), int(c), out_a, out_b, cnt, float(cnt_value)))

    def clear(self):
        self.items = []

    def flush(self):
        """Issue the queued sums on the CURRENT stream (the one their partial rows were written on)."""
        if not self.items:
            return
        items, self.items = self.items, []
        arr = (_lib.ColsumItem * len(items))()
        for a, (partial, nb, c, out_a, out_b, cnt, cv) in zip(arr, items):
            a.d_partial, a.n_partials, a.C, a.cnt_value = partial.data_ptr(), nb, c, cv
            a.d_out_a, a.d_out_b, a.d_cnt = out_a.data_ptr(), out_b.data_ptr(), (cnt.data_ptr() if cnt is not None else None)
        check(lib().vitta_colsum2_multi_f32(arr, len(items), _stream()), "vitta_colsum2_multi_f32")


# d gamma / d beta sums of the LayerNorm backward passes running inside `deferred_grad_colsums()` (tta.ViTTAAdapter._backward):
# queued while the sink is live `.grad` storage (nothing else adds to it before the flush), flushed when the context ends or a
# gradient bucket is about to leave (flush_grad_colsums)
_GRAD_COLSUMS = ColsumQueue()
_grad_colsums_deferred = False
DEFER_COLSUMS = os.environ.get("VITTA_DEFER_COLSUMS", "1") != "0"


class deferred_grad_colsums:
    def __enter__(self):
        global _grad_colsums_deferred
        self.prev = _grad_colsums_deferred
        if not self.prev:
            _GRAD_COLSUMS.clear()  # (a backward that raised leaves nothing behind)
        _grad_colsums_deferred = DEFER_COLSUMS
        return self

    def __exit__(self, et, ev, tb):
        global _grad_colsums_deferred
        _grad_colsums_deferred = self.prev
        if et is None:
            _GRAD_COLSUMS.flush()
        else:
            _GRAD_COLSUMS.clear()
        return False


def flush_grad_colsums():
    _GRAD_COLSUMS.flush()


class FusedLayerNorm(torch.autograd.Function):
    """y = LayerNorm_C(x') with x' = x + scale_b * branch (branch optional), one pass; a hooked layer (`site`) leaves the
    shifted channel sums of y in the engine's statistics buffer, and its backward adds the statistics-loss gradient.
    Returns y, or (x', y) with a branch.  Backward: one pass for dx (+ the gradient arriving at x'), d branch, and
    per-workgroup d gamma / d beta partials, then one column sum.
    passthrough (no branch): returns (x, y) -- the caller hands the returned x to the block's second reader (the residual
    update), so that x has ONE consumer in the autograd graph and the two gradients meet inside the backward pass instead of in
    an accumulation launch over the whole residual stream (the first block of a stage, whose norm1 no previous pass has fused)."""

    @staticmethod
    def forward(ctx, x, branch, scale, weight, bias, eps, site, y_bf16=False, passthrough=False):
        """y_bf16: the normalised output only feeds a dense product of the bf16 recipe -- written as bfloat16 (the statistics of a
        hooked layer are taken from the fp32 values in registers; x' stays fp32).  A bfloat16 `branch` (what such a product wrote)
        is read as it is; its gradient leaves the backward as bfloat16."""
        ctx.set_materialize_grads(False)
        _require_cuda_f32(x, "x")
        x = x.contiguous()
        c = x.shape[-1]
        rows = x.numel() // c
        rps = rows // x.shape[0]
        f = dict(dtype=torch.float32, device=x.device)
        y = torch.empty_like(x, dtype=torch.bfloat16 if y_bf16 else torch.float32)
        mean, rstd = torch.empty(rows, **f), torch.empty(rows, **f)
        xnew = None
        flags = _lib.LN_Y_BF16 if y_bf16 else 0
        if branch is not None:
            if branch.dtype not in (torch.float32, torch.bfloat16) or not branch.is_cuda:
                raise _lib.VittaHipError(f"branch must be a float32 or bfloat16 device tensor (got {branch.dtype})")
            branch = branch.contiguous()
            xnew = torch.empty_like(x)
            flags |= _lib.LN_BRANCH_BF16 if branch.dtype == torch.bfloat16 else 0
        shift = partial = None
        nb = int(lib().vitta_ln_num_partials(rows))
        if site is not None:
            shift, s1, s2, cnt = site.begin(rows, c)
            partial = torch.empty(nb * 2 * c, **f)
        check(lib().vitta_ln_fwd_mixed(_p(x), _p(branch), _p(scale), rows, rps, c, _p(weight), _p(bias), float(eps), _p(xnew),
                                       _p(y), _p(mean), _p(rstd), _p(shift), _p(partial), flags, _stream()), "vitta_ln_fwd_mixed")
        if site is not None:
            q = site.colsum_queue() if (DEFER_COLSUMS and hasattr(site, "colsum_queue")) else None
            if q is not None:  # the engine issues the sums of all hooked layers in one launch before it reads them
                q.add(partial, nb, c, s1, s2, cnt, float(rows))
            else:
                check(lib().vitta_colsum2_f32(_p(partial), nb, c, _p(s1), _p(s2), _p(cnt), float(rows), _stream()),
                      "vitta_colsum2_f32")
        ctx.save_for_backward(xnew if xnew is not None else x, mean, rstd, weight, bias, scale)
        ctx.meta = (rows, rps, c, site, nb, branch is not None, branch is not None and branch.dtype == torch.bfloat16,
                    passthrough and branch is None)
        if branch is not None:
            return xnew, y
        return (x.view(x.shape), y) if passthrough else y

    @staticmethod
    def backward(ctx, *grads):
        xn, mean, rstd, weight, bias, scale = ctx.saved_tensors
        rows, rps, c, site, nb, has_branch, branch16, passthrough = ctx.meta
        g_xnew, gy = (grads[0], grads[1]) if (has_branch or passthrough) else (None, grads[0])
        if gy is None:  # the normalised output took no part in the loss: only the residual path carries a gradient
            gb = None
            if has_branch and g_xnew is not None:
                gb = g_xnew if scale is None else g_xnew * scale.view((-1,) + (1,) * (g_xnew.dim() - 1))
                gb = gb.to(torch.bfloat16) if branch16 else gb
            return g_xnew, gb, None, None, None, None, None, None, None
        f = dict(dtype=torch.float32, device=xn.device)
        gy = gy.contiguous()
        g_xnew = g_xnew.contiguous() if g_xnew is not None else None
        gx = torch.empty_like(xn)
        # the branch gradient is its own tensor when it is scaled (stochastic depth) or leaves as bfloat16
        gbranch = torch.empty_like(xn, dtype=torch.bfloat16 if branch16 else torch.float32) \
            if (has_branch and (scale is not None or branch16)) else None
        partial = torch.empty(nb * 2 * c, **f)
        mu = ca = cb = gs = None
        if site is not None:
            mu, ca, cb, gs = site.coefficients()
        flags = (_lib.LN_GY_BF16 if gy.dtype == torch.bfloat16 else 0) | (_lib.LN_GBRANCH_BF16 if branch16 else 0)
        check(lib().vitta_ln_bwd_mixed(_p(gy), _p(g_xnew), _p(xn), _p(mean), _p(rstd), _p(weight), _p(bias), _p(scale), _p(mu),
                                       _p(ca), _p(cb), _p(gs), rows, rps, c, _p(gx), _p(gbranch), _p(partial), flags, _stream()),
              "vitta_ln_bwd_mixed")
        r_w = r_b = None
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            dw, r_w = _grad_sink(weight, True)   # live .grad storage, or a zeroed buffer: the column sum adds into it
            db, r_b = _grad_sink(bias, True)
            if _grad_colsums_deferred and r_w is None and r_b is None:
                _GRAD_COLSUMS.add(partial, nb, c, dw, db)
            else:
                check(lib().vitta_colsum2_f32(_p(partial), nb, c, _p(dw), _p(db), None, 0.0, _stream()), "vitta_colsum2_f32")
            if not ctx.needs_input_grad[3]:
                r_w = None
            if not ctx.needs_input_grad[4]:
                r_b = None
        g_branch = (gbranch if gbranch is not None else gx) if has_branch else None
        return gx, g_branch, None, r_w, r_b, None, None, None, None


# ------------------------------------------------------------------------------------------------
# ResNet stem tail: eval BN -> ReLU -> MaxPool(3, 2, 1) in one pass (affine-only backward)
# ------------------------------------------------------------------------------------------------
class FusedStemPool(torch.autograd.Function):
    """pooled = maxpool3x3s2p1(relu(bn_eval(x))) for an x that needs NO gradient (the stem convolution is frozen and
    its input is the video): the backward only accumulates d gamma / d beta, recomputing the window maxima."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps):
        _require_cuda_f32(x, "x")
        x = x.contiguous()
        n, c, h, w = x.shape
        out = torch.empty(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, dtype=torch.float32, device=x.device)
        check(lib().vitta_stem_bn_relu_pool_fwd_f32(_p(x), _ptr4(weight, bias, running_mean, running_var), float(eps), n, c, h,
                                                    w, _p(out), _stream()), "vitta_stem_bn_relu_pool_fwd_f32")
        ctx.save_for_backward(x, weight, bias, running_mean, running_var)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, bias, running_mean, running_var = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise RuntimeError("FusedStemPool has no input gradient: use the unfused stem when the convolution below trains")
        n, c, h, w = x.shape
        dw, r_w = _grad_sink(weight, True)
        db, r_b = _grad_sink(bias, True)
        check(lib().vitta_stem_bn_relu_pool_bwd_affine_f32(_p(x), _p(gout.contiguous()), _ptr4(weight, bias, running_mean,
                                                                                               running_var), ctx.eps, n, c, h, w,
                                                           _p(dw), _p(db), _stream()), "vitta_stem_bn_relu_pool_bwd_affine_f32")
        return (None, r_w if ctx.needs_input_grad[1] else None, r_b if ctx.needs_input_grad[2] else None, None, None, None)


class HeadLinear(torch.autograd.Function):
    """y = x @ weight.T + bias for the handful of pooled frame features of the classification head (vitta_linear_*_f32,
    models/tanet_models/tanet.py:243-251).  Parameter gradients go straight into their `.grad` storage (_grad_sink)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_cuda_f32(x, "x")
        _require_cuda_f32(weight, "weight")
        x = x.contiguous()
        m, k = x.shape
        n = weight.shape[0]
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        check(lib().vitta_linear_fwd_f32(_p(x), _p(weight), _p(bias), m, n, k, _p(y), _stream()), "vitta_linear_fwd_f32")
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias = ctx.saved_tensors
        m, k = x.shape
        n = weight.shape[0]
        gy = gy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, r_w = _grad_sink(weight, ctx.needs_input_grad[1])
        db, r_b = _grad_sink(bias, bias is not None and ctx.needs_input_grad[2])
        check(lib().vitta_linear_bwd_f32(_p(gy), _p(x), _p(weight), m, n, k, _p(dx), _p(dw), _p(db), _stream()),
              "vitta_linear_bwd_f32")
        return dx, r_w, r_b


_tickets = {}


class TanetHead(torch.autograd.Function):
    """(video logits [B, K], loss_consis) = the TANet head of the adaptation pass from the pooled frame features [B V T, D]:
    dropout -> new_fc -> segment consensus -> compute_pred_consis over the views -> mean over the views
    (models/tanet_models/tanet.py:243-251, corpus/basics.py:640-668, utils/pred_consistency_utils.py:15-31) as ONE launch behind
    ATen's dropout forward and ONE launch backward (vitta_tanet_head_{fwd,bwd}_f32).  The segment consensus is linear, so it is
    taken before the product.  Parameter gradients go straight into their `.grad` storage (_grad_sink)."""

    @staticmethod
    def forward(ctx, feat, weight, bias, p, train, T, V):
        _require_cuda_f32(feat, "feat")
        _require_cuda_f32(weight, "weight")
        ctx.set_materialize_grads(False)
        feat = feat.contiguous()
        f, d = feat.shape
        k = weight.shape[0]
        b = f // (V * T)
        if b * V * T != f:
            raise VittaHipError(f"TanetHead: {f} frame rows are not B x {V} views x {T} segments")
        mask = None
        y = feat
        if train and p >= 1.0:
            raise VittaHipError("TanetHead: dropout p must be < 1 (tanet.TSN.fused_head_ok sends p = 1 to the module chain)")
        if train and p > 0.0:
            y, mask = torch.native_dropout(feat, float(p), True)
        dev = feat.device
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        ybar, lv, out, loss, gradc = e(b * V, d), e(b * V, k), e(b, k), e(1), e(b * V, k)
        from . import conv as CV
        ticket = CV.zeroed_per_stream(_tickets, dev, 64, spares=4)
        check(lib().vitta_tanet_head_fwd_f32(_p(y), _p(weight), _p(bias), b, V, T, k, d, _p(ybar), _p(lv), _p(ticket), _p(out), _p(loss),
                                             _p(gradc), _stream()), "vitta_tanet_head_fwd_f32")
        ctx.save_for_backward(weight, bias, mask, gradc, ybar)
        ctx.dims = (b, V, T, k, d, (1.0 / (T * (1.0 - float(p)))) if mask is not None else 1.0 / T)
        return out, loss.reshape(())

    @staticmethod
    def backward(ctx, g_out, g_loss):
        weight, bias, mask, gradc, ybar = ctx.saved_tensors
        b, V, T, k, d, scale = ctx.dims
        dev = gradc.device
        dfeat = torch.empty(b * V * T, d, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        if g_out is not None:
            g_out = g_out.contiguous()
        if g_loss is not None:
            g_loss = g_loss.reshape(1)
        dw, r_w = _grad_sink(weight, ctx.needs_input_grad[1])
        db, r_b = _grad_sink(bias, bias is not None and ctx.needs_input_grad[2])
        dl = torch.empty(b * V, k, dtype=torch.float32, device=dev) if (dw is not None or db is not None) else None
        if dfeat is None:  # (the kernel always writes the feature gradient)
            dfeat = torch.empty(b * V * T, d, dtype=torch.float32, device=dev)
        check(lib().vitta_tanet_head_bwd_f32(_p(gradc), _p(g_loss), _p(g_out), _p(weight), _p(mask), float(scale), b, V, T, k, d, _p(dfeat),
                                             _p(ybar), _p(dl), _p(dw), _p(db), _stream()), "vitta_tanet_head_bwd_f32")
        return (dfeat if ctx.needs_input_grad[0] else None), r_w, r_b, None, None, None, None


def tanet_head_eval(feat, weight, bias, b, v, t):
    """Video logits [B, K] of an evaluation pass from the pooled frame features [B V T, D] (no dropout in eval()): new_fc -> segment
    consensus -> mean over the V crops x clips (models/tanet_models/tanet.py:243-251, corpus/basics.py:700-705) with the forward
    launch of TanetHead (its consistency outputs go to scratch nobody reads)."""
    _require_cuda_f32(feat, "feat")
    feat = feat.contiguous()
    f, d = feat.shape
    k = weight.shape[0]
    if b * v * t != f:
        raise VittaHipError(f"tanet_head_eval: {f} frame rows are not {b} x {v} views x {t} segments")
    dev = feat.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    lv, out, loss, gradc = e(b * v, k), e(b, k), e(1), e(b * v, k)
    from . import conv as CV
    ticket = CV.zeroed_per_stream(_tickets, dev, 64, spares=4)
    check(lib().vitta_tanet_head_fwd_f32(_p(feat), _p(weight), _p(bias), b, v, t, k, d, None, _p(lv), _p(ticket), _p(out), _p(loss),
                                         _p(gradc), _stream()), "vitta_tanet_head_fwd_f32")
    return out


def tanet_head_supported(feat, linear, b, v):
    return (feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 2 and feat.shape[1] % 4 == 0 and b * v <= 8
            and linear.weight.dtype == torch.float32 and not linear._forward_hooks and not linear._forward_pre_hooks
            and int(lib().vitta_tanet_head_lds_bytes(b, v, linear.weight.shape[0], feat.shape[1])) <= 128 * 1024)


_UNIT = {}


def unit_gradient(device):
    """A cached scalar 1.0 on `device`: `loss.backward(gradient=unit_gradient(dev))` spares autograd's ones_like launch, and
    WeightedLoss.backward recognises it (its two upstream gradients for d loss = 1 were written by the forward launch).  Created
    on first use -- outside any graph capture (the eager warm-up steps come first)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else -1))
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


class WeightedLoss(torch.autograd.Function):
    """total = la * loss_a + lb * loss_b (corpus/basics.py:668) as one launch that also leaves BOTH upstream gradients for d total = 1
    -- the first one in `slot` (the statistics engine's device scalar `gscale`, which the injection kernels read: norm_stats._LossReg
    then finds its gradient already in place).  Backward: nothing to launch when the incoming gradient is ops.unit_gradient (what the
    adapter's `backward` call passes); one launch otherwise."""

    @staticmethod
    def forward(ctx, loss_a, loss_b, la, lb, slot):
        _require_cuda_f32(loss_a, "loss_a")
        ctx.set_materialize_grads(False)
        dev = loss_a.device
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ga = slot if slot is not None else torch.empty(1, dtype=torch.float32, device=dev)
        gb = torch.empty(1, dtype=torch.float32, device=dev) if loss_b is not None else None
        check(lib().vitta_loss_axpby_f32(_p(loss_a.reshape(1)), _p(None if loss_b is None else loss_b.reshape(1)), float(la), float(lb), _p(out),
                                         _p(ga), _p(gb), _stream()), "vitta_loss_axpby_f32")
        ctx.la, ctx.lb, ctx.ga, ctx.gb = float(la), float(lb), ga, gb
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        ga, gb = ctx.ga, ctx.gb
        unit = _UNIT.get((ga.device.type, ga.device.index))
        if g is None or unit is None or g.data_ptr() != unit.data_ptr():
            check(lib().vitta_loss_axpby_bwd_f32(_p(None if g is None else g.reshape(1)), ctx.la, ctx.lb, _p(ga), _p(gb), _stream()),
                  "vitta_loss_axpby_bwd_f32")
        return ga.reshape(()), (gb.reshape(()) if gb is not None else None), None, None, None


def head_linear_supported(x, linear):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] % 4 == 0 and x.shape[0] <= 4096
            and linear.weight.dtype == torch.float32 and not linear._forward_hooks and not linear._forward_pre_hooks)


# ------------------------------------------------------------------------------------------------
# dense layers of Video Swin-B (csrc/gemm.hip): qkv / proj / Mlp / PatchMerging.reduction
# ------------------------------------------------------------------------------------------------
# callable(kind, flops) -> object with .stop(), or None (the product never sets it): bench.py brackets the dense and window
# attention launches of an eager repeat with stream events to report their achieved TFLOP/s
KTIMING = None
GEMM_TILE = 0       # 0: the library's choice; tools force 1 (128 x 128) / 2 (64 x 128) / 3 (64 x 64)
DENSE_BF16 = False  # opt-in (--dense_bf16): bf16 MFMA operands for the dense layers, fp32 accumulation / epilogues

_W_CACHE = {}       # (id(weight), transposed, bf16) -> (weakref, version, operand copy): frozen weights are prepared once


def gemm_nt_supported(m, n, k):
    return bool(lib().vitta_gemm_nt_supported(m, n, k))


def bf16_flow():
    """The bf16 recipe's DATA FLOW (round 4; BASELINE config 5, recognizer3d.py:36-40): with --dense_bf16 the activations between a
    LayerNorm and the product it feeds, inside the MLP (fc1 -> GELU -> fc2, pre-activation included) and the MLP's branch into the
    residual update are bfloat16 in memory, and those products run on gemm_bf16x.hip (both operands by LDS-DMA).  The residual
    stream, every statistic, softmax and accumulator stay fp32.  VITTA_BF16_FLOW=0: bf16 operands only (the round-3 form: fp32
    activations rounded while staged)."""
    return DENSE_BF16 and BF16_FLOW


BF16_FLOW = __import__("os").environ.get("VITTA_BF16_FLOW", "1") != "0"


def gemm_bf16x_supported(m, n, k):
    return bool(lib().vitta_gemm_bf16x_supported(int(m), int(n), int(k)))


def gemm_bf16x(a, b, bias=None, mode=0, aux=None, pre=None, out_bf16=False):
    """The same product on bfloat16 operands IN MEMORY (gemm_bf16x.hip): a [M][K], b [N][K] bf16; y fp32 or bf16; mode 1 keeps the
    pre-activation in `pre` (bf16), mode 2 multiplies by gelu'(aux) with aux the bf16 pre-activation."""
    if not (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.is_contiguous() and b.is_contiguous()):
        raise _lib.VittaHipError("gemm_bf16x: contiguous bfloat16 device operands")
    m, k = a.shape
    n = b.shape[0]
    assert b.shape[1] == k
    y = torch.empty(m, n, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
    tm = KTIMING("gemm_bf16", 2.0 * m * n * k) if KTIMING is not None else None
    check(lib().vitta_gemm_nt_bf16x(_p(a), _p(b), _p(bias), _p(aux), _p(y), _p(pre), m, n, k, int(mode), int(bool(out_bf16)), _stream()),
          "vitta_gemm_nt_bf16x")
    if tm is not None:
        tm.stop()
    return y


# stream-K form of the fp32 product (vitta_gemm_nt_sk_f32) where the 64 x 64 tile count quantises badly on the 256 CUs: grid =
# GEMM_SK_GRID workgroups (0: never).  One workspace per stream, created on first use (before any capture: a step runs eagerly first).
GEMM_SK_GRID = int(os.environ.get("VITTA_GEMM_SK", "512"))
_SK_WS = {}


def gemm_sk_pays(m, n, k):
    """Fewer than 1024 tiles that leave >= 15 % of the CUs' last round idle, and a reduction long enough to cut: measured
    (tools/bench_gemm_sk.py, profiles/r5_gemm_stream_k.txt) -5..-14 % for K >= 1024, slower for K = 512 (a partial tile's trip
    through the workspace costs what a quarter of such a tile's slabs do)."""
    tiles = ((m + 63) // 64) * ((n + 63) // 64)
    if GEMM_SK_GRID <= 0 or tiles < 128 or tiles >= 1024 or k < 1024:
        return False
    rounds = -(-tiles // 256)
    return tiles / (256.0 * rounds) <= 0.85


def _sk_workspace(device):
    from . import conv
    nbytes = int(lib().vitta_gemm_nt_sk_workspace_bytes(GEMM_SK_GRID))
    return conv.zeroed_per_stream(_SK_WS, device, nbytes, spares=3)  # (zero once, eagerly; spares for the streams a capture brings)


def gemm_nt(a, b, bias=None, mode=0, aux=None, pre=None, out=None):
    """y[m][n] = epi(sum_k a[m][k] b[n][k]) (mode 0: + bias, 1: bias + GELU (pre-activation kept in `pre`), 2: times
    gelu'(aux)).  b float32: vitta_gemm_nt_f32 (exact fp32 MFMA); b bfloat16: vitta_gemm_nt_bf16w_f32 (a is rounded to
    bf16 while staged)."""
    _require_cuda_f32(a, "a")
    if not b.is_cuda or b.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.VittaHipError(f"b must be a float32 or bfloat16 device tensor (got {b.dtype} on {b.device})")
    m, k = a.shape
    n = b.shape[0]
    assert a.is_contiguous() and b.is_contiguous() and b.shape[1] == k
    y = out if out is not None else torch.empty(m, n, dtype=torch.float32, device=a.device)
    fn, name = (lib().vitta_gemm_nt_f32, "vitta_gemm_nt_f32") if b.dtype == torch.float32 else \
        (lib().vitta_gemm_nt_bf16w_f32, "vitta_gemm_nt_bf16w_f32")
    tm = KTIMING("gemm_bf16" if b.dtype == torch.bfloat16 else "gemm_f32", 2.0 * m * n * k) if KTIMING is not None else None
    if b.dtype == torch.float32 and GEMM_TILE == 0 and gemm_sk_pays(m, n, k):
        ws = _sk_workspace(a.device)
        check(lib().vitta_gemm_nt_sk_f32(_p(a), _p(b), _p(bias), _p(aux), _p(y), _p(pre), m, n, k, mode, GEMM_SK_GRID, _p(ws),
                                         ws.numel(), _stream()), "vitta_gemm_nt_sk_f32")
    else:
        check(fn(_p(a), _p(b), _p(bias), _p(aux), _p(y), _p(pre), m, n, k, mode, GEMM_TILE, _stream()), name)
    if tm is not None:
        tm.stop()
    return y


def _operand(weight, transposed, m_rows=None, force_bf16=False):
    """The B operand of a dense product: the nn.Linear weight itself ([out][in], forward) or its [in][out] transpose (data
    gradient), as bfloat16 when DENSE_BF16 is on and the reduction length allows.  A frozen weight (LN-affine adaptation)
    is prepared once per version; a trainable one on every call (the flat-arena optimizer updates storage without touching
    `_version`, and a captured graph must hold the copy launch)."""
    w2d = weight.detach().reshape(weight.shape[0], -1)  # a Conv3d patch-embedding kernel counts as [out][in * kd * kh * kw]
    kred = w2d.shape[0] if transposed else w2d.shape[1]
    nout = w2d.shape[1] if transposed else w2d.shape[0]
    # force_bf16 (gemm_bf16x's weight copies): an explicit argument, not a flip of the module global -- backward passes run on
    # autograd's worker threads; the reduction-length condition is gemm_bf16x_supported's (K % 32), the fp32-activation bf16 kernel's K % 64
    bf16 = bool((force_bf16 and kred % 32 == 0) or (DENSE_BF16 and kred % 64 == 0))
    if not transposed and not bf16:
        return w2d

    def make():
        w = w2d.t() if transposed else w2d
        return w.to(torch.bfloat16).contiguous() if bf16 else w.contiguous()

    if weight.requires_grad:
        return make()
    import weakref
    key = (id(weight), transposed, bf16)
    ent = _W_CACHE.get(key)
    if ent is not None and ent[0]() is weight and ent[1] == weight._version and ent[2].device == weight.device:
        return ent[2]
    op = make()
    _W_CACHE[key] = (weakref.ref(weight), weight._version, op)
    return op


DENSE_WGRAD = __import__("os").environ.get("VITTA_DENSE_WGRAD", "conv") != "library"


def _weight_grad_conv(g2, x2, out, accumulate):
    """out [N, K] (+)= g2^T x2 on the hand-written convolution kernel: with the TOKENS as its channel axis a pointwise
    convolution IS this product -- x2 [M, K] read as planes `[C = M][P = K]`, g2 [M, N] as the packed weight `[1][C = M][N]`,
    the channel-major output `[N][P = K]` is the nn.Linear weight layout -- and its stream-K form (conv_sk.hip) was built for
    few tiles with a long reduction.  `accumulate`: the existing content of `out` enters as the epilogue's residual
    (same address, same lane).  False when the shape does not qualify (the caller falls back to the library product)."""
    from . import conv as CV
    m, n = g2.shape
    k = x2.shape[1]
    if m % 16 or n % 32 or k % 4 or m * k * 4 >= (1 << 31) or not (g2.is_contiguous() and x2.is_contiguous() and out.is_contiguous()):
        return False
    geom = _WGRAD_GEOMS.get(k)
    if geom is None:
        geom = _WGRAD_GEOMS[k] = CV.Geometry.forward(1, 1, k)
    CV.launch(geom, x2, CV.Pack(g2, None), out.view(n, k), m, n, flags=CV.CONV_RES if accumulate else 0,
              res=out.view(n, k) if accumulate else None)
    return True


_WGRAD_GEOMS = {}


def _weight_grad(weight, needed, g2, x2):
    """g2^T x2 handed to the weight's gradient sink: accumulated IN the product when `.grad` is a live arena view, so no
    separate add pass runs over the 88 M weights of Swin-B.  The product runs on `vitta_conv_f32` (_weight_grad_conv);
    VITTA_DENSE_WGRAD=library or a shape it declines: the library's TN product."""
    if not needed:
        return None
    sink, ret = _grad_sink(weight, True, zero=False)
    if ret is None:
        if not (DENSE_WGRAD and _weight_grad_conv(g2, x2, sink.view(weight.shape[0], -1), True)):
            _said_library_wgrad(weight, g2)
            sink.view(weight.shape[0], -1).addmm_(g2.t(), x2)
        return None
    if DENSE_WGRAD and _weight_grad_conv(g2, x2, ret.view(weight.shape[0], -1), False):
        return ret.view_as(weight)
    _said_library_wgrad(weight, g2)
    return torch.mm(g2.t(), x2, out=ret.view(weight.shape[0], -1)).view_as(weight)


def _said_library_wgrad(weight, g2):
    if g2.is_cuda:
        from ._lib import loud_once
        loud_once(("dense_wgrad", tuple(weight.shape), DENSE_WGRAD), f"dense weight gradient {tuple(weight.shape)} over {g2.shape[0]} rows runs "
                  "on the vendor library's TN product, not on vitta_conv_f32" + ("" if DENSE_WGRAD else ": VITTA_DENSE_WGRAD=library (an A/B switch)"))


TN_WGRAD_BF16 = os.environ.get("VITTA_TN_WGRAD_BF16", "1") != "0"  # 0: the fp32 convolution-kernel product behind .float() copies (A/B)
_tn_ws = {}  # (device index, stream) -> grow-only scratch of the splits' partial tiles


def _wgrad_bf16(weight, w_needed, bias, b_needed, g2, x2):
    """d weight (+ d bias) of one nn.Linear from 2-byte operands: g2 [M, N], x2 [M, K] bfloat16 as the bf16 data flow leaves them
    (no .float() copies), on gemm_tn_bf16.hip.  Returns (handled, dw, db); handled False: the caller takes the fp32 path."""
    if not (TN_WGRAD_BF16 and w_needed and g2.dtype == torch.bfloat16 and x2 is not None and x2.dtype == torch.bfloat16
            and g2.is_contiguous() and x2.is_contiguous() and weight.dim() == 2):
        return False, None, None
    m, n = g2.shape
    k = x2.shape[1]
    if not lib().vitta_gemm_tn_bf16_supported(m, n, k):
        return False, None, None
    sink, ret = _grad_sink(weight, True, zero=False)
    bsink = bret = None
    if b_needed and bias is not None:
        bsink, bret = _grad_sink(bias, True, zero=True)
    need = int(lib().vitta_gemm_tn_bf16_workspace_bytes(m, n, k))
    ws = None
    if need:
        key = (g2.device.index, torch.cuda.current_stream(g2.device).cuda_stream)
        ws = _tn_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = _tn_ws[key] = torch.empty(need, dtype=torch.uint8, device=g2.device)
    check(lib().vitta_gemm_tn_bf16(_p(g2), _p(x2), _p(sink), m, n, k, 1 if ret is None else 0, _p(bsink), _p(ws), ws.numel() if ws is not None else 0,
                                   _stream()), "vitta_gemm_tn_bf16")
    return True, (None if ret is None else ret.view_as(weight)), bret


def _bias_grad(bias, needed, g2):
    """Column sums of the output gradient added into the bias gradient's sink by vitta_colsum2_f32 (rows taken in pairs:
    both of its outputs point at the sink); torch's column reduce ran at ~1.2 TB/s on these shapes."""
    if not needed:
        return None
    m, n = g2.shape
    if m % 2 == 0:
        sink, ret = _grad_sink(bias, True, zero=True)
        check(lib().vitta_colsum2_f32(_p(g2), m // 2, n, _p(sink), _p(sink), None, 0.0, _stream()), "vitta_colsum2_f32")
        return ret
    sink, ret = _grad_sink(bias, True, zero=False)
    if ret is None:
        sink.add_(g2.sum(0))
        return None
    return torch.sum(g2, 0, out=ret)


def _bf16_weight(weight, transposed):
    """bfloat16 [out][in] (or [in][out]) copy of a dense weight for gemm_bf16x (cached per version while frozen)."""
    return _operand(weight, transposed, force_bf16=True)


def _x_dtype_grad(dx, like_bf16):
    if dx is None:
        return None
    return dx if (dx.dtype == torch.bfloat16) == like_bf16 else dx.to(torch.bfloat16 if like_bf16 else torch.float32)


class DenseLinear(torch.autograd.Function):
    """F.linear(x, weight, bias) on the hand-written GEMM: forward y = x W^T + b, backward dx = dy W (the same kernel
    against the transposed weight); weight / bias gradients (SGD over all parameters only) on the convolution kernel.
    bfloat16 activations (the bf16 data flow, ops.bf16_flow): a bfloat16 x goes to gemm_bf16x.hip as it is; `out_bf16` writes
    the output as bfloat16; the data gradient runs on the same kernel whenever the incoming gradient is bfloat16 (else the
    fp32-activation kernel, rounded while staged) and leaves in x's dtype."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_bf16=False):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        ctx.x16 = x2.dtype == torch.bfloat16
        if ctx.x16:
            y = gemm_bf16x(x2, _bf16_weight(weight, False), bias, out_bf16=out_bf16)
        else:
            y = gemm_nt(x2, _operand(weight, False, x2.shape[0]), bias)
            if out_bf16:
                y = y.to(torch.bfloat16)
        ctx.save_for_backward(x2 if weight.requires_grad else None, weight, bias)
        ctx.xshape = shape
        return y.view(shape[:-1] + (weight.shape[0],))  # (weight may be a Conv3d kernel [out, ...]: its flattened form is used)

    @staticmethod
    def backward(ctx, gy):
        x2, weight, bias = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            kin = weight.numel() // weight.shape[0]
            if g2.dtype == torch.bfloat16 and gemm_bf16x_supported(g2.shape[0], kin, g2.shape[1]):
                dx = gemm_bf16x(g2, _bf16_weight(weight, True), out_bf16=ctx.x16)
            else:
                dx = _x_dtype_grad(gemm_nt(g2.float() if g2.dtype != torch.float32 else g2, _operand(weight, True, g2.shape[0])), ctx.x16)
            dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            handled, dw, db = _wgrad_bf16(weight, ctx.needs_input_grad[1], bias, bias is not None and ctx.needs_input_grad[2], g2, x2)
            if not handled:
                g32 = g2 if g2.dtype == torch.float32 else g2.float()
                x32 = x2 if (x2 is None or x2.dtype == torch.float32) else x2.float()
                dw = _weight_grad(weight, ctx.needs_input_grad[1], g32, x32)
                db = _bias_grad(bias, bias is not None and ctx.needs_input_grad[2], g32)
        else:
            dw = db = None
        return dx, dw, db, None


class FusedMlp(torch.autograd.Function):
    """fc2(gelu(fc1(x))) (swin_transformer.py:30-35 with drop = 0): bias + GELU in fc1's epilogue (the pre-activation h is
    kept for the backward), gelu'(h) in the epilogue of fc2's data gradient -- no stand-alone activation pass in either
    direction.  bfloat16 x (ops.bf16_flow): the four products run on gemm_bf16x.hip, h, gelu(h), the output and every
    gradient between them are bfloat16 in memory.  out_f32 (bfloat16 x only): the output is written as float32 -- what a reader
    outside the fused LayerNorm passes takes (the residual update that closes a stage)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, out_f32=False):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need = any(ctx.needs_input_grad)
        ctx.x16 = x2.dtype == torch.bfloat16
        train_w = w1.requires_grad or w2.requires_grad
        if ctx.x16:
            h = torch.empty(x2.shape[0], w1.shape[0], dtype=torch.bfloat16, device=x.device) if need else None
            a = gemm_bf16x(x2, _bf16_weight(w1, False), b1, mode=1, pre=h, out_bf16=True)
            y = gemm_bf16x(a, _bf16_weight(w2, False), b2, out_bf16=not out_f32)
        else:
            h = torch.empty(x2.shape[0], w1.shape[0], dtype=torch.float32, device=x.device) if need else None
            a = gemm_nt(x2, _operand(w1, False, x2.shape[0]), b1, mode=1, pre=h)
            y = gemm_nt(a, _operand(w2, False, x2.shape[0]), b2)
        ctx.save_for_backward(x2 if train_w else None, h, a if train_w else None, w1, b1, w2, b2)
        ctx.xshape = shape
        return y.view(shape[:-1] + (w2.shape[0],))

    @staticmethod
    def backward(ctx, gy):
        x2, h, a, w1, b1, w2, b2 = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        if ctx.x16:
            if g2.dtype != torch.bfloat16:
                g2 = g2.to(torch.bfloat16)
            gh = gemm_bf16x(g2, _bf16_weight(w2, True), mode=2, aux=h, out_bf16=True)
            dx = gemm_bf16x(gh, _bf16_weight(w1, True), out_bf16=True).view(ctx.xshape) if ctx.needs_input_grad[0] else None
            if any(ctx.needs_input_grad[1:]):  # SGD over all parameters: the weight-gradient products on the 2-byte operands as they are
                ok1, dw1, db1 = _wgrad_bf16(w1, ctx.needs_input_grad[1], b1, b1 is not None and ctx.needs_input_grad[2], gh, x2)
                ok2, dw2, db2 = _wgrad_bf16(w2, ctx.needs_input_grad[3], b2, b2 is not None and ctx.needs_input_grad[4], g2, a)
                if not ok1:  # (a shape gemm_tn_bf16.hip declines, or VITTA_TN_WGRAD_BF16=0: fp32 copies for the convolution-kernel product)
                    gh32, x32 = gh.float(), (x2.float() if x2 is not None else None)
                    dw1 = _weight_grad(w1, ctx.needs_input_grad[1], gh32, x32)
                    db1 = _bias_grad(b1, b1 is not None and ctx.needs_input_grad[2], gh32)
                if not ok2:
                    g32, a32 = g2.float(), (a.float() if a is not None else None)
                    dw2 = _weight_grad(w2, ctx.needs_input_grad[3], g32, a32)
                    db2 = _bias_grad(b2, b2 is not None and ctx.needs_input_grad[4], g32)
                return dx, dw1, db1, dw2, db2, None
        else:
            gh = gemm_nt(g2, _operand(w2, True, g2.shape[0]), mode=2, aux=h)
            dx = gemm_nt(gh, _operand(w1, True, g2.shape[0])).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw1 = _weight_grad(w1, ctx.needs_input_grad[1], gh, x2)
        db1 = _bias_grad(b1, b1 is not None and ctx.needs_input_grad[2], gh)
        dw2 = _weight_grad(w2, ctx.needs_input_grad[3], g2, a)
        db2 = _bias_grad(b2, b2 is not None and ctx.needs_input_grad[4], g2)
        return dx, dw1, db1, dw2, db2, None


def dense_supported(x, *linears):
    """The hand-written dense path takes fp32 device activations through hook-free nn.Linear modules with K % 32 == 0 -- or, in
    the bf16 data flow, bfloat16 activations where gemm_bf16x.hip covers the shapes (N % 128 == 0)."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.numel() > 0):
        return False
    m = x.numel() // x.shape[-1]
    k = x.shape[-1]
    x16 = x.dtype == torch.bfloat16
    for lin in linears:
        if lin.weight.dtype != torch.float32 or lin._forward_hooks or lin._forward_pre_hooks or lin.weight.shape[1] != k:
            return False
        n = lin.weight.shape[0]
        if x16 and not (gemm_bf16x_supported(m, n, k) and gemm_bf16x_supported(m, k, n)):  # forward and data gradient
            return False
        if not x16 and not gemm_nt_supported(m, n, k):
            return False
        k = n
    return True


def wmsa_io16_ok(n_tok, head_dim, table):
    """True when the window attention takes a bfloat16 qkv and returns a bfloat16 context (bf16 data flow on, the bf16-operand
    kernels enabled and covering the window; a trainable bias table where the one-pass backward carries its gradient)."""
    return bool(bf16_flow() and WMSA_BF16 and lib().vitta_wmsa_bf16_supported(int(n_tok), int(head_dim), table.shape[0])
                and (not table.requires_grad or lib().vitta_wmsa_bf16_dtable_supported(int(n_tok), int(head_dim), table.shape[0])))


def bf16_dense_ok(rows, *linears):
    """True when `linears` (applied in a chain to `rows` rows) can take a bfloat16 input on gemm_bf16x.hip, forward and backward."""
    if not bf16_flow():
        return False
    k = linears[0].weight.shape[1]
    for lin in linears:
        n = lin.weight.shape[0]
        if (lin.weight.dtype != torch.float32 or lin._forward_hooks or lin._forward_pre_hooks or lin.weight.shape[1] != k
                or not (gemm_bf16x_supported(rows, n, k) and gemm_bf16x_supported(rows, k, n))):
            return False
        k = n
    return True
