"""Host side of the hand-written convolutions (vitta_amd/csrc/conv.hip, C ABI `vitta_conv_f32`).

Reference call sites: models/tanet_models/temporal_module.py:85-106 (TemporalBottleneck) over torchvision's ResNet-50
Bottleneck (tanet.py:125-150).  Everything here is geometry and pointer plumbing; the arithmetic is the library's.

Tensors are "channel-major planes" (CM): a [C, N*H*W] fp32 matrix, pixel index p = n*H*W + h*W + w.  `to_cm` /
`from_cm` convert from / to the reference's [N, C, H, W].

Packed weights (made once per weight version, `pack_fwd` / `pack_bwd`):
    forward : [kh*kw][C][K]   (w.permute(2, 3, 1, 0))
    backward: [kh*kw][K][C]   (w.permute(2, 3, 0, 1)); the tap table supplies the flip of the transposed convolution.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import (CONV_BWD_BN, CONV_BWD_RELU, CONV_EPI_APPLY, CONV_EPI_RELU, CONV_PRO_BN_RELU, CONV_RES, CONV_RES_HALF,
                   CONV_INJ_RAW, CONV_POOL, CONV_STATS, CONV_STATS_RAW, ConvDesc, check, lib)


def to_cm(x):
    """[N, C, H, W] -> channel-major planes [C, N*H*W] (a copy)."""
    n, c, h, w = x.shape
    return x.permute(1, 0, 2, 3).reshape(c, n * h * w).contiguous()


def from_cm(x, n, h, w):
    """[C, N*H*W] -> [N, C, H, W] (a copy)."""
    return x.view(x.shape[0], n, h, w).permute(1, 0, 2, 3).contiguous()


def pack_fwd(w):
    k, c, kh, kw = w.shape
    return w.detach().permute(2, 3, 1, 0).reshape(kh * kw, c, k).contiguous()


def pack_bwd(w):
    k, c, kh, kw = w.shape
    return w.detach().permute(2, 3, 0, 1).reshape(kh * kw, k, c).contiguous()


# Arithmetic of vitta_conv_f32 launches: "b3" = split-bf16 operands on the bf16 matrix pipe (conv_b3.hip: three bf16 terms per
# fp32 operand, six products per multiply-add, fp32 accumulation -- fp32-roundoff-class error) wherever the shape qualifies,
# "f32" = the exact-fp32 MFMA kernels everywhere.  The reference computes in fp32 (cuDNN / PyTorch default, no TF32 on the
# parts it was published on); both forms are held to the same bound against fp64 (tests/test_gpu_conv.py).
ARITH = os.environ.get("VITTA_CONV_ARITH", "b3")


class Pack:
    """A packed weight: .f32 [taps][R][O] fp32 (vitta_conv_desc::w) and .b3, its split-bf16 image (::w_b3) or None."""
    __slots__ = ("f32", "b3")

    def __init__(self, f32, b3=None):
        self.f32, self.b3 = f32, b3


def b3_eligible(wp):
    return wp.dim() == 3 and wp.shape[1] % 32 == 0 and wp.shape[2] % 64 == 0


def pack_b3(wp, out=None):
    """Split-bf16 image (uint8 tensor) of a packed fp32 weight [taps][R][O] (`vitta_conv_pack_b3`)."""
    if not wp.is_cuda or wp.dtype != torch.float32 or not wp.is_contiguous():
        raise _lib.VittaHipError("pack_b3: contiguous fp32 tensor on the GPU")
    taps, r, o = wp.shape
    nbytes = int(lib().vitta_conv_pack_b3_bytes(taps, r, o))
    if nbytes == 0:
        raise _lib.VittaHipError("pack_b3: the reduction axis must be a multiple of 32")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=wp.device)
    check(lib().vitta_conv_pack_b3(C.c_void_p(wp.data_ptr()), C.c_void_p(out.data_ptr()), taps, r, o,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vitta_conv_pack_b3")
    return out


def make_pack(wp):
    """Pack of a fp32 packed weight, with the split image where the arithmetic mode and the shape allow it."""
    return Pack(wp, pack_b3(wp) if (ARITH == "b3" and b3_eligible(wp)) else None)


_derived = {}  # id(fp32 pack) -> (weakref, version, b3 image): launches handed a bare fp32 pack (tests, tools)


def _derive(wp):
    if ARITH != "b3" or not b3_eligible(wp):
        return None
    hit = _derived.get(id(wp))
    if hit is not None and hit[0]() is wp and hit[1] == wp._version:
        return hit[2]
    if len(_derived) > 256:
        for k in [k for k, v in _derived.items() if v[0]() is None]:
            del _derived[k]
    b3 = pack_b3(wp)
    _derived[id(wp)] = (weakref.ref(wp), wp._version, b3)
    return b3


def out_size(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _bn4(arr, bn):
    """bn: (gamma, beta, running_mean, running_var) tensors or None."""
    for i in range(4):
        arr[i] = bn[i].data_ptr() if bn is not None else None


class Geometry:
    """Tap table + grids of one launch (see the header comment of vitta_conv_desc)."""

    def __init__(self, n, hs, ws, hg, wg, hy, wy, taps, sstride=1, ostride=1, oa=0, ob=0, cls_ntaps=None):
        self.n, self.hs, self.ws, self.hg, self.wg, self.hy, self.wy = n, hs, ws, hg, wg, hy, wy
        self.taps, self.sstride, self.ostride, self.oa, self.ob = taps, sstride, ostride, oa, ob
        self.cls_ntaps = cls_ntaps  # parity-merged data gradient: taps per class (VITTA_CONV_PARITY4)

    @staticmethod
    def forward(n, h, w, k=1, stride=1, pad=0):
        ho, wo = out_size(h, k, stride, pad), out_size(w, k, stride, pad)
        taps = [(dh - pad, dw - pad, dh * k + dw) for dh in range(k) for dw in range(k)]
        return Geometry(n, h, w, ho, wo, ho, wo, taps, sstride=stride)

    @staticmethod
    def dgrad(n, h, w, k=1, stride=1, pad=0):
        """Launches producing d input [C, N*h*w] from d output [K, N*ho*wo] of conv(k, stride, pad) on h x w planes:
        one launch for stride 1, one per parity class of the input pixel for stride 2 (k = 3) -- or a single launch on
        the half-resolution grid for a strided pointwise convolution (its result is added at even positions by the
        consumer, VITTA_CONV_RES_HALF)."""
        ho, wo = out_size(h, k, stride, pad), out_size(w, k, stride, pad)
        if stride == 1:
            # d in[i] = sum_e d out[i + e - pad'] w[k-1-e], pad' = k - 1 - pad
            taps = [(eh - (k - 1 - pad), ew - (k - 1 - pad), (k - 1 - eh) * k + (k - 1 - ew)) for eh in range(k)
                    for ew in range(k)]
            return [Geometry(n, ho, wo, h, w, h, w, taps)]
        if stride != 2:
            raise ValueError("stride 1 or 2")
        if k == 1:
            return [Geometry(n, ho, wo, ho, wo, ho, wo, [(0, 0, 0)])]
        out = []
        for a in range(2):
            for b in range(2):
                taps = []
                for dh in range(k):
                    if (a + pad - dh) % 2:
                        continue
                    for dw in range(k):
                        if (b + pad - dw) % 2:
                            continue
                        taps.append(((a + pad - dh) // 2, (b + pad - dw) // 2, dh * k + dw))
                out.append(Geometry(n, ho, wo, (h + 1) // 2, (w + 1) // 2, h, w, taps, ostride=2, oa=a, ob=b))
        return out

    @staticmethod
    def dgrad_merged(n, h, w, k=3, stride=2, pad=1):
        """The four parity classes of Geometry.dgrad(stride 2) as ONE launch (VITTA_CONV_PARITY4; conv_b3.hip), or None
        where a class would be empty."""
        classes = Geometry.dgrad(n, h, w, k, stride, pad)
        if stride != 2 or len(classes) != 4 or any(len(g.taps) < 1 for g in classes):
            return None
        g0 = classes[0]
        taps = [t for g in classes for t in g.taps]
        if len(taps) > 9:
            return None
        return Geometry(n, g0.hs, g0.ws, g0.hg, g0.wg, h, w, taps, ostride=2, cls_ntaps=[len(g.taps) for g in classes])

    def wgrad_tables(self, device):
        """(src_off, src_mask) int32 [N * Hg * Wg] of vitta_wgrad_desc for this (forward) geometry, built once."""
        key = str(device)
        hit = getattr(self, "_wg_tables", {}).get(key)
        if hit is None:
            n = torch.arange(self.n, device=device).view(-1, 1, 1)
            i = torch.arange(self.hg, device=device).view(1, -1, 1)
            j = torch.arange(self.wg, device=device).view(1, 1, -1)
            si, sj = i * self.sstride, j * self.sstride
            off = (n * (self.hs * self.ws) + si * self.ws + sj).reshape(-1).to(torch.int32)
            mask = torch.zeros(self.n, self.hg, self.wg, dtype=torch.int32, device=device)
            for t, (dh, dw, _) in enumerate(self.taps):
                ok = ((si + dh >= 0) & (si + dh < self.hs) & (sj + dw >= 0) & (sj + dw < self.ws)).expand(self.n, -1, -1)
                mask |= ok.to(torch.int32) << t
            hit = (off.contiguous(), mask.reshape(-1).contiguous())
            if not hasattr(self, "_wg_tables"):
                self._wg_tables = {}
            self._wg_tables[key] = hit
        return hit

    def fill(self, d):
        d.N, d.Hs, d.Ws, d.Hg, d.Wg, d.Hy, d.Wy = self.n, self.hs, self.ws, self.hg, self.wg, self.hy, self.wy
        d.sstride, d.ostride, d.oa, d.ob, d.ntaps = self.sstride, self.ostride, self.oa, self.ob, len(self.taps)
        for i, (dh, dw, wt) in enumerate(self.taps):
            d.dh[i], d.dw[i], d.wt[i] = dh, dw, wt
        if self.cls_ntaps is not None:
            for i, nt in enumerate(self.cls_ntaps):
                d.cls_ntaps[i] = nt


WORKSPACE_BYTES = 48 << 20
_workspaces = {}
_spares = {}


def zeroed_per_stream(table, device, nbytes, spares=2):
    """A zero-filled uint8 buffer that belongs to the CURRENT stream (`table`: {(device index, stream): buffer}); its users
    leave it zero at rest, so it is filled exactly once -- and never inside a hipGraph capture: torch.cuda.graph captures on a
    stream of its own, and a buffer first touched there would have its zero fill RECORDED and replayed with every step (round 5
    found the 48 MB fill of the split-K workspace, 9.4 us, and the fill of the TAM meeting counters, 5.9 us, in the replayed
    adaptation chain).  A few zeroed spares are therefore made eagerly with the first buffer and handed to streams that show up
    while capturing."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = table.get(key)
    if buf is not None:
        return buf
    pool = _spares.setdefault((id(table), device.index, nbytes), [])
    if torch.cuda.is_current_stream_capturing():
        if not pool:
            raise RuntimeError("vitta_amd: a per-stream zeroed buffer was first requested inside a graph capture with no spare left; "
                               "run one eager step before capturing")
        buf = pool.pop()
    else:
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        while len(pool) < spares:
            pool.append(torch.zeros(nbytes, dtype=torch.uint8, device=device))
    table[key] = buf
    return buf


def workspace(device):
    """The split-K workspace of the CURRENT stream (zero-filled once; the kernels leave their counters at zero).  One
    per stream: launches on different streams may overlap and must not share slabs."""
    return zeroed_per_stream(_workspaces, device, WORKSPACE_BYTES)


def launch(geom, x, wp, y, c, k, flags=0, y_raw=None, res=None, pro_bn=None, epi_bn=None, bwd_bn=None, eps=1e-5,
           stats=None, bwd_x=None, bwd_mask=None, inj=None, dgamma=None, dbeta=None, tile=0, ksplit=0, pool=None):
    """One `vitta_conv_f32` launch on the current stream.  x [C, *], wp packed [taps][C][K] (fp32 tensor or Pack), y [K, *].
    stats = (shift, s1, s2); inj = (mu, a, b, gscale).
    pool: a zeroed int64 [frames, K] tensor the per-(frame, channel) means of relu(epi_bn(raw output)) are ADDED to as
    fixed-point numbers with 32 fractional bits (CONV_POOL)."""
    if isinstance(wp, Pack):
        wp, b3 = wp.f32, wp.b3
    else:
        b3 = _derive(wp) if (wp.is_cuda and wp.dtype == torch.float32 and wp.is_contiguous()) else None
    for t in (x, wp, y):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.VittaHipError("convolution operands must be contiguous fp32 tensors on the GPU (no CPU fallback)")
    d = ConvDesc()
    d.x, d.w, d.y, d.y_raw, d.res = x.data_ptr(), wp.data_ptr(), y.data_ptr(), _ptr(y_raw), _ptr(res)
    d.w_b3 = _ptr(b3) if ARITH == "b3" else None
    _bn4(d.pro_bn, pro_bn)
    _bn4(d.epi_bn, epi_bn)
    _bn4(d.bwd_bn, bwd_bn)
    d.pro_eps = d.epi_eps = d.bwd_eps = float(eps)
    if stats is not None:
        d.st_shift, d.st_s1, d.st_s2 = (t.data_ptr() for t in stats)
    d.bwd_x, d.bwd_mask = _ptr(bwd_x), _ptr(bwd_mask)
    if inj is not None:
        d.inj_mu, d.inj_a, d.inj_b, d.inj_gscale = (_ptr(t) for t in inj)
    d.dgamma, d.dbeta = _ptr(dgamma), _ptr(dbeta)
    if pool is not None:
        flags = int(flags) | CONV_POOL
        if pool.dtype != torch.int64 or not pool.is_contiguous():
            raise _lib.VittaHipError("pool must be a contiguous int64 tensor (fixed-point sums)")
        d.pool, d.pool_scale = pool.data_ptr(), 1.0 / (geom.hy * geom.wy)
    d.C, d.K, d.flags, d.tile, d.ksplit = int(c), int(k), int(flags) | (_lib.CONV_PARITY4 if geom.cls_ntaps is not None else 0), int(tile), int(ksplit)
    geom.fill(d)
    if ksplit not in (1, -1):
        ws = workspace(x.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if KERNEL_COUNTS is not None:  # tests: which kernel family ran (vitta_conv_kernel)
        kid = int(lib().vitta_conv_kernel(C.byref(d)))
        KERNEL_COUNTS[kid] = KERNEL_COUNTS.get(kid, 0) + 1
    if TIMING is not None:  # bench.py: one event pair per launch, attached to the kernel's dispatch
        if KERNEL_TRACE is not None:
            KERNEL_TRACE.append(int(lib().vitta_conv_kernel(C.byref(d))))
        # algorithmic bytes of the launch (SURVEY 8d style: every operand once): x, the fp32 weights, y, and the epilogue's
        # input streams (residual / BatchNorm-backward input / mask); NOT the optional second output y_raw
        xp, yp = geom.n * geom.hs * geom.ws, geom.n * geom.hy * geom.wy
        rp = geom.n * ((geom.hy + 1) // 2) * ((geom.wy + 1) // 2) if (flags & CONV_RES_HALF) else yp
        abytes = 4 * (int(c) * xp + int(c) * int(k) * len(geom.taps) + int(k) * yp + (int(k) * rp if res is not None else 0)
                      + (int(k) * yp if bwd_x is not None else 0) + (int(k) * yp if bwd_mask is not None else 0))
        ev = TIMING(int(lib().vitta_conv_flops(C.byref(d))), (int(c), int(k), len(geom.taps), geom.n * geom.hg * geom.wg, abytes,
                                                             pool is not None,
                                                             bool(bwd_mask is not None and res is not None and (int(flags) & CONV_BWD_BN)),
                                                             4 * (int(c) * xp + int(c) * int(k) * len(geom.taps) + int(k) * yp)))
        check(lib().vitta_conv_timed_f32(C.byref(d), st, ev.start, ev.stop), "vitta_conv_timed_f32")
        return y
    check(lib().vitta_conv_f32(C.byref(d), st), "vitta_conv_f32")
    return y


def merged_dgrad_supported(geom, wp, c, k):
    """True when the parity-merged data gradient `geom` (Geometry.dgrad_merged) can run as one launch with this weight."""
    if geom is None or geom.cls_ntaps is None or ARITH != "b3" or os.environ.get("VITTA_CONV_B3_MERGED", "1") == "0":
        return False
    b3 = wp.b3 if isinstance(wp, Pack) else None
    if b3 is None:
        return False
    d = ConvDesc()
    d.x = d.y = d.w_b3 = b3.data_ptr()  # (only non-null pointers matter to the query)
    d.C, d.K, d.flags = int(c), int(k), _lib.CONV_PARITY4
    geom.fill(d)
    return bool(lib().vitta_conv_supported(C.byref(d)))


WGRAD_SLOT_BYTES = 25 << 20  # 768 workgroups x 2 partial tiles x 16 KiB, rounded up
WGRAD_SLOTS = 4
_wgrad_workspaces = {}


def wgrad_workspace(device):
    """Partial-tile areas of DEFERRED weight-gradient launches on the current stream (one slot per launch of a group)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _wgrad_workspaces.get(key)
    if ws is None:
        ws = _wgrad_workspaces[key] = torch.empty(WGRAD_SLOTS * WGRAD_SLOT_BYTES, dtype=torch.uint8, device=device)
    return ws


def wgrad_reduce(descs):
    """One launch adding the partial tiles of the deferred launches `descs` (wgrad(..., defer=slot)) to their gradients."""
    if not descs:
        return
    arr = (C.POINTER(_lib.WgradDesc) * len(descs))(*[C.pointer(d) for d in descs])
    check(lib().vitta_conv_wgrad_reduce_f32(arr, len(descs), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "vitta_conv_wgrad_reduce_f32")


def wgrad(geom, x, dy, grad_w, c, k, pro_bn=None, eps=1e-5, defer=None):
    """grad_w [K, C, kh, kw] += weight gradient of the convolution with FORWARD geometry `geom` (Geometry.forward):
    x [C, *] its input planes (raw, with pro_bn = the BatchNorm whose relu(bn(.)) the forward applied on load),
    dy [K, *] the gradient of its raw output (`vitta_conv_wgrad_f32`).  defer = slot 0..3: leave the partial tiles in that
    slot of the stream's wgrad workspace and return the descriptor for `wgrad_reduce` (one launch for a group)."""
    for t in (x, dy, grad_w):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.VittaHipError("convolution operands must be contiguous fp32 tensors on the GPU (no CPU fallback)")
    d = _lib.WgradDesc()
    d.x, d.dy, d.grad_w = x.data_ptr(), dy.data_ptr(), grad_w.data_ptr()
    _bn4(d.pro_bn, pro_bn)
    d.pro_eps = float(eps)
    d.flags = CONV_PRO_BN_RELU if pro_bn is not None else 0
    d.C, d.K, d.N = int(c), int(k), geom.n
    d.Hs, d.Ws, d.Hg, d.Wg, d.sstride = geom.hs, geom.ws, geom.hg, geom.wg, geom.sstride
    d.ntaps, d.wtaps = len(geom.taps), grad_w.shape[2] * grad_w.shape[3]
    for i, (dh, dw, wt) in enumerate(geom.taps):
        d.dh[i], d.dw[i], d.wt[i] = dh, dw, wt
    pointwise = len(geom.taps) == 1 and geom.sstride == 1 and geom.taps[0][:2] == (0, 0) and (geom.hg, geom.wg) == (geom.hs, geom.ws)
    if not pointwise:
        off, mask = geom.wgrad_tables(x.device)
        d.src_off, d.src_mask = off.data_ptr(), mask.data_ptr()
    if defer is not None:
        ws = wgrad_workspace(x.device)
        d.flags |= _lib.WGRAD_DEFER_REDUCE
        d.workspace, d.workspace_bytes = ws.data_ptr() + int(defer) * WGRAD_SLOT_BYTES, WGRAD_SLOT_BYTES
    else:
        ws = workspace(x.device)
        # the partial tiles use the slab region of the split-K workspace (its first 64 KiB are the convolutions' counters)
        d.workspace, d.workspace_bytes = ws.data_ptr() + 65536, ws.numel() - 65536
    check(lib().vitta_conv_wgrad_f32(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vitta_conv_wgrad_f32")
    return d if defer is not None else grad_w


def stem_wgrad(x, dy, grad_w):
    """grad_w [64, 3, 7, 7] += weight gradient of the stem convolution: x [N, 3, H, W] its input, dy [N, 64, OH, OW] the
    gradient of its raw output (`vitta_stem_conv7_wgrad_f32`)."""
    for t in (x, dy, grad_w):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.VittaHipError("convolution operands must be contiguous fp32 tensors on the GPU (no CPU fallback)")
    n, _, h, w = x.shape
    ws = workspace(x.device)
    need = int(lib().vitta_stem_conv7_wgrad_workspace_bytes())
    check(lib().vitta_stem_conv7_wgrad_f32(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), n, h, w, C.c_void_p(grad_w.data_ptr()),
                                           C.c_void_p(ws.data_ptr() + 65536), min(ws.numel() - 65536, max(need, 0) + (1 << 20)),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vitta_stem_conv7_wgrad_f32")
    return grad_w


# callable(flops, shape_key) -> ops.KernelEventPair, or None (the product never sets it)
TIMING = None
# dict {kernel family id (_lib.CONV_KERNEL_*): launches} filled while it is not None (tests assert the path under test)
KERNEL_COUNTS = None
# list of the kernel family of every TIMED launch, in order, while it is not None (bench.py: which peak a launch is held to)
KERNEL_TRACE = None


def pack_stem(w):
    """[64, 3, 7, 7] stem parameter -> [148][64]: row (c * 7 + kh) * 7 + kw = w[:, c, kh, kw], row 147 zero."""
    k, c, kh, kw = w.shape
    if (k, c, kh, kw) != (64, 3, 7, 7):
        raise _lib.VittaHipError("the stem kernel is Conv2d(3, 64, 7, stride 2, pad 3)")
    out = torch.zeros(148, 64, dtype=torch.float32, device=w.device)
    out[:147] = w.detach().permute(1, 2, 3, 0).reshape(147, 64)
    return out


def stem_conv(x, wp):
    """Raw output [N, 64, OH, OW] of the 7x7 / stride 2 / pad 3 stem convolution of x [N, 3, H, W] (vitta_stem_conv7_f32)."""
    for t in (x, wp):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.VittaHipError("convolution operands must be contiguous fp32 tensors on the GPU (no CPU fallback)")
    n, c, h, w = x.shape
    y = torch.empty(n, 64, (h - 1) // 2 + 1, (w - 1) // 2 + 1, dtype=torch.float32, device=x.device)
    check(lib().vitta_stem_conv7_f32(C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), n, h, w, C.c_void_p(y.data_ptr()),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vitta_stem_conv7_f32")
    return y


__all__ = ["Geometry", "launch", "Pack", "pack_b3", "make_pack", "b3_eligible", "wgrad", "stem_wgrad", "pack_stem", "stem_conv", "to_cm", "from_cm", "pack_fwd", "pack_bwd", "out_size", "CONV_PRO_BN_RELU", "CONV_EPI_APPLY",
           "CONV_EPI_RELU", "CONV_INJ_RAW", "CONV_POOL", "CONV_STATS", "CONV_STATS_RAW", "CONV_RES", "CONV_RES_HALF", "CONV_BWD_BN", "CONV_BWD_RELU"]
