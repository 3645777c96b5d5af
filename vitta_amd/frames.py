"""Device-side crop + resize + normalise of decoded frames, bit-identical to the reference's
PIL pipeline (models/tanet_models/transforms.py:277-384, :46-54, :170-184, :637-678, :140-152).

The host builds, per view, what depends only on (crop size, output size): the tap windows and 22-bit fixed-point
weights of Pillow's 8-bit BILINEAR resampler (Pillow's Resample.c `precompute_coeffs` + `normalize_coeffs_8bpc`, in
double precision, same operation order) and the 3 x 256 table `(byte / 255 - mean_c) / std_c` (with the reference's own
float32 ops, so every entry is the value the host pipeline would produce).  `vitta_frames_resample_norm_f32` does the two
integer passes, the byte rounding between them and the table look-up for all frames of all views in ONE launch, reading
the uploaded uint8 frames (a quarter of the PCIe bytes of a float clip) and writing the [V*T*3, H, W] input directly.

No CPU fallback: `resample_normalise` raises VittaHipError for host tensors (the host pipeline is vitta_amd.data_video).
"""
import ctypes as C
import functools
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

PRECISION_BITS = 32 - 8 - 2
TILE_ROWS = 8
FAST_TAPS = 4
LDS_BYTES = 64 * 1024


@functools.lru_cache(maxsize=256)
def bilinear_taps(in_size, out_size):
    """Pillow's tap windows for resizing `in_size` samples to `out_size` with the BILINEAR (triangle) filter:
    bounds int32 [out_size, 2] = (first input sample, tap count), coefs int32 [out_size, ksize] fixed point."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws, ww = [], 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            ws.append(w)
            ww += w
        for x in range(xmax):
            w = ws[x] / ww if ww != 0.0 else ws[x]
            coefs[xx, x] = int(-0.5 + w * one) if w < 0 else int(0.5 + w * one)
        bounds[xx] = (xmin, xmax)
    bounds.setflags(write=False)  # cached: shared between plans
    coefs.setflags(write=False)
    return bounds, coefs


def normalise_table(mean, std):
    """[3, 256] float32: ToTorchFormatTensor(div=True) then GroupNormalize on every possible byte."""
    v = torch.arange(256, dtype=torch.uint8).float().div(255)
    m = torch.tensor(list(mean), dtype=torch.float32).view(-1, 1)
    s = torch.tensor(list(std), dtype=torch.float32).view(-1, 1)
    return v.unsqueeze(0).repeat(len(mean), 1).sub_(m).div_(s)


class ViewSpec:
    """One view: crop `box` = (x0, y0, w, h) of the frame, resized to `resize` = (W, H), of which the `window`
    (left, top) + the plan's output size is kept.  tta view: window (0, 0) and resize == output size; eval: box = whole
    frame, resize = short edge to scale_size, window = the centre crop."""

    def __init__(self, box, resize, window=(0, 0)):
        self.box, self.resize, self.window = tuple(box), tuple(resize), tuple(window)


class FramePlan:
    """Tap tables of a list of views for one launch (host-built, uploaded once per clip: a few KB)."""

    def __init__(self, views, out_size, device, mean, std, lut=None):
        out_w, out_h = out_size
        xs, ys = [], []
        for v in views:
            x0, y0, w, h = v.box
            rw, rh = v.resize
            left, top = v.window
            if not (0 <= left and left + out_w <= rw and 0 <= top and top + out_h <= rh):
                raise ValueError(f"output window {v.window}+{out_size} outside the resized view {v.resize}")
            bx, cx = bilinear_taps(w, rw)
            by, cy = bilinear_taps(h, rh)
            xs.append((bx[left:left + out_w], cx[left:left + out_w]))
            ys.append((by[top:top + out_h], cy[top:top + out_h]))
        # row stride of the weight tables = the longest window actually used (Pillow's ksize is an upper bound); windows
        # of up to FAST_TAPS taps take the kernel's unrolled path, which wants exactly that stride (zero padded)
        longest = lambda tabs: max(int(b[:, 1].max()) for b, _ in tabs)
        stride = lambda n: FAST_TAPS if n <= FAST_TAPS else n
        self.kx, self.ky = stride(longest(xs)), stride(longest(ys))
        pad = lambda c, k: np.pad(c, ((0, 0), (0, max(k - c.shape[1], 0))))[:, :k]
        tile, rows = TILE_ROWS, None
        while True:  # the input rows one tile of output rows spans must fit the workgroup's LDS
            rows = max(int((b[t:t + tile, 0] + b[t:t + tile, 1]).max() - b[t, 0]) for b, _ in ys for t in range(0, out_h, tile))
            if rows * 4 * out_w + 3072 <= LDS_BYTES or tile == 1:  # one dword per (row, column) + the byte table
                break
            tile //= 2
        if rows * 4 * out_w + 3072 > LDS_BYTES:
            raise _lib.VittaHipError(f"a single output row spans {rows} input rows of {out_w} columns: beyond the kernel's LDS tile")
        self.tile_rows, self.lds_rows = tile, rows
        host = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=True)
        self.origin = host(np.array([[v.box[0], v.box[1]] for v in views], dtype=np.int32))
        self.xb, self.xc = host(np.stack([b for b, _ in xs])), host(np.stack([pad(c, self.kx) for _, c in xs]))
        self.yb, self.yc = host(np.stack([b for b, _ in ys])), host(np.stack([pad(c, self.ky) for _, c in ys]))
        self.lut = lut if lut is not None else normalise_table(mean, std).to(device)
        self.out_w, self.out_h, self.n_views = out_w, out_h, len(views)
        self.boxes = [v.box for v in views]


def resample_normalise(frames, plan, frames_per_view, out=None):
    """frames: uint8 [F, H, W, 3] on the GPU -> float32 [F*3, out_h, out_w] (the reference's stacked clip layout)."""
    if not frames.is_cuda:
        raise _lib.VittaHipError(f"frames must live on the GPU (got {frames.device}); the HIP path has no CPU fallback")
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise _lib.VittaHipError(f"frames must be uint8 [F, H, W, 3] (got {frames.dtype} {tuple(frames.shape)})")
    frames = frames.contiguous()
    f, h, w, _ = frames.shape
    if f != plan.n_views * frames_per_view:
        raise ValueError(f"{f} frames for {plan.n_views} views of {frames_per_view}")
    for x0, y0, bw, bh in plan.boxes:
        if not (0 <= x0 and x0 + bw <= w and 0 <= y0 and y0 + bh <= h):
            raise ValueError(f"crop box {(x0, y0, bw, bh)} outside the {w}x{h} frame")
    if out is None:
        out = torch.empty(f * 3, plan.out_h, plan.out_w, dtype=torch.float32, device=frames.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib().vitta_frames_resample_norm_f32(p(frames), f, h, w, frames_per_view, p(plan.origin), p(plan.xb), p(plan.xc),
                                               plan.kx, p(plan.yb), p(plan.yc), plan.ky, p(plan.lut), p(out), plan.out_h,
                                               plan.out_w, plan.tile_rows, plan.lds_rows,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "vitta_frames_resample_norm_f32")
    return out


def eval_view(frame_size, scale_size, input_size):
    """GroupScale(scale_size) + GroupCenterCrop(input_size) as one ViewSpec (torchvision 0.8.2 Resize / CenterCrop)."""
    w, h = frame_size
    if (w <= h and w == scale_size) or (h <= w and h == scale_size):
        rw, rh = w, h
    elif w < h:
        rw, rh = scale_size, int(scale_size * h / w)
    else:
        rw, rh = int(scale_size * w / h), scale_size
    top, left = int(round((rh - input_size) / 2.0)), int(round((rw - input_size) / 2.0))
    return ViewSpec((0, 0, w, h), (rw, rh), (left, top))


# ------------------------------------------------------------------------------------------------------------------------
# Video Swin pipeline (models/videoswintransformer_models/video_dataset.py:60-101 over transforms_backup.py Resize :193-349,
# RandomResizedCrop / CenterCrop, Normalize, FormatShape NCTHW): every resize is mmcv.imresize = cv2.resize(...,
# INTER_LINEAR) on uint8 frames.  cv2 is not installed in this image: the arithmetic below restates OpenCV's published 8-bit
# bilinear resampler (modules/imgproc/src/resize.cpp: 11-bit fixed-point weights, integer horizontal pass, the
# (b0 (S0 >> 4) >> 16) + (b1 (S1 >> 4) >> 16) + 2 >> 2 vertical pass, the exact-2x area shortcut) and is NOT pinned against
# cv2 itself.  Host path: numpy; device path: vitta_frames_cv2_resize_* (bit-identical to the host path).
# ------------------------------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=256)
def cv2_linear_axis(src, dst, vertical):
    """(first index, second index, w0, w1) int32 [dst] of cv2.resize(INTER_LINEAR) along one axis of `src` -> `dst` samples."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if vertical:  # rows are clipped, the weights keep the unclamped fraction
        i0, i1 = np.clip(s, 0, src - 1), np.clip(s + 1, 0, src - 1)
    else:         # columns: fraction forced to 0 at both borders
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, src - 1, s))
        i0, i1 = s, np.minimum(s + 1, src - 1)
    w1 = np.rint(f * np.float32(2048)).astype(np.int32)
    w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int32)
    return i0.astype(np.int32), i1.astype(np.int32), w0, w1


def cv2_resize_linear(img, dw, dh):
    """uint8 [..., H, W, C] -> [..., dh, dw, C] as cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR) (host path)."""
    h, w = img.shape[-3], img.shape[-2]
    if (h, w) == (dh, dw):
        return img.copy()
    if h == 2 * dh and w == 2 * dw:  # INTER_LINEAR at exactly 2x down takes cv2's INTER_AREA fast path
        v = img.astype(np.int32)
        return ((v[..., 0::2, 0::2, :] + v[..., 0::2, 1::2, :] + v[..., 1::2, 0::2, :] + v[..., 1::2, 1::2, :] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = cv2_linear_axis(w, dw, False)
    y0, y1, b0, b1 = cv2_linear_axis(h, dh, True)
    v = img.astype(np.int32)
    hb = v[..., :, x0, :] * a0[:, None] + v[..., :, x1, :] * a1[:, None]          # [..., H, dw, C]
    s0, s1 = hb[..., y0, :, :], hb[..., y1, :, :]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def swin_norm_constants(mean, std):
    """mmcv.imnormalize: float32 (x - mean) * (1 / std), mean and 1 / std rounded to float32."""
    m = np.asarray(mean, dtype=np.float64).astype(np.float32)
    s = (1.0 / np.asarray(std, dtype=np.float64)).astype(np.float32)
    return m, s


def swin_clip_host(frames, views, clip_len, scale_size, input_size, crop_box, mean, std):
    """frames uint8 [F, H, W, 3] (F = views * clip_len) -> float32 [views, 3, clip_len, S, S]: short edge -> scale_size,
    crop_box(nh, nw) -> (l, t, r, b) (None: centre crop of input_size), resize to S x S, normalise, NCTHW."""
    f, h, w, _ = frames.shape
    nh, nw = swin_scaled_size(h, w, scale_size)
    x = cv2_resize_linear(frames, nw, nh)
    l, t, r, b = crop_box(nh, nw) if crop_box is not None else centre_box(nh, nw, input_size)
    x = cv2_resize_linear(np.ascontiguousarray(x[:, t:b, l:r]), input_size, input_size)
    m, s = swin_norm_constants(mean, std)
    y = (x.astype(np.float32) - m) * s
    return torch.from_numpy(np.ascontiguousarray(y.reshape(views, clip_len, input_size, input_size, 3).transpose(0, 4, 1, 2, 3)))


def swin_scaled_size(h, w, short):
    """mmcv.rescale_size((w, h), (inf, short)): the short edge becomes `short`, the other int(x * factor + 0.5)."""
    factor = short / min(h, w)
    return int(h * float(factor) + 0.5), int(w * float(factor) + 0.5)


def centre_box(nh, nw, s):
    l, t = (nw - s) // 2, (nh - s) // 2
    return l, t, l + s, t + s


class Cv2ResizePlan:
    """Device tables of one cv2-style resize (source size, crop box inside it, destination size)."""

    def __init__(self, src_hw, box, dst_hw, device):
        (sh, sw), (l, t, r, b), (dh, dw) = src_hw, box, dst_hw
        ch, cw = b - t, r - l
        self.src_hw, self.box, self.dst_hw = (sh, sw), (l, t, r, b), (dh, dw)
        self.mode = 0 if (ch, cw) == (dh, dw) else (1 if (ch == 2 * dh and cw == 2 * dw) else 2)
        x0, x1, a0, a1 = cv2_linear_axis(cw, dw, False) if self.mode == 2 else (np.zeros(dw, np.int32),) * 4
        y0, y1, b0, b1 = cv2_linear_axis(ch, dh, True) if self.mode == 2 else (np.zeros(dh, np.int32),) * 4
        tab = np.concatenate([x0 + l, x1 + l, a0, a1, y0 + t, y1 + t, b0, b1]).astype(np.int32)
        self.table = torch.from_numpy(tab).to(device)


def cv2_resize_device(frames, plan, out_u8=None, out_f32=None, clip_len=0, mean=None, stdinv=None):
    """frames uint8 [F, sh, sw, 3] on the GPU -> uint8 [F, dh, dw, 3] (out_u8) or normalised float32 [V, 3, T, dh, dw]
    (out_f32, V = F / clip_len) through `vitta_frames_cv2_resize`."""
    if not frames.is_cuda or frames.dtype != torch.uint8 or not frames.is_contiguous():
        raise _lib.VittaHipError("frames must be a contiguous uint8 tensor on the GPU (no CPU fallback)")
    f, sh, sw, _ = frames.shape
    dh, dw = plan.dst_hw
    l, t, _, _ = plan.box
    ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p(0)
    check(lib().vitta_frames_cv2_resize(ptr(frames), f, sh, sw, l, t, plan.mode, ptr(plan.table), dh, dw, ptr(out_u8), ptr(out_f32),
                                        int(clip_len), ptr(mean), ptr(stdinv), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "vitta_frames_cv2_resize")
    return out_u8 if out_u8 is not None else out_f32


def swin_clip_on_device(frames_u8, device, views, clip_len, scale_size, input_size, crop_box, mean, std):
    """The same pipeline as swin_clip_host in two launches on the uploaded uint8 frames (bit-identical to it)."""
    fr = torch.from_numpy(np.ascontiguousarray(frames_u8)).to(device) if not torch.is_tensor(frames_u8) else frames_u8.to(device)
    f, h, w, _ = fr.shape
    nh, nw = swin_scaled_size(h, w, scale_size)
    mid = torch.empty(f, nh, nw, 3, dtype=torch.uint8, device=device)
    cv2_resize_device(fr, Cv2ResizePlan((h, w), (0, 0, w, h), (nh, nw), device), out_u8=mid)
    box = crop_box(nh, nw) if crop_box is not None else centre_box(nh, nw, input_size)
    m, s = swin_norm_constants(mean, std)
    out = torch.empty(views, 3, clip_len, input_size, input_size, dtype=torch.float32, device=device)
    cv2_resize_device(mid, Cv2ResizePlan((nh, nw), box, (input_size, input_size), device), out_f32=out, clip_len=clip_len,
                      mean=torch.from_numpy(m).to(device), stdinv=torch.from_numpy(s).to(device))
    return out
