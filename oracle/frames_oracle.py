"""CPU restatement of the reference's frame pre-processing arithmetic -- TEST INFRASTRUCTURE ONLY (imported by tests/
only, never by vitta_amd/).

What the reference runs (models/tanet_models/transforms.py:277-384 crop + `img.resize(size, Image.BILINEAR)`, :46-54 /
:170-184 scale + centre crop, :637-678 stack + /255, :140-152 normalise) bottoms out in a third-party dependency that is
not under /root/reference: **Pillow** (`requirements.txt` of the reference pins no version; 12.2.0 in this image), file
src/libImaging/Resample.c.  Its published 8-bit algorithm, restated here in numpy:

  precompute_coeffs:  scale = in/out; filterscale = max(scale, 1); support = 1 * filterscale (triangle filter);
      per output sample xx: center = (xx + .5) * scale; xmin = max(0, int(center - support + .5));
      xmax = min(in, int(center + support + .5)) - xmin; w_x = tri((x + xmin - center + .5) / filterscale), normalised
      by their running sum;
  normalize_coeffs_8bpc:  k = int(.5 + w * 2**22);
  ImagingResampleHorizontal_8bpc / Vertical_8bpc:  out = clip8((2**21 + sum_x pixel_x * k_x) >> 22); the horizontal
      pass writes a byte image which the vertical pass resamples.

PINNED: tests/test_frames_cpu.py checks this restatement against Pillow itself (`Image.crop(...).resize(..., BILINEAR)`)
on up- and down-scaling crops, and against tests/golden/data_pipeline.npz (outputs of the reference's own transform
classes).
"""
import numpy as np

BITS = 22


def _taps(n_in, n_out):
    scale = n_in / n_out
    fs = max(scale, 1.0)
    support = fs
    out = []
    for xx in range(n_out):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), n_in)
        pos = (np.arange(lo, hi, dtype=np.float64) - center + 0.5) * (1.0 / fs)
        w = np.where(np.abs(pos) < 1.0, 1.0 - np.abs(pos), 0.0)
        total = np.cumsum(w)[-1] if len(w) else 0.0  # running (left to right) sum, as the C loop
        if total != 0.0:
            w = w / total
        out.append((lo, (0.5 + w * float(1 << BITS)).astype(np.int64)))  # weights are non-negative: int() == floor
    return out


def _pass(img, taps, axis):
    """Resample `img` (uint8, [H, W, C]) along `axis` (0 rows / 1 columns)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    dst = np.empty((len(taps),) + src.shape[1:], dtype=np.uint8)
    for i, (lo, k) in enumerate(taps):
        acc = (1 << (BITS - 1)) + np.tensordot(k, src[lo:lo + len(k)], axes=(0, 0))
        dst[i] = np.clip(acc >> BITS, 0, 255)
    return np.moveaxis(dst, 0, axis)


def resize_bilinear(img, size):
    """Pillow `Image.resize(size, BILINEAR)` of an RGB byte image [H, W, 3]; size = (W, H)."""
    w, h = size
    out = img
    if w != img.shape[1]:
        out = _pass(out, _taps(img.shape[1], w), 1)
    if h != img.shape[0]:
        out = _pass(out, _taps(img.shape[0], h), 0)
    return out


def clip_input(frames, boxes, frames_per_view, size, mean, std, resize=None, window=(0, 0)):
    """frames uint8 [F, H, W, 3]; view v crops boxes[v] = (x0, y0, w, h), resizes to `resize` (default `size`), keeps the
    `window` of `size`; frames stacked on the channel axis, /255, (x - mean_c) / std_c in float32 -> [F*3, H, W]."""
    resize = resize or size
    planes = []
    for f, frame in enumerate(frames):
        x0, y0, w, h = boxes[f // frames_per_view]
        img = resize_bilinear(frame[y0:y0 + h, x0:x0 + w], resize)
        img = img[window[1]:window[1] + size[1], window[0]:window[0] + size[0]]
        for c in range(3):
            v = img[:, :, c].astype(np.float32) / np.float32(255)
            planes.append((v - np.float32(mean[c])) / np.float32(std[c]))
    return np.stack(planes).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------------
# Video Swin pipeline: mmcv.imresize = cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) on uint8 frames
# (models/videoswintransformer_models/transforms_backup.py:193-349, video_dataset.py:60-101).  Third-party dependency absent
# from /root/reference AND from this image: **OpenCV** (pulled in through mmcv; no version pinned by the reference).
# PARITY UNPINNED: there is no cv2 here to check against and the reference's tests hold no resize fixture; the
# restatement follows the published algorithm of modules/imgproc/src/resize.cpp (4.x):
#   scale = 1 / (dst / src) (double);  per destination sample d: f = float((d + 0.5) * scale - 0.5); s = floor(f); f -= s
#   columns: s < 0 -> (s, f) = (0, 0); s >= src - 1 -> (src - 1, 0);   rows: indices s, s + 1 clipped to [0, src - 1]
#   weights: short(round_half_even((1 - f) * 2048)), short(round_half_even(f * 2048))     (INTER_RESIZE_COEF_BITS = 11)
#   horizontal: H = S[s] * a0 + S[s + 1] * a1   (int32);  vertical: ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2 >> 2
#   same size: copy;  exactly 2x down in both axes: INTER_AREA's (a + b + c + d + 2) >> 2.
# Written with scalar loops on purpose (the product's numpy form in vitta_amd/frames.py is vectorised differently).
# ------------------------------------------------------------------------------------------------------------------------
def _cv2_axis(src, dst, vertical):
    import math
    out = []
    scale = 1.0 / (float(dst) / float(src))
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        if vertical:
            i0, i1 = min(max(s, 0), src - 1), min(max(s + 1, 0), src - 1)
        else:
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= src - 1:
                s, f = src - 1, np.float32(0)
            i0, i1 = s, min(s + 1, src - 1)
        w0 = int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048)))
        w1 = int(np.rint(f * np.float32(2048)))
        out.append((i0, i1, w0, w1))
    return out


def cv2_resize_linear(img, dw, dh):
    """uint8 [H, W, C] -> [dh, dw, C], the (unpinned) restatement of cv2.resize(..., INTER_LINEAR)."""
    h, w, c = img.shape
    if (h, w) == (dh, dw):
        return img.copy()
    out = np.zeros((dh, dw, c), dtype=np.uint8)
    src = img.astype(np.int64)
    if h == 2 * dh and w == 2 * dw:
        for y in range(dh):
            for x in range(dw):
                out[y, x] = (src[2 * y, 2 * x] + src[2 * y, 2 * x + 1] + src[2 * y + 1, 2 * x] + src[2 * y + 1, 2 * x + 1] + 2) >> 2
        return out
    xs, ys = _cv2_axis(w, dw, False), _cv2_axis(h, dh, True)
    hbuf = np.zeros((h, dw, c), dtype=np.int64)
    for x, (i0, i1, a0, a1) in enumerate(xs):
        hbuf[:, x] = src[:, i0] * a0 + src[:, i1] * a1
    for y, (j0, j1, b0, b1) in enumerate(ys):
        v = (((b0 * (hbuf[j0] >> 4)) >> 16) + ((b1 * (hbuf[j1] >> 4)) >> 16) + 2) >> 2
        out[y] = np.clip(v, 0, 255)
    return out


def swin_clip(frames, views, clip_len, scale_size, input_size, box, mean, std):
    """frames uint8 [F, H, W, 3] -> float32 [views, 3, clip_len, S, S] as the reference's Video Swin test pipelines:
    Resize((-1, scale_size)) -> crop `box(nh, nw)` (RandomResizedCrop's box, or None: CenterCrop) -> Resize to S x S (TTA)
    -> Normalize -> FormatShape('NCTHW')."""
    f, h, w, _ = frames.shape
    factor = scale_size / min(h, w)
    nh, nw = int(h * float(factor) + 0.5), int(w * float(factor) + 0.5)
    m = np.asarray(mean, dtype=np.float64).astype(np.float32)
    sinv = (1.0 / np.asarray(std, dtype=np.float64)).astype(np.float32)
    out = np.zeros((views, 3, clip_len, input_size, input_size), dtype=np.float32)
    for i in range(f):
        x = cv2_resize_linear(frames[i], nw, nh)
        if box is None:
            l, t = (nw - input_size) // 2, (nh - input_size) // 2
            r, b = l + input_size, t + input_size
        else:
            l, t, r, b = box(nh, nw)
        x = cv2_resize_linear(np.ascontiguousarray(x[t:b, l:r]), input_size, input_size)
        y = (x.astype(np.float32) - m) * sinv
        out[i // clip_len, :, i % clip_len] = y.transpose(2, 0, 1)
    return out
