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
