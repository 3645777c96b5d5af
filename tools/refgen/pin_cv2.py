"""Pin the restated OpenCV resize against OpenCV itself -- ONE command on any box that has cv2 (this build image does not):

    python tools/refgen/pin_cv2.py            # writes tests/golden/cv2_resize.npz

The Video Swin input pipeline of the reference (models/videoswintransformer_models/transforms_backup.py:193-349, 1120-1286:
Resize / RandomResizedCrop / CenterCrop through mmcv.imresize) is cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) on
uint8 frames.  vitta_amd/frames.py::cv2_resize_linear, oracle/frames_oracle.py::cv2_resize_linear and the kernel
vitta_frames_cv2_resize restate that arithmetic and are bit-identical to EACH OTHER (tests/test_frames_cpu.py,
tests/test_gpu_frames.py); this script records what cv2 itself returns for the same seeded images and sizes, and
tests/test_frames_cpu.py::test_cv2_restatement_against_opencv_itself compares all restatements with it, bit for bit, whenever
the fixture exists (skipped with the reason "no cv2 fixture" otherwise).  Inputs are regenerated from the seeds below: the
fixture holds cv2's OUTPUTS only (data, no source text)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden", "cv2_resize.npz")

# (h, w, dst_h, dst_w): up- and down-scaling, the same-size copy, the exact-2x case (cv2's INTER_AREA shortcut), extreme aspect
# changes, the shapes of the shipped pipeline (320x240 -> short edge 256 -> crop -> 224^2) and the widest frame the kernel takes
CASES = [(240, 320, 256, 341), (256, 341, 224, 224), (97, 131, 224, 224), (60, 80, 30, 40), (50, 70, 50, 70), (33, 47, 11, 200),
         (120, 90, 7, 5), (224, 224, 224, 224), (480, 640, 224, 298), (1080, 1920, 256, 455), (131, 97, 262, 194), (17, 23, 224, 224)]


def image(case_index, h, w):
    """The seeded uint8 test image of a case (full byte range, plus a smooth gradient in channel 2)."""
    rng = np.random.RandomState(1000 + case_index)
    img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    img[..., 2] = ((np.arange(h)[:, None] * 3 + np.arange(w)[None, :] * 5) % 256).astype(np.uint8)
    return img


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: run this script on a box with opencv-python (any 4.x); nothing written", file=sys.stderr)
        return 2
    out = {"cv2_version": np.array(cv2.__version__), "cases": np.array(CASES, dtype=np.int32)}
    for i, (h, w, dh, dw) in enumerate(CASES):
        out[f"out{i}"] = cv2.resize(image(i, h, w), (dw, dh), interpolation=cv2.INTER_LINEAR)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.1f} KiB) with cv2 {cv2.__version__}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
