"""N1 device pre-processing, CPU side: the oracle's restatement of Pillow's 8-bit BILINEAR resampler pinned against
Pillow itself and against the golden of the reference's own transform classes; the product's host-built tap tables and
normalisation table against the oracle."""
import random

import numpy as np
import pytest
import torch

import helpers as H
from oracle import frames_oracle as FO
from vitta_amd import data_video as DV
from vitta_amd import frames as FR

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def pil_frames(n, w, h, seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        base = rng.randint(0, 256, size=(h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
        out.append(Image.fromarray(base).resize((w, h), Image.BICUBIC))
    return out


@pytest.mark.parametrize("in_wh,box,out_wh", [
    ((340, 256), (58, 16, 224, 224), (224, 224)),   # same size: identity
    ((340, 256), (0, 0, 256, 256), (224, 224)),     # mild down-scaling: 3-tap windows widen
    ((340, 256), (29, 8, 168, 192), (224, 224)),    # up-scaling, distorted aspect
    ((320, 240), (0, 0, 320, 240), (341, 256)),     # eval: short edge 240 -> 256
    ((720, 480), (10, 20, 700, 450), (112, 112)),   # strong down-scaling: 15-tap windows
    ((97, 61), (3, 5, 90, 50), (31, 77)),           # odd sizes, one axis down, the other up
])
def test_oracle_resampler_is_pillow(in_wh, box, out_wh):
    from PIL import Image
    img = pil_frames(1, *in_wh, seed=4)[0]
    x0, y0, w, h = box
    ref = np.asarray(img.crop((x0, y0, x0 + w, y0 + h)).resize(out_wh, Image.BILINEAR))
    got = FO.resize_bilinear(np.asarray(img)[y0:y0 + h, x0:x0 + w], out_wh)
    np.testing.assert_array_equal(got, ref)
    noise = np.random.RandomState(1).randint(0, 256, size=(in_wh[1], in_wh[0], 3)).astype(np.uint8)  # full-range bytes
    ref = np.asarray(Image.fromarray(noise).crop((x0, y0, x0 + w, y0 + h)).resize(out_wh, Image.BILINEAR))
    np.testing.assert_array_equal(FO.resize_bilinear(noise[y0:y0 + h, x0:x0 + w], out_wh), ref)


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_clip_matches_reference_transforms(case):
    """The whole chain (per-view crop, resize, stack, /255, normalise) against the golden of the reference's own
    SubgroupWise_MultiScaleCrop / Stack / ToTorchFormatTensor / GroupNormalize and, bit for bit, the host pipeline."""
    g = H.golden("data_pipeline.npz")
    w, h, views, T, size = (int(v) for v in g[f"tanet_{case}_cfg"])
    frames = pil_frames(views * T, w, h, 11)
    random.seed(5)
    boxes = []
    for _ in range(views):
        cw, ch, ow, oh = DV.sample_multiscale_crop((w, h), (size, size))
        boxes.append((ow, oh, cw, ch))
    arr = np.stack([np.asarray(f) for f in frames])
    got = FO.clip_input(arr, boxes, T, (size, size), MEAN, STD)
    assert list(got.shape) == g[f"tanet_{case}_shape"].tolist()
    np.testing.assert_allclose(got[:, ::16, ::16], g[f"tanet_{case}_sub"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got.astype(np.float64).sum((1, 2)), g[f"tanet_{case}_chsum"], rtol=1e-9, atol=1e-3)
    random.seed(5)
    host = DV.stack_to_tensor(DV.subgroup_multiscale_crop(frames, views, T, size), MEAN, STD)
    assert torch.equal(torch.from_numpy(got), host)


@pytest.mark.parametrize("n_in,n_out", [(224, 224), (256, 224), (168, 224), (240, 256), (700, 112), (1080, 224), (7, 3), (3, 7)])
def test_product_tap_tables_match_oracle(n_in, n_out):
    bounds, coefs = FR.bilinear_taps(n_in, n_out)
    taps = FO._taps(n_in, n_out)
    assert bounds.shape == (n_out, 2) and coefs.dtype == np.int32
    for i, (lo, k) in enumerate(taps):
        assert bounds[i, 0] == lo and bounds[i, 1] == len(k)
        np.testing.assert_array_equal(coefs[i, :len(k)], k)
        assert not coefs[i, len(k):].any()


def test_normalise_table_is_the_host_pipeline_on_every_byte():
    from PIL import Image
    lut = FR.normalise_table(MEAN, STD)
    img = Image.fromarray(np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2))
    host = DV.stack_to_tensor([img], MEAN, STD)  # [3, 16, 16]
    assert torch.equal(lut, host.reshape(3, 256))


def test_eval_view_is_scale_then_centre_crop():
    for (w, h) in [(320, 240), (340, 256), (240, 320), (256, 256)]:
        img = pil_frames(1, w, h, 9)[0]
        ref = np.asarray(DV.center_crop(DV.scale_short_edge(img, 256), 224))
        v = FR.eval_view((w, h), 256, 224)
        got = FO.resize_bilinear(np.asarray(img), v.resize)[v.window[1]:v.window[1] + 224, v.window[0]:v.window[0] + 224]
        np.testing.assert_array_equal(got, ref)


def test_frame_plan_tiles_fit_lds_and_refuse_host_tensors():
    views = [FR.ViewSpec((0, 0, 1920, 1080), (398, 224)), FR.ViewSpec((100, 50, 168, 192), (398, 224), (0, 0))]
    plan = FR.FramePlan(views, (224, 224), "cpu", MEAN, STD)
    assert plan.lds_rows * 3 * 224 <= FR.LDS_BYTES and plan.tile_rows >= 1
    assert plan.xc.shape[0] == 2 and plan.kx == int(FR.bilinear_taps(1920, 398)[0][:, 1].max()) and plan.xc.shape[2] == plan.kx
    small = FR.FramePlan([FR.ViewSpec((0, 0, 256, 240), (224, 256))], (224, 224), "cpu", MEAN, STD)
    assert small.kx == 4 and small.ky == 4 and small.xc.shape == (1, 224, 4)  # 3-tap windows padded to the unrolled path
    with pytest.raises(Exception, match="GPU"):
        FR.resample_normalise(torch.zeros(2, 1080, 1920, 3, dtype=torch.uint8), plan, 1)
    with pytest.raises(ValueError):
        FR.FramePlan([FR.ViewSpec((0, 0, 100, 100), (200, 200), (10, 10))], (224, 224), "cpu", MEAN, STD)


# ---- Video Swin pipeline: cv2.resize(INTER_LINEAR) restated (UNPINNED: no cv2 in this image) ---------------------------------
@pytest.mark.parametrize("h,w,dh,dw", [(240, 320, 256, 341), (256, 341, 224, 224), (97, 131, 224, 224), (60, 80, 30, 40),
                                       (50, 70, 50, 70), (33, 47, 11, 200), (120, 90, 7, 5)])
def test_cv2_linear_host_path_equals_the_oracle_restatement(h, w, dh, dw):
    """The product's vectorised host resize (vitta_amd/frames.py::cv2_resize_linear) against the scalar-loop restatement of
    OpenCV's 8-bit INTER_LINEAR in oracle/frames_oracle.py, bit for bit (up- and down-scaling, the same-size copy, the exact-2x
    area shortcut, full-range bytes).  Neither is pinned against cv2 itself -- DESIGN.md section 5."""
    from oracle import frames_oracle as FO
    from vitta_amd import frames as F
    rng = np.random.RandomState(h * 7 + w + dh + dw)
    img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    np.testing.assert_array_equal(F.cv2_resize_linear(img, dw, dh), FO.cv2_resize_linear(img, dw, dh))


def test_cv2_restatement_against_opencv_itself():
    """Closes SURVEY N1 for Video Swin on any box that has cv2: `python tools/refgen/pin_cv2.py` there writes
    tests/golden/cv2_resize.npz (cv2's own outputs for seeded images); host path and oracle restatement must match it bit for
    bit.  This image has no cv2 and ships no fixture: skipped, and DESIGN.md section 5 says "parity with cv2 unpinned"."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cv2_resize.npz")
    if not os.path.exists(path):
        pytest.skip("no cv2 fixture (tests/golden/cv2_resize.npz): run tools/refgen/pin_cv2.py on a box with opencv-python")
    spec = importlib.util.spec_from_file_location("pin_cv2", os.path.join(os.path.dirname(os.path.dirname(path)), "..", "tools", "refgen", "pin_cv2.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    from oracle import frames_oracle as FO
    from vitta_amd import frames as F
    g = np.load(path, allow_pickle=True)
    assert g["cases"].tolist() == [list(c) for c in pin.CASES]
    for i, (h, w, dh, dw) in enumerate(pin.CASES):
        img = pin.image(i, h, w)
        np.testing.assert_array_equal(F.cv2_resize_linear(img, dw, dh), g[f"out{i}"], err_msg=f"host path, case {i}, cv2 {g['cv2_version']}")
        if h * w <= 100000:  # the scalar-loop oracle on the small cases
            np.testing.assert_array_equal(FO.cv2_resize_linear(img, dw, dh), g[f"out{i}"], err_msg=f"oracle, case {i}")


def test_cv2_linear_restatement_properties():
    """What any correct bilinear byte resampler satisfies: constants stay constant, a resize is within one byte (plus the
    fixed-point rounding) of the float bilinear interpolation with half-pixel centres, weights sum to 2048 away from the rows'
    borders."""
    import torch
    from vitta_amd import frames as F
    c = np.full((40, 60, 3), 201, np.uint8)
    assert np.unique(F.cv2_resize_linear(c, 224, 224)).tolist() == [201]
    rng = np.random.RandomState(3)
    base = rng.randint(0, 256, size=(12, 16, 3)).astype(np.uint8)
    img = np.repeat(np.repeat(base, 8, axis=0), 8, axis=1)  # piecewise constant: no aliasing in the comparison
    got = F.cv2_resize_linear(img, 150, 120).astype(np.float32)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(120, 150), mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1.5
    i0, i1, w0, w1 = F.cv2_linear_axis(97, 224, False)
    assert ((w0 + w1) == 2048).all() and (i1 - i0 <= 1).all() and i0.min() == 0 and i1.max() == 96


def test_swin_clip_host_pipeline_equals_oracle():
    from oracle import frames_oracle as FO
    from vitta_amd import frames as F
    rng = np.random.RandomState(5)
    frames = rng.randint(0, 256, size=(4, 48, 64, 3)).astype(np.uint8)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    box = lambda nh, nw: (5, 3, 5 + 40, 3 + 37)
    for bx in (None, box):
        got = F.swin_clip_host(frames, 2, 2, 56, 32, bx, mean, std).numpy()
        ref = FO.swin_clip(frames, 2, 2, 56, 32, bx, mean, std)
        np.testing.assert_array_equal(got, ref)
