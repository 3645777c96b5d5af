"""N1 on the GPU: `vitta_frames_resample_norm_f32` (crop + Pillow-BILINEAR resize + stack + /255 + normalise in one
launch) against the oracle's restatement of Pillow, bit for bit, and against the golden of the reference's own
transform classes."""
import random

import numpy as np
import pytest
import torch

import helpers as H
from oracle import frames_oracle as FO
from vitta_amd import data_video as DV
from vitta_amd import frames as FR

pytestmark = pytest.mark.gpu
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
DEV = "cuda:0"


def byte_frames(n, w, h, seed, smooth=True):
    rng = np.random.RandomState(seed)
    if not smooth:
        return rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    base = rng.randint(0, 256, size=(n, h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    return np.stack([FO.resize_bilinear(b, (w, h)) for b in base])


def run(frames, views, out_size, fpv):
    plan = FR.FramePlan(views, out_size, DEV, MEAN, STD)
    out = FR.resample_normalise(torch.from_numpy(frames).to(DEV), plan, fpv)
    torch.cuda.synchronize()
    return out.cpu().numpy(), plan


@pytest.mark.parametrize("case", ["a", "b"])
def test_tta_views_match_reference_transforms(case):
    g = H.golden("data_pipeline.npz")
    w, h, views, T, size = (int(v) for v in g[f"tanet_{case}_cfg"])
    from test_frames_cpu import pil_frames
    frames = np.stack([np.asarray(f) for f in pil_frames(views * T, w, h, 11)])
    random.seed(5)
    specs, boxes = [], []
    for _ in range(views):
        cw, ch, ow, oh = DV.sample_multiscale_crop((w, h), (size, size))
        boxes.append((ow, oh, cw, ch))
        specs.append(FR.ViewSpec((ow, oh, cw, ch), (size, size)))
    got, _ = run(frames, specs, (size, size), T)
    assert list(got.shape) == g[f"tanet_{case}_shape"].tolist()
    np.testing.assert_allclose(got[:, ::16, ::16], g[f"tanet_{case}_sub"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got.astype(np.float64).sum((1, 2)), g[f"tanet_{case}_chsum"], rtol=1e-9, atol=1e-3)
    np.testing.assert_array_equal(got, FO.clip_input(frames, boxes, T, (size, size), MEAN, STD))


@pytest.mark.parametrize("in_wh,boxes,fpv,out_wh,smooth", [
    ((340, 256), [(58, 16, 224, 224), (0, 0, 256, 256)], 3, (224, 224), True),
    ((340, 256), [(29, 8, 168, 192), (116, 32, 224, 168)], 2, (224, 224), False),  # up-scaling, full-range bytes
    ((720, 480), [(10, 20, 700, 450)], 4, (112, 112), False),                      # 15-tap windows
    ((1920, 1080), [(0, 0, 1920, 1080), (400, 100, 1080, 900)], 1, (224, 224), False),  # tile shrinks to fit LDS
    ((97, 61), [(3, 5, 90, 50)], 5, (31, 77), False),                             # odd sizes, ragged last tile
])
def test_kernel_is_bit_exact_against_the_oracle(in_wh, boxes, fpv, out_wh, smooth):
    frames = byte_frames(len(boxes) * fpv, *in_wh, seed=21, smooth=smooth)
    got, plan = run(frames, [FR.ViewSpec(b, out_wh) for b in boxes], out_wh, fpv)
    ref = FO.clip_input(frames, boxes, fpv, out_wh, MEAN, STD)
    np.testing.assert_array_equal(got, ref)
    assert plan.lds_rows * 3 * out_wh[0] <= FR.LDS_BYTES


def test_eval_view_scale_then_centre_crop():
    for (w, h) in [(320, 240), (240, 320), (340, 256)]:
        frames = byte_frames(8, w, h, seed=5, smooth=False)
        v = FR.eval_view((w, h), 256, 224)
        got, _ = run(frames, [v], (224, 224), 8)
        ref = FO.clip_input(frames, [v.box], 8, (224, 224), MEAN, STD, resize=v.resize, window=v.window)
        np.testing.assert_array_equal(got, ref)


def test_out_buffer_and_argument_checks():
    frames = torch.from_numpy(byte_frames(2, 64, 48, seed=1)).to(DEV)
    plan = FR.FramePlan([FR.ViewSpec((0, 0, 64, 48), (32, 32))], (32, 32), DEV, MEAN, STD)
    out = torch.full((6, 32, 32), float("nan"), device=DEV)
    assert FR.resample_normalise(frames, plan, 2, out=out) is out and torch.isfinite(out).all()
    with pytest.raises(ValueError):
        FR.resample_normalise(frames, plan, 1)  # 2 frames, 1 view of 1
    with pytest.raises(ValueError):
        FR.resample_normalise(frames[:, :40], plan, 2)  # crop box outside the frame
    with pytest.raises(Exception, match="uint8"):
        FR.resample_normalise(frames.float(), plan, 2)


@pytest.mark.parametrize("mode", ["tta", "eval"])
def test_dataset_transform_chain_on_device_is_the_host_chain(mode):
    """`tanet_clip_on_device` (what VideoTANetDataset runs with --device_preprocess) against the host PIL chain of the
    same dataset on the same decoded frames and the same `random` draws: equal bit for bit."""
    from PIL import Image
    T, views = 8, 2
    frames = byte_frames(views * T if mode == "tta" else T, 320, 240, seed=33)
    pil = [Image.fromarray(f) for f in frames]
    random.seed(17)
    if mode == "tta":
        host = DV.stack_to_tensor(DV.subgroup_multiscale_crop(pil, views, T, 224), MEAN, STD)
    else:
        host = DV.stack_to_tensor([DV.center_crop(DV.scale_short_edge(f, 256), 224) for f in pil], MEAN, STD)
    random.seed(17)
    dev = DV.tanet_clip_on_device(frames, DEV, T, 224, 256, MEAN, STD, tta_views=views if mode == "tta" else None)
    assert dev.is_cuda and torch.equal(dev.cpu(), host)


def test_dataset_with_device_preprocess_equals_the_host_dataset(tmp_path, monkeypatch):
    """`VideoTANetDataset(device_preprocess=...)` end to end (index sampling -> decode -> upload -> one launch) against
    the same dataset on the host PIL pipeline, with a stand-in for the uninstalled decoder: equal bit for bit, sample
    resident on the GPU, loader without worker processes."""
    from vitta_amd import tta as T
    fake = H.FakeDecord(n_frames=33)
    monkeypatch.setattr(DV, "_decord", lambda: fake)
    lst = tmp_path / "list.txt"
    lst.write_text("clipA 40 3\nclipB 33 7\n")
    kw = dict(vid_format=".mp4")
    for views in (2, None):
        extra = dict(tta_views=views, tta_styles=["uniform_equidist"]) if views else {}
        host = DV.VideoTANetDataset(str(lst), 8, str(tmp_path), **kw, **extra)
        dev = DV.VideoTANetDataset(str(lst), 8, str(tmp_path), device_preprocess=DEV, **kw, **extra)
        assert dev.on_device
        for i in range(2):
            random.seed(11 + i)
            xh, yh = host[i]
            random.seed(11 + i)
            xd, yd = dev[i]
            assert xd.is_cuda and yd == yh and torch.equal(xd.cpu(), xh)
    args = H.tanet_args(tmp_path, workers=4, batch_size=2)
    loader = T._loader(dev, args)
    assert loader.num_workers == 0
    xb, yb = next(iter(loader))
    assert xb.is_cuda and xb.shape == (2, 24, 224, 224) and yb.tolist() == [3, 7]


@pytest.mark.parametrize("h,w,scale,size,box", [(240, 320, 256, 224, None), (240, 320, 224, 224, (17, 9, 17 + 180, 9 + 150)),
                                                (128, 96, 64, 32, None), (100, 100, 200, 64, (0, 0, 100, 100)),
                                                (64, 48, 32, 16, (2, 4, 2 + 8, 4 + 8))])
def test_swin_cv2_pipeline_on_device_equals_host(h, w, scale, size, box):
    """vitta_frames_cv2_resize (two launches: short-edge rescale to uint8, crop + resize + normalise + NCTHW) against the host
    numpy path of the same restated cv2 arithmetic, bit for bit; the box cases include the exact-2x shortcut and a copy."""
    from vitta_amd import frames as F
    rng = np.random.RandomState(h + w + size)
    frames = rng.randint(0, 256, size=(6, h, w, 3)).astype(np.uint8)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    bx = (lambda nh, nw: box) if box is not None else None
    ref = F.swin_clip_host(frames, 2, 3, scale, size, bx, mean, std)
    got = F.swin_clip_on_device(frames, torch.device("cuda:0"), 2, 3, scale, size, bx, mean, std)
    assert got.shape == ref.shape == (2, 3, 3, size, size)
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())
