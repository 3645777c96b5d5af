"""Input pipelines against goldens produced by the reference's own transform classes /
samplers (tools/refgen/gen_golden.py data).  decord itself is not installed here: decode is not exercised."""
import random

import numpy as np
import pytest
import torch

import helpers as H
from vitta_amd import data_video as DV


def _frames(n, w, h, seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        base = rng.randint(0, 256, size=(h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
        out.append(Image.fromarray(base).resize((w, h), Image.BICUBIC))
    return out


@pytest.mark.parametrize("case", ["a", "b"])
def test_tanet_tta_transform_matches_reference(case):
    g = H.golden("data_pipeline.npz")
    w, h, views, T, size = (int(v) for v in g[f"tanet_{case}_cfg"])
    frames = _frames(views * T, w, h, 11)
    random.seed(5)  # same draws as the reference: (crop pair, offset) per view
    imgs = DV.subgroup_multiscale_crop(frames, views, T, size)
    ten = DV.stack_to_tensor(imgs, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert list(ten.shape) == g[f"tanet_{case}_shape"].tolist()
    torch.testing.assert_close(ten[:, ::16, ::16], torch.from_numpy(g[f"tanet_{case}_sub"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ten.double().sum((1, 2)), torch.from_numpy(g[f"tanet_{case}_chsum"]), rtol=1e-9, atol=1e-3)


def test_crop_candidates_and_eval_transform():
    offs = DV.fixed_crop_offsets(340, 256, 224, 224)
    assert len(offs) == 13 and offs[0] == (0, 0) and offs[4] == (2 * 29, 2 * 8) and offs[3] == (116, 32)
    random.seed(0)
    w, h, ow, oh = DV.sample_multiscale_crop((340, 256), (224, 224))
    assert w in (256, 224, 192, 168) and h in (256, 224, 192, 168) and 0 <= ow <= 340 - w and 0 <= oh <= 256 - h
    img = _frames(1, 340, 256, 3)[0]
    s = DV.scale_short_edge(img, 256)
    assert s.size == (340, 256)
    s2 = DV.scale_short_edge(_frames(1, 320, 240, 3)[0], 256)
    assert s2.size == (int(256 * 320 / 240), 256)
    c = DV.center_crop(s2, 224)
    assert c.size == (224, 224)


def test_swin_samplers_match_reference():
    g = H.golden("data_pipeline.npz")
    for key in g.files:
        parts = key.split("_")
        if key.startswith("swin_uniform"):
            T, n = int(parts[2][1:]), int(parts[3][1:])
            np.testing.assert_array_equal(DV.swin_uniform_indices(n, T), g[key], err_msg=key)
        elif key.startswith("swin_dense"):
            T, n, c = int(parts[2][1:]), int(parts[3][1:]), int(parts[4][1:])
            np.testing.assert_array_equal(DV.swin_dense_test_indices(n, T, 2, c), g[key], err_msg=key)


def test_random_resized_crop_box_is_inside_the_frame():
    rng = np.random.RandomState(0)
    for _ in range(50):
        l, t, r, b = DV.random_resized_crop_box(224, 298, rng=rng)
        assert 0 <= l < r <= 298 and 0 <= t < b <= 224


def test_real_video_datasets_need_decord(tmp_path):
    lst = tmp_path / "list.txt"
    lst.write_text("a.mp4 100 3\nb.mp4 2 1\n")
    try:
        import decord  # noqa: F401
        pytest.skip("decord present")
    except ImportError:
        pass
    with pytest.raises(ImportError):
        DV.VideoTANetDataset(str(lst), 8, str(tmp_path))


def _list_file(tmp_path):
    lst = tmp_path / "list.txt"
    lst.write_text("clipA 40 3\nclipB 33 7\n")
    return str(lst)


def test_tanet_dataset_host_pipeline_with_a_stand_in_decoder(tmp_path, monkeypatch):
    """The dataset class end to end on the host (index sampling -> decode -> PIL transforms -> stacked clip) with a
    stand-in for the uninstalled decoder: shapes, labels, the frames asked of the decoder, determinism under a seed."""
    fake = H.FakeDecord(n_frames=33)  # the list says 40 for clipA: indices are clamped to the decoder's own length
    monkeypatch.setattr(DV, "_decord", lambda: fake)
    tta = DV.VideoTANetDataset(_list_file(tmp_path), 8, str(tmp_path), vid_format=".mp4", tta_views=2,
                               tta_styles=["uniform_equidist"])
    ev = DV.VideoTANetDataset(_list_file(tmp_path), 8, str(tmp_path), vid_format=".mp4")
    assert len(tta) == 2 and not tta.on_device
    random.seed(3)
    x, y = tta[0]
    assert x.shape == (2 * 8 * 3, 224, 224) and x.dtype == torch.float32 and y == 3
    assert fake.opened[-1].endswith("clipA.mp4")
    random.seed(3)
    x2, _ = tta[0]
    assert torch.equal(x, x2)
    e, y = ev[1]
    assert e.shape == (8 * 3, 224, 224) and y == 7
    idx = tta.frame_indices(40)
    assert len(idx) == 16 and idx.max() == 40  # 1-based; the dataset clamps to the decoder's last frame (video_dataset.py:328)


def test_swin_dataset_host_pipeline_equals_oracle(tmp_path, monkeypatch):
    """VideoSwinDataset end to end on the host (index sampling -> decode (stand-in) -> Resize((-1, scale)) -> CenterCrop /
    RandomResizedCrop + Resize -> Normalize -> NCTHW; video_dataset.py:60-101) against the oracle restatement of the same
    chain (cv2 arithmetic: unpinned, see oracle/frames_oracle.py)."""
    import numpy as np
    from oracle import frames_oracle as FO
    fake = H.FakeDecord(n_frames=37, w=96, h=72)
    monkeypatch.setattr(DV, "_decord", lambda: fake)
    lst = tmp_path / "list.txt"
    lst.write_text("clipA 37 3\nclipB 37 7\n")
    cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])
    fixed = (6, 4, 6 + 50, 4 + 44)
    monkeypatch.setattr(DV, "random_resized_crop_box", lambda nh, nw, **kw: fixed)
    for views in (None, 2):
        extra = dict(tta_views=views, tta_styles=["uniform_equidist"]) if views else {}
        ds = DV.VideoSwinDataset(str(lst), 4, str(tmp_path), vid_format=".mp4", scale_size=64, input_size=48, img_norm_cfg=cfg, **extra)
        x, y = ds[1]
        n = views or 1
        assert y == 7 and tuple(x.shape) == (n, 3, 4, 48, 48)
        reader = fake.VideoReader(str(tmp_path / "clipB.mp4"))
        idx, _ = ds.frame_indices(len(reader))
        frames = reader.get_batch(np.minimum(idx, len(reader) - 1)).asnumpy()
        ref = FO.swin_clip(frames, n, 4, 64, 48, (lambda nh, nw: fixed) if views else None, cfg["mean"], cfg["std"])
        np.testing.assert_array_equal(x.numpy(), ref)
