"""Real-video input pipelines of the two model families.

TANet (models/tanet_models/video_dataset.py:305-341 + transforms.py):
    decord decode at the sampled indices (clamped to n_frames-1) -> PIL RGB ->
    'tta' : per-view multi-scale crop (scales 1/.875/.75/.66, 13 fixed offsets, transforms.py:277-384)
            resized to input_size with PIL BILINEAR
    'eval': resize smaller edge to scale_size (BILINEAR) + centre crop input_size (transforms.py:46-54,170-184)
    -> stack frames on the channel axis, /255, normalise per channel -> [V*T*3, H, W] float32.
  PIL's BILINEAR (with its reducing filter) is what defines "identical inputs"; tests/test_data_video.py checks every
  host step against the reference's own transform classes.  With `device_preprocess` (extension, --device_preprocess)
  the decoded uint8 frames are uploaded as they are and crop + resize + stack + normalise run in ONE HIP launch that is
  bit-identical to the PIL path (vitta_amd/frames.py, tests/test_gpu_frames.py): a quarter of the PCIe bytes, no
  per-frame PIL work on the host.
  The crop/offset draws consume `random` exactly like the reference (one choice of the (w, h) pair, one choice
  of the offset, per view), so a seeded run reproduces the reference's crops.

Video Swin (models/videoswintransformer_models/video_dataset.py:59-107 + transforms_backup.py):
    frame indices: the same 1-based view sampler for 'tta' (transforms_backup.py:571-650), uniform middle-of-segment
    sampling for 'eval' (get_seq_frames :549-568), clamped to n_frames-1 (:700); decode; Resize(short edge
    scale_size) -> RandomResizedCrop (area .08-1, ratio 3/4-4/3, ONE crop for all views) or CenterCrop -> Resize
    to input_size -> Normalize(mean/std on 0-255) -> [V, 3, T, H, W].
  The reference resizes with mmcv.imresize (cv2, not installed here): the Swin image ops restate cv2's 8-bit INTER_LINEAR
  resampler (vitta_amd/frames.py::cv2_resize_linear on the host, vitta_frames_cv2_resize with --device_preprocess; the two
  are bit-identical to each other and to oracle/frames_oracle.py, NOT pinned against cv2 itself; index sampler and
  crop-box arithmetic are pinned).

decord is imported lazily: constructing a dataset without it raises with a clear message.
"""
import math
import os.path as osp
import random

import numpy as np
import torch

from . import data
from . import frames as F


def _decord():
    try:
        import decord
        return decord
    except ImportError as e:  # pragma: no cover - decord is not installed in the build image
        raise ImportError("decord is required to read real videos (pip install decord==0.6.0); "
                          "use --datatype synthetic for seeded synthetic clips") from e


# ------------------------------------------------------------------------------------------------
# TANet image transforms (PIL)
# ------------------------------------------------------------------------------------------------
SCALES = (1, .875, .75, .66)


def fixed_crop_offsets(image_w, image_h, crop_w, crop_h, more_fix_crop=True):
    """The 5 (+8) candidate offsets of the multi-scale crop."""
    ws, hs = (image_w - crop_w) // 4, (image_h - crop_h) // 4
    grid = [(0, 0), (4, 0), (0, 4), (4, 4), (2, 2)]
    if more_fix_crop:
        grid += [(0, 2), (4, 2), (2, 4), (2, 0), (1, 1), (3, 1), (1, 3), (3, 3)]
    return [(a * ws, b * hs) for a, b in grid]


def sample_multiscale_crop(im_size, input_size, scales=SCALES, max_distort=1, rng=random):
    """(crop_w, crop_h, offset_w, offset_h): one draw of the crop size pair, one draw of the offset."""
    image_w, image_h = im_size
    in_w, in_h = input_size
    base = min(image_w, image_h)
    sizes = [int(base * s) for s in scales]
    crop_h = [in_h if abs(x - in_h) < 3 else x for x in sizes]
    crop_w = [in_w if abs(x - in_w) < 3 else x for x in sizes]
    pairs = [(w, h) for i, h in enumerate(crop_h) for j, w in enumerate(crop_w) if abs(i - j) <= max_distort]
    w, h = rng.choice(pairs)
    ow, oh = rng.choice(fixed_crop_offsets(image_w, image_h, w, h))
    return w, h, ow, oh


def subgroup_multiscale_crop(images, n_views, clip_len, input_size, rng=random):
    """Per temporal view: one random multi-scale crop applied to all frames of the view, resized (BILINEAR)."""
    from PIL import Image
    assert len(images) == n_views * clip_len
    size = (input_size, input_size) if isinstance(input_size, int) else tuple(input_size)
    out = []
    for v in range(n_views):
        w, h, ow, oh = sample_multiscale_crop(images[0].size, size, rng=rng)
        for img in images[v * clip_len:(v + 1) * clip_len]:
            out.append(img.crop((ow, oh, ow + w, oh + h)).resize(size, Image.BILINEAR))
    return out


def scale_short_edge(img, size):
    """torchvision.transforms.Resize(int) of 0.8.2: smaller edge -> size, aspect kept, BILINEAR."""
    from PIL import Image
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        return img.resize((size, int(size * h / w)), Image.BILINEAR)
    return img.resize((int(size * w / h), size), Image.BILINEAR)


def center_crop(img, size):
    w, h = img.size
    th = tw = size
    top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    return img.crop((left, top, left + tw, top + th))


def stack_to_tensor(images, mean, std):
    """[F*3, H, W] float32: frames stacked on the channel axis, /255, (x - mean_c) / std_c per RGB channel."""
    arr = np.concatenate([np.asarray(im) for im in images], axis=2)  # H, W, F*3
    t = torch.from_numpy(arr).permute(2, 0, 1).contiguous().float().div(255)
    reps = t.shape[0] // len(mean)
    m = torch.tensor(list(mean) * reps, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(list(std) * reps, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(m).div_(s)


def tanet_clip_on_device(frames_u8, device, clip_len, input_size, scale_size, mean, std, tta_views=None, rng=random,
                         lut=None):
    """The transform chain of `VideoTANetDataset.__getitem__` on the device: frames_u8 = the decoded uint8 frames
    [F, H, W, 3] (numpy or tensor).  tta_views: per-view multi-scale crop (same `random` draws, in the same order, as
    the host pipeline) resized to input_size; None: short edge -> scale_size, centre crop.  -> [F*3, S, S] float32 on
    `device`, bit-identical to stack_to_tensor(PIL path)."""
    from . import frames as FR
    t = torch.as_tensor(frames_u8)
    f, h, w, _ = t.shape
    size = (input_size, input_size) if isinstance(input_size, int) else tuple(input_size)
    if tta_views:
        assert f == tta_views * clip_len
        views = []
        for _ in range(tta_views):
            cw, ch, ow, oh = sample_multiscale_crop((w, h), size, rng=rng)
            views.append(FR.ViewSpec((ow, oh, cw, ch), size))
        per_view = clip_len
    else:
        views, per_view = [FR.eval_view((w, h), scale_size, size[0])], f
    plan = FR.FramePlan(views, size, device, mean, std, lut=lut)
    return FR.resample_normalise(t.to(device, non_blocking=True), plan, per_view)


class VideoTANetDataset(torch.utils.data.Dataset):
    def __init__(self, list_file, num_segments, video_data_dir, vid_format="", input_size=224, scale_size=256,
                 input_mean=(0.485, 0.456, 0.406), input_std=(0.229, 0.224, 0.225), test_sample="uniform-1",
                 tta_views=None, tta_styles=None, spatial_rand_cropping=True, test_crops=1, debug=False,
                 device_preprocess=None):
        if test_crops != 1:
            raise NotImplementedError(f"{test_crops} spatial crops not implemented!")
        # device_preprocess = a torch device: samples come back device-resident (the loader then runs without workers,
        # tta._loader); None: the host PIL pipeline
        self.device = torch.device(device_preprocess) if device_preprocess is not None else None
        self.on_device = self.device is not None
        self._lut = None
        self.records = data.parse_video_list(list_file, remove_missing=True, debug=debug)
        self.T, self.dir, self.fmt = num_segments, video_data_dir, vid_format
        self.input_size, self.scale_size, self.mean, self.std = input_size, scale_size, input_mean, input_std
        self.test_sample, self.tta_views, self.tta_styles = test_sample, tta_views, tta_styles
        self.spatial_rand_cropping = spatial_rand_cropping
        self._decord = _decord()

    def __len__(self):
        return len(self.records)

    def frame_indices(self, n_frames):
        if self.tta_views:
            idx = []
            for style in self.tta_styles:
                idx += list(data.tta_view_indices(n_frames, self.T, self.tta_views, style))
            return np.asarray(idx)
        return np.asarray(data.test_indices(n_frames, self.T, self.test_sample))

    def __getitem__(self, i):
        from PIL import Image
        rec = self.records[i]
        reader = self._decord.VideoReader(osp.join(self.dir, f"{rec.path}{self.fmt}"))
        idx = np.minimum(self.frame_indices(rec.num_frames), len(reader) - 1).astype(np.int64)
        if self.on_device:
            if self._lut is None:
                from . import frames as FR
                self._lut = FR.normalise_table(self.mean, self.std).to(self.device)
            tta = self.tta_views if (self.tta_views and self.spatial_rand_cropping) else None
            clip = tanet_clip_on_device(reader.get_batch(idx).asnumpy(), self.device, self.T, self.input_size,
                                        self.scale_size, self.mean, self.std, tta_views=tta, lut=self._lut)
            return clip, rec.label
        frames = [Image.fromarray(f).convert("RGB") for f in reader.get_batch(idx).asnumpy()]
        if self.tta_views and self.spatial_rand_cropping:
            frames = subgroup_multiscale_crop(frames, self.tta_views, self.T, self.input_size)
        else:
            frames = [center_crop(scale_short_edge(f, self.scale_size), self.input_size) for f in frames]
        return stack_to_tensor(frames, self.mean, self.std), rec.label


def _preprocess_device(args):
    if not getattr(args, "device_preprocess", False):
        return None
    dev = getattr(args, "device", None)
    return torch.device(dev) if dev is not None else torch.device("cuda", torch.cuda.current_device())


def tanet_video_dataset(args, dataset_type):
    tta = dataset_type == "tta" and args.if_sample_tta_aug_views
    input_size = args.scale_size if args.full_res else args.input_size
    return VideoTANetDataset(args.val_vid_list, args.clip_length, args.video_data_dir, vid_format=args.vid_format,
                             input_size=input_size, scale_size=args.scale_size, input_mean=args.input_mean,
                             input_std=args.input_std, test_sample=args.sample_style,
                             tta_views=args.n_augmented_views if tta else None,
                             tta_styles=args.tta_view_sample_style_list if tta else None,
                             spatial_rand_cropping=args.if_spatial_rand_cropping if tta else False,
                             test_crops=args.test_crops, debug=args.debug,
                             device_preprocess=_preprocess_device(args))


# ------------------------------------------------------------------------------------------------
# Video Swin
# ------------------------------------------------------------------------------------------------
def swin_uniform_indices(num_frames, clip_len):
    """get_seq_frames in test mode: the middle frame of each of clip_len equal segments (0-based)."""
    seg = float(num_frames - 1) / clip_len
    return np.array([(int(np.round(seg * i)) + int(np.round(seg * (i + 1)))) // 2 for i in range(clip_len)])


def swin_dense_test_indices(num_frames, clip_len, frame_interval, num_clips):
    """SampleFrames test clips (mmaction): clip starts evenly spread, indices looped modulo the length."""
    ori = clip_len * frame_interval
    avg = (num_frames - ori + 1) / float(num_clips)
    if num_frames > ori - 1:
        offsets = (np.arange(num_clips) * avg + avg / 2.0).astype(np.int64)
    else:
        offsets = np.zeros((num_clips,), dtype=np.int64)
    idx = offsets[:, None] + np.arange(clip_len)[None, :] * frame_interval
    return np.mod(idx, num_frames).reshape(-1)


def random_resized_crop_box(img_h, img_w, area_range=(0.08, 1.0), aspect_ratio_range=(3 / 4, 4 / 3), max_attempts=10,
                            rng=np.random):
    """mmaction RandomResizedCrop.get_crop_bbox: (left, top, right, bottom)."""
    area = img_h * img_w
    lo, hi = aspect_ratio_range
    ratios = np.exp(rng.uniform(np.log(lo), np.log(hi), size=max_attempts))
    target = rng.uniform(*area_range, size=max_attempts) * area
    cw = np.round(np.sqrt(target * ratios)).astype(np.int32)
    ch = np.round(np.sqrt(target / ratios)).astype(np.int32)
    for w, h in zip(cw, ch):
        if w <= img_w and h <= img_h:
            x, y = rng.randint(0, img_w - w + 1), rng.randint(0, img_h - h + 1)
            return x, y, x + w, y + h
    s = min(img_h, img_w)
    x, y = (img_w - s) // 2, (img_h - s) // 2
    return x, y, x + s, y + s


class VideoSwinDataset(torch.utils.data.Dataset):
    def __init__(self, list_file, clip_len, video_data_dir, vid_format="", frame_interval=2, num_clips=1,
                 frame_uniform=True, scale_size=224, input_size=224, img_norm_cfg=None, tta_views=None, tta_styles=None,
                 debug=False, device_preprocess=None):
        self.records = data.parse_video_list(list_file, remove_missing=False, debug=debug)
        self.T, self.dir, self.fmt = clip_len, video_data_dir, vid_format
        self.frame_interval, self.num_clips, self.frame_uniform = frame_interval, num_clips, frame_uniform
        self.scale_size, self.input_size = scale_size, input_size
        cfg = img_norm_cfg or dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])
        self.mean = torch.tensor(cfg["mean"], dtype=torch.float32).view(3, 1, 1, 1)
        self.std = torch.tensor(cfg["std"], dtype=torch.float32).view(3, 1, 1, 1)
        self.tta_views, self.tta_styles = tta_views, tta_styles
        # device_preprocess = a torch device: the uint8 frames are uploaded and resized / normalised there (two launches of
        # vitta_frames_cv2_resize, bit-identical to the host path below); None: the host numpy path
        self.device = torch.device(device_preprocess) if device_preprocess is not None else None
        self._decord = _decord()

    def __len__(self):
        return len(self.records)

    def frame_indices(self, total):
        if self.tta_views:
            idx = []
            for style in self.tta_styles:
                idx += list(data.tta_view_indices(total, self.T, self.tta_views, style))
            return np.asarray(idx), self.tta_views
        if self.frame_uniform:
            return swin_uniform_indices(total, self.T), self.num_clips
        return swin_dense_test_indices(total, self.T, self.frame_interval, self.num_clips), self.num_clips

    def __getitem__(self, i):
        rec = self.records[i]
        reader = self._decord.VideoReader(osp.join(self.dir, f"{rec.path}{self.fmt}"))
        idx, views = self.frame_indices(len(reader))
        idx = np.minimum(idx, len(reader) - 1).astype(np.int64)
        fr = reader.get_batch(idx).asnumpy()  # F, H, W, 3 uint8
        box = None
        if self.tta_views:  # ONE RandomResizedCrop box for all frames of the sample, drawn on the rescaled size
            h, w = fr.shape[1:3]
            nh, nw = F.swin_scaled_size(h, w, self.scale_size)
            fixed = random_resized_crop_box(nh, nw)
            box = lambda _nh, _nw: fixed
        mean, std = self.mean.flatten().tolist(), self.std.flatten().tolist()
        if self.device is not None:
            x = F.swin_clip_on_device(fr, self.device, views, self.T, self.scale_size, self.input_size, box, mean, std)
        else:
            x = F.swin_clip_host(fr, views, self.T, self.scale_size, self.input_size, box, mean, std)
        return x, rec.label


def swin_video_dataset(args, dataset_type):
    tta = dataset_type == "tta" and args.if_sample_tta_aug_views
    return VideoSwinDataset(args.val_vid_list, args.clip_length, args.video_data_dir, vid_format=args.vid_format,
                            frame_interval=args.frame_interval, num_clips=args.num_clips,
                            frame_uniform=args.frame_uniform, scale_size=args.scale_size, input_size=args.input_size,
                            img_norm_cfg=args.img_norm_cfg, tta_views=args.n_augmented_views if tta else None,
                            tta_styles=args.tta_view_sample_style_list if tta else None, debug=args.debug,
                            device_preprocess=_preprocess_device(args))
