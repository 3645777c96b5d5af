"""Test-video sources for the TTA driver.

* frame-index samplers of the TANet dataset, restated from
  models/tanet_models/video_dataset.py:159-230 (_sample_tta_augmented_views) and :271-303
  (_get_test_indices); 1-based like the reference (decord indexing subtracts nothing: the reference
  clamps to num_frames-1 at :328);
* video-list parsing (:144-152);
* SyntheticVideoDataset -- seeded N(0,1) clips in the exact tensor layouts the two reference datasets
  return (TANet: [V*T*3, H, W] / [T*3, H, W]; Swin: [V, 3, T, H, W] / [1, 3, T, H, W]); used by bench.py
  and the tests (no dataset or checkpoint is reachable offline);
* the decord/PIL pipelines are built in data_video.py when decord exists.
"""
import numpy as np
import torch


class VideoRecord:
    """One line '<path> <n_frames> <class_id>' of a video list."""

    def __init__(self, row):
        self._data = row

    @property
    def path(self):
        return self._data[0]

    @property
    def num_frames(self):
        return int(self._data[1])

    @property
    def label(self):
        return int(self._data[2])


def parse_video_list(list_file, remove_missing=True, debug=False, debug_vid=50):
    rows = [line.strip().split(" ") for line in open(list_file) if line.strip()]
    if remove_missing:
        rows = [r for r in rows if int(r[1]) >= 3]
    records = [VideoRecord(r) for r in rows]
    return records[:debug_vid] if debug else records


# np.linspace(start, stop, num, dtype=int): numpy 1.19.5 (the reference's pin, requirements.txt)
# truncates toward zero, NumPy >= 1.20 floors.  They differ only for a negative `stop`, i.e. videos
# with fewer frames than T.  'trunc' reproduces the environment the reference was published with;
# 'floor' reproduces the reference code executed on a current NumPy (what the golden vectors, captured
# here with numpy 2.2, contain).
LINSPACE_INT_MODE = "trunc"


def _linspace_trunc(start, stop, num):
    vals = np.linspace(start, stop, num=num)
    if LINSPACE_INT_MODE == "floor":
        return [int(np.floor(v)) for v in vals]
    return [int(v) for v in vals]


def tta_view_indices(num_frames, num_segments, n_views, style="uniform_equidist", new_length=1, rng=None):
    """Frame indices (1-based, all views concatenated) of the temporally augmented TTA views."""
    T = num_segments
    if style == "uniform":
        tick = (num_frames - new_length + 1) / float(T)
        return np.array([int(tick / 2.0 + tick * x) for x in range(T)]) + 1
    if style == "dense":
        t_stride = 64 // T
        sample_pos = max(1, 1 + num_frames - t_stride * T)
        start = sample_pos // 2
        return np.array([(i * t_stride + start) % num_frames for i in range(T)]) + 1
    if style == "uniform_equidist":
        tick = (num_frames - new_length + 1) / float(T)
        offsets = []
        for start in _linspace_trunc(0, tick - 1, n_views):
            offsets += [int(start + tick * x) % num_frames for x in range(T)]
        return np.array(offsets) + 1
    if style == "dense_equidist":
        t_stride = 64 // T
        sample_pos = max(1, 1 + num_frames - t_stride * T)
        offsets = []
        for start in _linspace_trunc(0, sample_pos - 1, n_views):
            offsets += [(i * t_stride + start) % num_frames for i in range(T)]
        return np.array(offsets) + 1
    rng = rng or np.random
    if style == "uniform_rand":
        avg = (num_frames - new_length + 1) // T
        if avg > 0:
            return np.multiply(list(range(T)), avg) + rng.randint(avg, size=T) + 1
        if num_frames > T:
            return np.sort(rng.randint(num_frames - new_length + 1, size=T)) + 1
        return np.zeros((T,)) + 1
    if style == "dense_rand":
        t_stride = 64 // T
        sample_pos = max(1, 1 + num_frames - t_stride * T)
        start = 0 if sample_pos == 1 else rng.randint(0, sample_pos - 1)
        return np.array([(i * t_stride + start) % num_frames for i in range(T)]) + 1
    if style == "random":
        if num_frames >= T:
            return np.sort(rng.choice(num_frames, size=T, replace=False))
        return np.array(list(range(num_frames)) + [num_frames - 1] * (T - num_frames))
    raise NotImplementedError(f"{style} not exist")


def test_indices(num_frames, num_segments, test_sample="uniform-1", new_length=1):
    """Frame indices (1-based) of the evaluation clip(s)."""
    T = num_segments
    num_clips = int(test_sample.split("-")[-1])
    if "dense" in test_sample:
        t_stride = 64 // T
        sample_pos = max(1, 1 + num_frames - t_stride * T)
        starts = [sample_pos // 2] if num_clips == 1 else _linspace_trunc(0, sample_pos - 1, num_clips)
        return np.array([(i * t_stride + s) % num_frames for s in starts for i in range(T)]) + 1
    if "uniform" in test_sample:
        tick = (num_frames - new_length + 1) / float(T)
        if num_clips == 1:
            return np.array([int(tick / 2.0 + tick * x) for x in range(T)]) + 1
        return np.array([int(s + tick * x) % num_frames for s in _linspace_trunc(0, tick - 1, num_clips)
                         for x in range(T)]) + 1
    raise NotImplementedError(f"{test_sample} not exist")


class SyntheticVideoDataset(torch.utils.data.Dataset):
    """Seeded synthetic clips: video i is N(0,1) noise from torch.Generator(seed0 + i).

    layout 'tanet': [n_views*T*3, H, W]; layout 'swin': [n_views, 3, T, H, W].  With `device` set the
    clips are generated once, kept resident in HBM and returned as device tensors (bench.py: inputs
    are already in HBM when the timed region starts)."""

    def __init__(self, n_videos, n_views, clip_length, size, num_classes, layout="tanet", seed0=0, device=None):
        self.n_videos, self.n_views, self.T, self.size = n_videos, n_views, clip_length, size
        self.num_classes, self.layout, self.seed0 = num_classes, layout, seed0
        self.device = torch.device(device) if device is not None else None
        self.on_device = self.device is not None and self.device.type == "cuda"
        self._cache = [self._make(i) for i in range(n_videos)] if self.on_device else None

    def _shape(self):
        if self.layout == "tanet":
            return (self.n_views * self.T * 3, self.size, self.size)
        return (self.n_views, 3, self.T, self.size, self.size)

    def _make(self, i):
        g = torch.Generator().manual_seed(self.seed0 + i)
        x = torch.randn(self._shape(), generator=g)
        y = torch.randint(self.num_classes, (1,), generator=g)[0]
        if self.on_device:
            return x.to(self.device), y.to(self.device)
        return x, y

    def __len__(self):
        return self.n_videos

    def __getitem__(self, i):
        return self._cache[i] if self._cache is not None else self._make(i)


def _synthetic(args, dataset_type, layout):
    views = args.n_augmented_views if (dataset_type == "tta" and args.if_sample_tta_aug_views) else 1
    size = args.scale_size if (args.full_res and layout == "tanet") else args.input_size
    return SyntheticVideoDataset(getattr(args, "synthetic_n_videos", 64), views, args.clip_length, size,
                                 args.num_classes, layout=layout, seed0=getattr(args, "synthetic_seed", 0),
                                 device=getattr(args, "synthetic_device", None))


def build_tanet_dataset(args, split="train", dataset_type=None):
    if split != "val":
        raise NotImplementedError("Training dataset processing for TANet to be added!")
    if args.datatype == "synthetic":
        return _synthetic(args, dataset_type, "tanet")
    from . import data_video
    return data_video.tanet_video_dataset(args, dataset_type)


def build_videoswin_dataset(args, split="train", dataset_type=None):
    if split != "val":
        raise NotImplementedError("Training dataset processing for Video Swin Transformer to be added!")
    if args.datatype == "synthetic":
        return _synthetic(args, dataset_type, "swin")
    from . import data_video
    return data_video.swin_video_dataset(args, dataset_type)
