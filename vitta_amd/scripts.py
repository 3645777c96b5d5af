"""Shared body of the entry scripts (tta_*_ucf101.py, sourceonly_*_corr.py, compute_stats/*).

The reference scripts (tta_tanet_ucf101.py:13-45 and siblings) mutate the parsed Namespace and loop
over the 12 corruption lists.  Two of their bugs are fixed here on purpose, not reproduced:
  * `args.val_vid_list = args.val_vid_list.format(corruption)` overwrites the `{}` template, so every
    corruption after the first silently re-uses the first list (tta_tanet_ucf101.py:33-34): the
    templates are kept and formatted per corruption;
  * the source-only scripts assign the (list, model) tuple returned by eval() to `epoch_result_list` and
    float() its items (sourceonly_tanet_ucf101_corr.py:40-44): the tuple is unpacked.
"""
from .main_eval import eval as run_eval
from .opts import get_opts
from .utils_ import get_writer_to_all_result

CORRUPTIONS = ["gauss_shuffled", "pepper_shuffled", "salt_shuffled", "shot_shuffled", "zoom_shuffled",
               "impulse_shuffled", "defocus_shuffled", "motion_shuffled", "jpeg_shuffled", "contrast_shuffled",
               "rain_shuffled", "h265_abr_shuffled"]


def _rank():
    import torch
    d = torch.distributed
    return d.get_rank() if d.is_available() and d.is_initialized() else 0


def run_over_corruptions(args, corruptions=CORRUPTIONS):
    """eval() once per corruption; one line of rounded top-1 per corruption in <result_dir>/<time>_all_result."""
    list_template, dir_template = args.val_vid_list, args.result_dir
    f_write, results = None, []
    for args.corruptions in corruptions:
        print(f"####Starting Evaluation for ::: {args.corruptions} corruption####")
        args.val_vid_list = list_template.format(args.corruptions)
        args.result_dir = dir_template.format(args.arch, args.dataset, args.corruptions)
        epoch_result_list, _ = run_eval(args=args)
        if f_write is None and _rank() == 0:  # data-parallel runs: one result file, written by rank 0
            f_write = get_writer_to_all_result(args)
        if epoch_result_list is not None and f_write is not None:
            f_write.write(" ".join(str(round(float(x), 3)) for x in epoch_result_list) + "\n")
            f_write.flush()
        results.append(epoch_result_list)
    if f_write is not None:
        f_write.close()
    args.val_vid_list, args.result_dir = list_template, dir_template
    return results


def tanet_ucf101_args(argv=None):
    args = get_opts(argv)
    args.gpus, args.arch, args.dataset = [0], "tanet", "ucf101"
    return args


def swin_ucf101_args(argv=None):
    """Overrides of tta_swin_ucf101.py:27-40."""
    args = get_opts(argv)
    args.gpus, args.arch, args.dataset = [0], "videoswintransformer", "ucf101"
    args.clip_length, args.num_clips, args.test_crops = 16, 1, 1
    args.frame_uniform, args.frame_interval, args.scale_size = True, 2, 224
    args.patch_size, args.window_size = (2, 4, 4), (8, 7, 7)
    args.lr, args.lambda_pred_consis, args.momentum_mvg = 0.00001, 0.05, 0.05
    args.chosen_blocks = ["module.backbone.layers.2", "module.backbone.layers.3", "module.backbone.norm"]
    return args


def source_only(args):
    """sourceonly_*_corr.py: tta off, baseline 'source', batch 32."""
    args.tta, args.evaluate_baselines, args.baseline = False, True, "source"
    args.batch_size = 32
    return args


def compute_stats(args):
    """compute_stats/*.py: statistics of the clean training videos (stat_type is a str here)."""
    args.tta, args.compute_stat, args.stat_type = True, "mean_var", "spatiotemp"
    args.batch_size = 32
    return args
