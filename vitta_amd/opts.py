"""Configuration surface of ViTTA -- every flag name and default of utils/opts.py:11-121, plus
`get_opts()` (:126-132).  Table-driven; differences from the reference, all deliberate:

* flags the reference declares with `type=bool` (any non-empty string parsed as True,
  opts.py:52,67,72-75,84-86,93,97) parse "false"/"0"/"no" as False here; defaults unchanged;
* `get_opts(argv=None)` accepts an explicit argv (the reference always reads sys.argv);
* `--datatype synthetic` (extension) selects the seeded synthetic video source used by bench.py;
* `--hip_graph` (extension, default True) lets tta_standard replay the step from captured hipGraphs.
* `--overlap_eval` (extension, default True) runs the evaluation of a video beside the next video's adaptation.
* `--dense_bf16` (extension, default False) runs Video Swin's dense layers (qkv / proj / MLP / patch merging) on bf16 MFMA
  operands with fp32 accumulation (vitta_amd/csrc/gemm.hip); the default is the exact-fp32 kernel of the same file.
* `--wmsa_bf16` (extension, default False) runs Video Swin's window attention on the bf16-operand kernels (fp32 softmax and
  accumulation; BASELINE config 5's recipe), vitta_amd/csrc/wmsa_bf16.hip.
* `--device_preprocess` (extension, default False) uploads the decoded uint8 frames and runs crop / resize / normalise
  of the TANet pipeline in one HIP launch (bit-identical to the PIL path, vitta_amd/frames.py).
* `--prefetch_input` (extension, default True) uploads the next video on a copy stream while the current one is adapted
  (vitta_amd/prefetch.py); False: the reference's upload in front of each step.
"""
import argparse

# TANet normalisation (0-1 range) and Video Swin normalisation (0-255 range)
input_mean = [0.485, 0.456, 0.406]
input_std = [0.229, 0.224, 0.225]
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)


def _bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ("1", "true", "t", "yes", "y", "on"):
        return True
    if s in ("0", "false", "f", "no", "n", "off", ""):
        return False
    raise argparse.ArgumentTypeError(f"expected a boolean, got {v!r}")


_REF = "/home/ivanl/data/UCF-HMDB"
_STAT_DIR = _REF + "/UCF-HMDB_all/corruptions_results/source/tanet_ucf101/compute_norm_spatiotempstats_clean_train_bn2d/"

# (flags, kwargs) in the reference's order
_FLAGS = [
    # data
    (("--dataset",), dict(type=str, default="ucf101", choices=["ucf101", "somethingv2", "kinetics"])),
    (("--modality",), dict(type=str, default="RGB")),
    (("--root_path",), dict(type=str, default="None")),
    (("--video_data_dir",), dict(type=str, default=_REF + "/video_pertubations/UCF101/level_5_ucf_val_split_1",
                                 help="directory of the corrupted videos")),
    (("--vid_format",), dict(type=str, default="", help="video suffix when the list omits it")),
    (("--datatype",), dict(type=str, default="vid", choices=["vid", "frame", "synthetic"])),
    (("--spatiotemp_mean_clean_file",), dict(type=str, default=_STAT_DIR + "list_spatiotemp_mean_20220908_235138.npy",
                                             help="spatiotemporal statistics - mean")),
    (("--spatiotemp_var_clean_file",), dict(type=str, default=_STAT_DIR + "list_spatiotemp_var_20220908_235138.npy",
                                            help="spatiotemporal statistics - variance")),
    (("--val_vid_list",), dict(type=str, default=_REF + "/video_pertubations/UCF101/list_video_perturbations/{}.txt",
                               help="list of corrupted videos, named after the corruption")),
    (("--result_dir",), dict(type=str, default=_REF + "/UCF-HMDB_all/corruptions_results/source/{}_{}/tta_{}",
                             help="result directory")),
    # model
    (("--arch",), dict(type=str, default="tanet", choices=["tanet", "videoswintransformer"], help="network architecture")),
    (("--model_path",), dict(type=str, default="/home/ivanl/data/DeepInversion_results/train_models/models/UCF/tanet/20220815_122340_ckpt.pth.tar")),
    (("--img_feature_dim",), dict(type=int, default=256, help="dimension of image feature on ResNet50")),
    (("--partial_bn",), dict(action="store_true")),
    # Video Swin
    (("--num_clips",), dict(type=int, default=1, help="number of temporal clips")),
    (("--frame_uniform",), dict(type=_bool, default=True, help="uniform (True) or dense sampling")),
    (("--frame_interval",), dict(type=int, default=2)),
    (("--flip_ratio",), dict(type=int, default=0)),
    (("--img_norm_cfg",), dict(default=img_norm_cfg)),
    (("--patch_size",), dict(default=(2, 4, 4))),
    (("--window_size",), dict(default=(8, 7, 7))),
    (("--drop_path_rate",), dict(default=0.2)),
    # runtime
    (("--gpus",), dict(nargs="+", type=int, default=None)),
    (("-j", "--workers"), dict(default=8, type=int, metavar="N", help="number of data loading workers")),
    (("--norm",), dict(action="store_true")),
    (("--debug",), dict(action="store_true", help="load only the first 50 videos of the list")),
    (("--verbose",), dict(type=_bool, default=True, help="more details in the logging file")),
    (("--print-freq", "-p"), dict(default=20, type=int, metavar="N", help="print frequency")),
    # learning
    (("--tta",), dict(type=_bool, default=True, help="perform test-time adaptation")),
    (("--use_src_stat_in_reg",), dict(type=_bool, default=True, help="use source statistics in the regularization loss")),
    (("--fix_BNS",), dict(type=_bool, default=True, help="freeze the BN statistics of the target model during the forward pass")),
    (("--running_manner",), dict(type=_bool, default=True, help="compute the target statistics in running manner")),
    (("--momentum_bns",), dict(type=float, default=0.1)),
    (("--update_only_bn_affine",), dict(action="store_true")),
    (("--compute_stat",), dict(action="store_true")),
    (("--momentum_mvg",), dict(type=float, default=0.1)),
    (("--stat_reg",), dict(type=str, default="mean_var", help="statistics regularization")),
    (("--if_tta_standard",), dict(type=str, default="tta_online")),
    (("--loss_type",), dict(type=str, default="nll", choices=["nll"])),
    (("--if_sample_tta_aug_views",), dict(type=_bool, default=True)),
    (("--if_spatial_rand_cropping",), dict(type=_bool, default=True)),
    (("--if_pred_consistency",), dict(type=_bool, default=True)),
    (("--lambda_pred_consis",), dict(type=float, default=0.1)),
    (("--lambda_feature_reg",), dict(type=int, default=1)),
    (("--n_augmented_views",), dict(type=int, default=2)),
    (("--tta_view_sample_style_list",), dict(default=["uniform_equidist"])),
    (("--stat_type",), dict(default=["spatiotemp"])),
    (("--before_norm",), dict(action="store_true")),
    (("--reduce_dim",), dict(type=_bool, default=True)),
    (("--reg_type",), dict(type=str, default="l1_loss")),
    (("--chosen_blocks",), dict(default=["layer3", "layer4"])),
    (("--moving_avg",), dict(type=_bool, default=True)),
    (("--hip_graph",), dict(type=_bool, default=True, help="(extension) replay the per-video step from captured hipGraphs")),
    (("--overlap_eval",), dict(type=_bool, default=True,
                               help="(extension) evaluate video i on a second stream beside the adaptation forward of "
                                    "video i+1 (same weights, same results)")),
    (("--prefetch_input",), dict(type=_bool, default=True,
                                 help="(extension) upload the next video on a copy stream beside the current step "
                                      "(vitta_amd/prefetch.py; the reference uploads in front of each step)")),
    (("--device_preprocess",), dict(type=_bool, default=False,
                                    help="(extension) TANet real-video pipeline: crop + PIL-BILINEAR resize + normalise on "
                                         "the GPU from the uploaded uint8 frames (bit-identical to the host PIL path)")),
    (("--wmsa_bf16",), dict(type=_bool, default=False,
                            help="(extension) Video Swin-B: window attention with bf16 MFMA operands, fp32 softmax / accumulation "
                                 "(windows up to 800 tokens in one pass; a trainable relative-position table takes the one-pass "
                                 "backward, which bins its gradient in LDS)")),
    (("--dense_bf16",), dict(type=_bool, default=False,
                             help="(extension) Video Swin-B: qkv / proj / MLP / patch-merging products with bf16 MFMA operands, "
                                  "fp32 accumulation and epilogues (vitta_gemm_nt_bf16w_f32); default: exact fp32 MFMA")),
    (("--n_gradient_steps",), dict(type=int, default=1, help="number of gradient steps per sample")),
    # input / optimiser
    (("--full_res",), dict(action="store_true")),
    (("--input_size",), dict(type=int, default=224)),
    (("--scale_size",), dict(type=int, default=256)),
    (("--batch_size",), dict(type=int, default=1)),
    (("--clip_length",), dict(type=int, default=16)),
    (("--sample_style",), dict(type=str, default="uniform-1", help="'dense-xx' or 'uniform-xx'; xx = number of temporal clips")),
    (("--test_crops",), dict(type=int, default=1, help="number of spatial crops")),
    (("--use_pretrained",), dict(action="store_true")),
    (("--input_mean",), dict(default=input_mean)),
    (("--input_std",), dict(default=input_std)),
    (("--lr",), dict(default=0.00005, type=float)),
    (("--n_epoch_adapat",), dict(default=1, type=int)),
    (("--momentum",), dict(default=0.9, type=float, metavar="M", help="momentum")),
    (("--weight-decay", "--wd"), dict(default=5e-4, type=float, metavar="W", help="weight decay")),
]


def build_parser():
    p = argparse.ArgumentParser(description="ViTTA")
    for flags, kw in _FLAGS:
        p.add_argument(*flags, **kw)
    return p


parser = build_parser()


def get_opts(argv=None):
    args = parser.parse_args(argv)
    args.evaluate_baselines = not args.tta
    args.baseline = "source"
    return args
