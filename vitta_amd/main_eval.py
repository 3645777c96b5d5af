"""`eval(args, model=None) -> (epoch_result_list, model)` -- the entry point every script calls.

Interface mirror of corpus/main_eval.py:30-232 for the branches on the ViTTA path: TTA
(`tta_standard`), source-statistics production (`compute_stat == 'mean_var'`) and source-only
evaluation (`tta=False`, baseline 'source').  The competing baselines (tent/norm/shot/dua/t3a,
main_eval.py:126-226) are out of scope and raise NotImplementedError.

Differences: device-agnostic (`args.device` or cuda:LOCAL_RANK when a GPU is visible, else CPU --
BASELINE config 0 runs source-only evaluation on the host); `SingleDeviceParallel` instead of
nn.DataParallel (same `module.` key prefix, one process per GPU); source-only returns the
(list, model) tuple the scripts can actually index (the reference's scripts mis-handle it,
sourceonly_tanet_ucf101_corr.py:40-44).
"""
import os
import os.path as osp
import time

import torch

from .tta import NUM_CLASSES, SingleDeviceParallel, _loader, compute_statistics, test_time_adapt, get_dataset_tanet, \
    get_dataset_videoswin, get_model, tta_standard, validate
from .utils_ import make_dir, model_analysis, path_logger


def pick_device(args):
    dev = getattr(args, "device", None)
    if dev is not None:
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None:  # "cuda": this rank's GPU
            dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        return dev
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cpu")


def init_distributed(device):
    """One process per GPU under torchrun (RANK / WORLD_SIZE in the environment): join the job's process group -- RCCL
    ('nccl') for GPU ranks, gloo for host ranks -- so that tta_standard / test_time_adapt shard the videos and run their
    two exchanges.  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    d = torch.distributed
    if world > 1 and d.is_available() and not d.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if device.type == "cuda":
            d.init_process_group("nccl", device_id=device)
        else:
            d.init_process_group("gloo")
    if d.is_available() and d.is_initialized():
        return d.get_rank(), d.get_world_size()
    return 0, 1


def load_checkpoint_into(model, args, logger, device):
    """Both checkpoint layouts of the reference (main_eval.py:55-65): keys with a `module.` prefix
    (TANet) are loaded into the wrapped model, keys without (Swin) into the bare model."""
    checkpoint = torch.load(args.model_path, map_location="cpu")
    logger.debug(f"Loading {args.model_path}")
    if args.arch == "tanet" and "epoch" in checkpoint:
        print("model epoch {} best prec@1: {}".format(checkpoint["epoch"], checkpoint.get("best_prec1")))
    state = checkpoint["state_dict"]
    if "module." in next(iter(state.keys())):
        model = SingleDeviceParallel(model)
        model.load_state_dict(state)
    else:
        model.load_state_dict(state)
        model = SingleDeviceParallel(model)
    return model.to(device)


def eval(args=None, model=None):
    log_time = time.strftime("%Y%m%d_%H%M%S")
    make_dir(args.result_dir)
    device = pick_device(args)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    rank, _ = init_distributed(device)
    # every rank keeps its own log file (rank 0 under the reference's name); the *_all_result line is rank 0's
    logger = path_logger(args.result_dir, log_time if rank == 0 else f"{log_time}_rank{rank}")
    if args.verbose:
        for arg in dir(args):
            if arg[0] != "_":
                logger.debug(f"{arg} {getattr(args, arg)}")
    args.num_classes = num_classes = NUM_CLASSES[args.dataset]
    if device.type == "cuda":
        if args.arch == "videoswintransformer":
            from . import ops
            ops.WMSA_BF16 = bool(getattr(args, "wmsa_bf16", False))
            ops.DENSE_BF16 = bool(getattr(args, "dense_bf16", False))

    if model is None:
        model = load_checkpoint_into(get_model(args, num_classes, logger), args, logger, device)
    if args.verbose:
        model_analysis(model, logger)
    args.crop_size = args.input_size

    if args.loss_type != "nll":
        raise ValueError("Unknown loss type")
    criterion = torch.nn.CrossEntropyLoss().to(device)

    epoch_result_list = None
    if args.tta:
        if args.compute_stat == "mean_var":
            compute_statistics(model, args=args, log_time=log_time)
        elif args.compute_stat is False:
            if args.if_tta_standard:
                epoch_result_list = tta_standard(model, criterion, args=args, logger=logger, writer=None)
                model = None
            else:  # epoch-style: returns the adapted model (main_eval.py:96-98)
                epoch_result_list, model = test_time_adapt(model, criterion, args=args, logger=logger, writer=None)
        else:
            raise NotImplementedError(f"compute_stat={args.compute_stat!r} is outside the ViTTA path")
    elif args.evaluate_baselines:
        if args.baseline != "source":
            raise NotImplementedError(f"baseline {args.baseline!r} is not part of the ViTTA path")
        if args.arch == "tanet":
            dataset = get_dataset_tanet(args, split="val", dataset_type="eval")
        elif args.arch == "videoswintransformer":
            dataset = get_dataset_videoswin(args, split="val", dataset_type="eval")
        else:
            raise NotImplementedError(f"Incorrect model type {args.arch}")
        model.eval()
        top1_acc = validate(_loader(dataset, args), model, criterion, 0, epoch=0, args=args, logger=logger)
        epoch_result_list = [top1_acc]
    logger.handlers.clear()
    return epoch_result_list, model
