"""Meters, EMA, accuracy and logging helpers of the TTA driver.

Interface mirror of the pieces of utils/utils_.py the hot path touches:
    AverageMeter :171-187, AverageMeterTensor :190-202, MovingAverageTensor :204-211,
    accuracy :224-237, make_dir :19-21, path_logger :92-110, model_analysis :113-121,
    get_writer_to_all_result :252-267
Differences: device-agnostic (the reference hard-codes .cuda()), otherwise same names, arguments,
file names and line formats so downstream result parsers keep working.
"""
import logging
import os
import os.path as osp
import time

import numpy as np
import torch


def make_dir(dir_):
    os.makedirs(dir_, exist_ok=True)


class AverageMeter(object):
    """Running average of python scalars."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class AverageMeterTensor(object):
    """Running average of tensors; only the newest sample keeps its graph."""

    def __init__(self, device=None):
        self.device = device
        self.reset()

    def reset(self):
        z = torch.tensor(0.0, device=self.device)
        self.val, self.avg, self.sum, self.count = z, z, z, 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum.detach().to(val.device) + val * n
        self.count += n
        self.avg = self.sum / self.count


class MovingAverageTensor(object):
    """avg <- m*val + (1-m)*avg.detach(); avg0 is the scalar 0 (no bias correction)."""

    def __init__(self, momentum=0.1, device=None):
        self.momentum = momentum
        self.device = device
        self.reset()

    def reset(self):
        self.avg = torch.tensor(0.0, device=self.device)

    def update(self, val):
        self.avg = self.momentum * val + (1.0 - self.momentum) * self.avg.detach().to(val.device)


def accuracy(output, target, topk=(1,)):
    """precision@k in percent for every k of topk."""
    maxk = min(max(topk), output.size(1))
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand(maxk, batch_size))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0).mul_(100.0 / batch_size) for k in topk]


def path_logger(result_dir, log_time):
    """Logger 'basic' writing to the console and to <result_dir>/<log_time>."""
    logger = logging.getLogger("basic")
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(asctime)s - %(levelno)s - %(filename)s - %(funcName)s - %(message)s")
    for handler in (logging.StreamHandler(), logging.FileHandler(osp.join(result_dir, f"{log_time}"), mode="w")):
        handler.setLevel(logging.DEBUG)
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    return logger


def model_analysis(model, logger, print_structure=False):
    if print_structure:
        print("Model Structure")
        print(model)
    params = sum(int(np.prod(p.size())) for p in model.parameters() if p.requires_grad)
    logger.debug("#################################################")
    logger.debug(f"Number of trainable parameters: {params}")
    logger.debug("#################################################")


def get_writer_to_all_result(args, custom_path=None):
    log_time = time.strftime("%Y%m%d_%H%M%S")
    if custom_path is None:
        f_write = open(osp.join(args.result_dir, f"{log_time}_all_result"), "w+")
    else:
        f_write = open(osp.join(custom_path, f"{args.baseline}_{log_time}_all_result"), "w+")
    for arg in dir(args):
        if arg[0] != "_":
            f_write.write(f"{arg} {getattr(args, arg)}\n")
    f_write.write("#############################\n" * 2)
    f_write.write("\n\n")
    return f_write
