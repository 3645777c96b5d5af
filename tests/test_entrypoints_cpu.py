"""The reference's entry points, end to end on the CPU with synthetic videos: source-only evaluation
(BASELINE config 0: no GPU), eval() -> tta_standard (HIP launches replaced by the oracle backend),
eval() -> compute_statistics, and the script loop over corruptions."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import helpers as H
from oracle.oracle_backend import OracleBackend
from vitta_amd import scripts, tta


def _checkpoint(tmp_path, T=8):
    model = H.build_tanet(101, T, 0)
    wrapped = tta.SingleDeviceParallel(model)
    path = os.path.join(str(tmp_path), "tanet_synth.pth.tar")
    torch.save({"state_dict": wrapped.state_dict(), "epoch": 7, "best_prec1": 12.5}, path)
    return path, model


def _args(tmp_path, **over):
    a = scripts.tanet_ucf101_args([])
    a.datatype, a.clip_length, a.input_size, a.workers, a.device = "synthetic", 8, 64, 0, "cpu"
    a.synthetic_n_videos = 4
    a.val_vid_list = os.path.join(str(tmp_path), "lists", "{}.txt")
    a.result_dir = os.path.join(str(tmp_path), "results", "{}_{}", "tta_{}")
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_source_only_over_corruptions_on_cpu(tmp_path):
    ckpt, _ = _checkpoint(tmp_path)
    args = scripts.source_only(_args(tmp_path, model_path=ckpt))
    args.batch_size = 2
    res = scripts.run_over_corruptions(args, ["gauss_shuffled", "pepper_shuffled"])
    assert len(res) == 2 and all(len(r) == 1 and 0.0 <= r[0] <= 100.0 for r in res)
    assert "{}" in args.val_vid_list and "{}" in args.result_dir  # templates survive the loop (reference bug fixed)
    d0 = os.path.join(str(tmp_path), "results", "tanet_ucf101", "tta_gauss_shuffled")
    d1 = os.path.join(str(tmp_path), "results", "tanet_ucf101", "tta_pepper_shuffled")
    assert os.path.isdir(d0) and os.path.isdir(d1)
    all_result = glob.glob(os.path.join(d0, "*_all_result"))
    assert len(all_result) == 1
    lines = [l for l in open(all_result[0]).read().split("#############################\n")[-1].split("\n") if l.strip()]
    assert len(lines) == 2 and all(float(l) >= 0 for l in lines)
    log = [p for p in glob.glob(os.path.join(d0, "*")) if not p.endswith("_all_result")][0]
    text = open(log).read()
    assert "Testing Results: Prec@1" in text and "Test: [0/2]" in text


def test_eval_tta_online_on_cpu(tmp_path, monkeypatch):
    ckpt, model = _checkpoint(tmp_path)
    bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    mp, vp = H.write_stat_files(str(tmp_path), [np.zeros(b.num_features, np.float32) for b in bn2d],
                                [np.ones(b.num_features, np.float32) for b in bn2d])
    args = _args(tmp_path, model_path=ckpt, spatiotemp_mean_clean_file=mp, spatiotemp_var_clean_file=vp, verbose=True)
    args.val_vid_list, args.result_dir = "unused", os.path.join(str(tmp_path), "tta_run")
    monkeypatch.setattr(tta, "BACKEND_FACTORY", OracleBackend)
    from corpus.main_eval import eval as run_eval
    res, returned_model = run_eval(args=args)
    assert returned_model is None and len(res) == 1 and 0.0 <= res[0] <= 100.0
    log = [p for p in glob.glob(os.path.join(args.result_dir, "*"))][0]
    text = open(log).read()
    for i in range(4):
        assert f"TTA Epoch1: [{i}/4]" in text
    assert "Loss reg" in text and "Loss consis" in text and "Prec@1" in text


def test_eval_compute_statistics_on_cpu(tmp_path, monkeypatch):
    ckpt, model = _checkpoint(tmp_path)
    args = scripts.compute_stats(_args(tmp_path, model_path=ckpt))
    args.batch_size = 2
    args.val_vid_list, args.result_dir = "unused", os.path.join(str(tmp_path), "stats_run")
    monkeypatch.setattr(tta, "BACKEND_FACTORY", OracleBackend)
    from corpus.main_eval import eval as run_eval
    res, _ = run_eval(args=args)
    assert res is None
    mean_file = glob.glob(os.path.join(args.result_dir, "list_spatiotemp_mean_*.npy"))
    var_file = glob.glob(os.path.join(args.result_dir, "list_spatiotemp_var_*.npy"))
    assert len(mean_file) == 1 and len(var_file) == 1
    means = np.load(mean_file[0], allow_pickle=True)
    vars_ = np.load(var_file[0], allow_pickle=True)
    assert means.dtype == object and len(means) == 53 and len(vars_) == 53
    chans = [m.num_features for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    assert [len(m) for m in means] == chans and all(np.all(v >= 0) for v in vars_)
    # the file is directly consumable by the TTA path
    args2 = _args(tmp_path, model_path=ckpt, spatiotemp_mean_clean_file=mean_file[0], spatiotemp_var_clean_file=var_file[0])
    chosen = tta.candidate_layers_for(args2, tta.SingleDeviceParallel(model))
    m2, v2 = tta.load_source_statistics(args2, chosen)
    assert len(m2) == 85 and sum(x is None for x in m2) == 32
