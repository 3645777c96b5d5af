"""Rank body of the Swin data-parallel GPU test (tests/test_gpu_entrypoints.py): `world` ranks share cuda:0 over gloo,
rank r adapts to video r; world == 1: one process with both videos as a batch of two."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def run(rank, world, port, tmp, dev_name="cuda:0", steps=3):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import helpers as H
    from vitta_amd import data, scripts, tta
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device(dev_name)
    if dev.type == "cpu":
        torch.set_num_threads(4)
    g = H.golden("tta3_swin.npz")
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    rdir = os.path.join(tmp, f"w{world}r{rank}")
    os.makedirs(rdir, exist_ok=True)
    mp_, vp_ = H.write_stat_files(rdir, [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                  [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    model = H.build_swin(101, 0, drop_path_rate=0.0)
    model.cls_head.dropout = None  # no stochastic layer: ranks and the batched run see the same arithmetic
    args = scripts.swin_ucf101_args([])
    args.datatype, args.input_size, args.scale_size, args.workers, args.verbose = "synthetic", size, size, 0, False
    args.result_dir, args.num_classes, args.batch_size = rdir, 101, 1
    args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp_, vp_
    sgd_all = os.environ.get("VITTA_TEST_SWIN_SGD_ALL", "0") == "1"  # the reference's default optimizer: 351 MB gradient arena
    args.update_only_bn_affine, args.lr = (False, cfg["lr_sgd"]) if sgd_all else (True, cfg["lr_adam"])
    if dev_name != "cuda:0":
        from oracle.oracle_backend import OracleBackend
        tta.BACKEND_FACTORY = OracleBackend
    adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(dev), args)
    tta.BACKEND_FACTORY = None
    tta_set = data.SyntheticVideoDataset(6, 2, T, size, 101, "swin", seed0=cfg["seed0"])
    out = {}
    for step in range(steps):
        vids = [2 * step + rank] if world > 1 else [2 * step, 2 * step + 1]
        x = torch.stack([tta_set[v][0] for v in vids]).to(dev)
        adapter.set_adapt_mode()
        _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
        out[f"step{step}_loss_reg"] = float(loss_reg)
        out[f"step{step}_loss_consis"] = float(loss_consis)
        out[f"step{step}_ema"] = adapter.engine.ema_mean.detach().cpu().numpy().copy()
        out[f"step{step}_param_sum"] = float(sum(float(p.double().sum()) for p in adapter.model.parameters()))
        out[f"step{step}_grad"] = adapter.arena.grad.detach().cpu().numpy()[::997].copy()  # a sample of the reduced arena
    plan = adapter.bucket_plan()
    out["n_buckets"] = 0 if plan is None else len(plan["buckets"])
    out["buckets_from_backward"] = int(adapter.n_from_backward)
    np.savez(os.path.join(tmp, f"w{world}r{rank}.npz"), **out)
    if world > 1:
        torch.distributed.destroy_process_group()
