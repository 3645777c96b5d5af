"""Data-parallel path on the CPU: 2 ranks over gloo reproduce the reference run with batch_size=2
(SURVEY section 8e).  Exchanges under test: ONE all-reduce of the packed moments [cnt|s1|s2] before the
EMA update, ONE SUM all-reduce of the flat gradient arena before the optimizer step, the ragged tail."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, tmp, mode, device="cpu"):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    import test_host_cpu as T
    from oracle.oracle_backend import OracleBackend
    from vitta_amd import data, tta
    g = H.golden("tta3_bz2.npz")
    cfg = json.loads(str(g["config"]))
    Tn, size = cfg["T"], cfg["size"]
    model = H.build_tanet(101, Tn, 0)
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    rdir = os.path.join(tmp, f"r{rank}")
    os.makedirs(rdir, exist_ok=True)
    mp_, vp_ = H.write_stat_files(rdir, [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                  [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    args = H.tanet_args(rdir, clip_length=Tn, input_size=size, batch_size=1, spatiotemp_mean_clean_file=mp_,
                        spatiotemp_var_clean_file=vp_, lr=cfg["lr_sgd"])
    # rank r sees rows [16 r, 16 r + 16) of the reference's batch-of-two dropout masks
    masks = []
    for i in range(3):
        m = H.unpack_mask(g[f"sgd_step{i}_dropmask"], g[f"sgd_step{i}_dropmask_shape"])
        masks.append(m[rank * 2 * Tn:(rank + 1) * 2 * Tn])
    dev = torch.device(device)
    if dev.type == "cpu":
        tta.BACKEND_FACTORY = OracleBackend  # host logic over the oracle; on the GPU the product's HIP backend runs
    adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(dev), args)
    tta.BACKEND_FACTORY = None
    assert adapter.world == 2 and adapter.bucket is not None and adapter.engine.distributed
    adapter.model.module.base_model.fc = H.ReplayDropout(0.8, masks)
    tta_set = data.SyntheticVideoDataset(cfg["n_videos"], 2, Tn, size, 101, "tanet", seed0=cfg["seed0"])
    eval_set = data.SyntheticVideoDataset(cfg["n_videos"], 1, Tn, size, 101, "tanet", seed0=cfg["seed0"])
    out = {}
    steps = 3 if mode == "full" else 2
    for step in range(steps):
        vid = 2 * step + rank
        has_video = not (mode == "ragged" and step == 1 and rank == 1)  # rank 1 runs dry on the last step
        adapter.set_adapt_mode()
        x = adapter.shape_tta_input(tta_set[vid][0].unsqueeze(0).to(dev)) if has_video else None
        _, loss_reg, loss_consis = adapter.adapt_step(x, has_video)
        named = dict(adapter.model.named_parameters())
        adapter.close_hooks()
        logits = adapter.evaluate(adapter.shape_eval_input(eval_set[vid][0].unsqueeze(0).to(dev)))
        adapter.add_hooks_back()
        out[f"step{step}_loss_reg"] = float(loss_reg)
        out[f"step{step}_loss_consis"] = float(loss_consis) if loss_consis is not None else float("nan")
        out[f"step{step}_logits"] = logits.detach().cpu().numpy()
        out[f"step{step}_ema_sum"] = float(adapter.engine.ema_mean.double().sum())
        out[f"step{step}_param_sum"] = float(sum(float(p.double().sum()) for p in named.values()))
        for name in map(str, g["sampled_params"]):
            out[f"step{step}_param::{name}"] = named[name].detach().cpu().numpy()[:int(g["sample_rows"])].copy()
            out[f"step{step}_grad::{name}"] = named[name].grad.detach().cpu().numpy()[:int(g["sample_rows"])].copy()
    out["buckets_from_backward"] = int(getattr(adapter, "n_from_backward", 0))
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), **out)
    torch.distributed.destroy_process_group()


def _run(tmp_path, mode, device="cpu"):
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path), mode, device), nprocs=2, join=True)
    return [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(2)]


def test_two_ranks_equal_reference_batch_of_two(tmp_path):
    g = H.golden("tta3_bz2.npz")
    r0, r1 = _run(tmp_path, "full")
    rows = int(g["sample_rows"])
    for i in range(3):
        k = f"sgd_step{i}_"
        ref_reg, ref_con = float(g[k + "loss_reg"]), float(g[k + "loss_consis"])
        floor_reg = 4 * float(g[k + "noise_loss_reg"])
        # identical on both ranks (same all-reduced moments), equal to the reference's pooled statistics
        assert r0[f"step{i}_loss_reg"] == pytest.approx(float(r1[f"step{i}_loss_reg"]), rel=1e-6)
        assert abs(float(r0[f"step{i}_loss_reg"]) - ref_reg) <= max(1e-5 * ref_reg, floor_reg) + 1e-7
        # the consistency loss is a SUM over videos: per-rank parts add up to the reference's batch value
        total_con = float(r0[f"step{i}_loss_consis"]) + float(r1[f"step{i}_loss_consis"])
        assert abs(total_con - ref_con) <= max(1e-5 * ref_con, 4 * float(g[k + "noise_loss_consis"])) + 1e-7
        # replicas stay identical: same EMA, same weights
        assert float(r0[f"step{i}_ema_sum"]) == pytest.approx(float(r1[f"step{i}_ema_sum"]), rel=1e-9)
        assert float(r0[f"step{i}_param_sum"]) == pytest.approx(float(r1[f"step{i}_param_sum"]), rel=1e-12)
        ref_logits = g[k + "eval_logits"]
        got = np.concatenate([r0[f"step{i}_logits"], r1[f"step{i}_logits"]])
        bound = max(2e-3 * np.abs(ref_logits).max(), 4 * float(g[k + "noise_eval_logits"]))
        assert np.abs(got - ref_logits).max() <= bound
        for name in map(str, g["sampled_params"]):
            ref_g = g[k + f"grad::{name}"]
            bound = max(5e-3 * np.abs(ref_g).max(), 4 * float(g[k + f"noise_grad::{name}"])) + 1e-10
            assert np.abs(r0[f"step{i}_grad::{name}"] - ref_g).max() <= bound, (i, name)  # all-reduced SUM gradient
            np.testing.assert_array_equal(r0[f"step{i}_grad::{name}"], r1[f"step{i}_grad::{name}"])
            np.testing.assert_array_equal(r0[f"step{i}_param::{name}"], r1[f"step{i}_param::{name}"])


def test_ragged_tail_keeps_replicas_in_sync(tmp_path):
    """3 videos on 2 ranks: in the last step rank 1 has no video, contributes n = 0 to the moments and
    zeros to the gradient all-reduce, and still ends with the same EMA state and weights as rank 0."""
    r0, r1 = _run(tmp_path, "ragged")
    assert float(r0["step1_ema_sum"]) == pytest.approx(float(r1["step1_ema_sum"]), rel=1e-9)
    assert float(r0["step1_param_sum"]) == pytest.approx(float(r1["step1_param_sum"]), rel=1e-12)
    assert float(r0["step1_loss_reg"]) == pytest.approx(float(r1["step1_loss_reg"]), rel=1e-6)


def _epoch_rank_main(rank, world, port, tmp):
    """Epoch-style test_time_adapt (N4) on 2 ranks, one video per rank and step = the reference's batch of two."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    from oracle.oracle_backend import OracleBackend
    from vitta_amd import tta
    g = H.golden("epoch.npz")
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    rdir = os.path.join(tmp, f"r{rank}")
    os.makedirs(rdir, exist_ok=True)
    mp_, vp_ = H.write_stat_files(rdir, [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                  [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    args = H.tanet_args(rdir, clip_length=8, input_size=64, spatiotemp_mean_clean_file=mp_, spatiotemp_var_clean_file=vp_,
                        lr=1e-3, if_tta_standard=False, update_only_bn_affine=True, batch_size=1, batch_size_eval=2,
                        synthetic_n_videos=4, synthetic_seed=500, device="cpu")
    masks = []
    for i in range(2):  # rank r = rows [16 r, 16 r + 16) of the reference's batch-of-two dropout mask
        m = H.unpack_mask(g[f"step{i}_dropmask"], g[f"step{i}_dropmask_shape"])
        masks.append(m[rank * 16:(rank + 1) * 16])
    model = tta.SingleDeviceParallel(H.build_tanet(101, 8, 0))
    model.module.base_model.fc = H.ReplayDropout(0.8, masks)
    tta.BACKEND_FACTORY = OracleBackend
    seen, logits = [], []
    real_step, real_eval = tta.ViTTAAdapter.adapt_step, tta.ViTTAAdapter._evaluate_eager

    def step(self, *a, **k):
        out = real_step(self, *a, **k)
        seen.append((float(out[1]), float(out[2])))
        return out

    def evaluate(self, x):
        o = real_eval(self, x)
        logits.append(o.detach().numpy().copy())
        return o

    tta.ViTTAAdapter.adapt_step, tta.ViTTAAdapter._evaluate_eager = step, evaluate
    res, adapted = tta.test_time_adapt(model, torch.nn.CrossEntropyLoss(), args=args, logger=logging.getLogger("t"), writer=None)
    named = dict(adapted.named_parameters())
    out = {"top1": np.array(res), "logits": np.concatenate(logits), "loss_reg": np.array([s[0] for s in seen]),
           "loss_consis": np.array([s[1] for s in seen])}
    for name in map(str, g["sampled_params"]):
        out[f"param::{name}"] = named[name].detach().numpy()[:4].copy()
    np.savez(os.path.join(tmp, f"epoch_rank{rank}.npz"), **out)
    torch.distributed.destroy_process_group()


def test_epoch_style_on_two_ranks_equals_reference_batch_of_two(tmp_path):
    """test_time_adapt data-parallel: videos (0, 1) then (2, 3) adapted one per rank = the reference's two steps of
    batch size two; validate_brief splits the list and adds up the counts."""
    port = _free_port()
    mp.spawn(_epoch_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(os.path.join(str(tmp_path), f"epoch_rank{r}.npz")) for r in range(2)]
    g = H.golden("epoch.npz")
    for i in range(2):
        assert float(r0["loss_reg"][i]) == pytest.approx(float(r1["loss_reg"][i]), rel=1e-6)
        assert float(r0["loss_reg"][i]) == pytest.approx(float(g[f"step{i}_loss_reg"]), rel=1e-5 if i == 0 else 2e-3)
        total = float(r0["loss_consis"][i]) + float(r1["loss_consis"][i])  # a SUM over the batch's videos
        assert total == pytest.approx(float(g[f"step{i}_loss_consis"]), rel=1e-5 if i == 0 else 1e-2)
    for name in map(str, g["sampled_params"]):
        np.testing.assert_array_equal(r0[f"param::{name}"], r1[f"param::{name}"])  # replicas identical
        ref = g[f"step1_param::{name}"]
        assert np.abs(r0[f"param::{name}"] - ref).max() <= 2.5e-3 + 1e-5 * np.abs(ref).max()
    ref = g["eval_logits"]
    got = np.stack([r0["logits"][0], r1["logits"][0], r0["logits"][1], r1["logits"][1]])  # rank r evaluated videos r, 2 + r
    assert np.abs(got - ref).max() <= max(2e-3 * np.abs(ref).max(), 2 * float(g["noise_eval_logits"]))
    assert r0["top1"].tolist() == r1["top1"].tolist() == pytest.approx(g["top1"].tolist())


def _arena_rank(rank, world, port, tmp):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn as nn
    from vitta_amd import tta
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 16, 1), nn.BatchNorm2d(16), nn.Linear(16, 5))
    out = {}
    for mode in ("mono", "bucketed"):
        arena = tta.FlatArena(list(net.parameters()))
        g = torch.Generator().manual_seed(100 + rank)
        arena.grad.copy_(torch.randn(arena.grad.numel(), generator=g))
        if mode == "mono":
            arena.all_reduce()
        else:
            groups = [list(net[i].parameters()) for i in range(5)]
            spans = [arena.span(ps) for ps in groups]
            assert spans[0][0] == 0 and spans[-1][1] == arena.grad.numel() and all(spans[i][1] == spans[i + 1][0] for i in range(4))
            for lo, hi in reversed(spans):  # the order a backward would finish them in
                arena.reduce_range(lo, hi, async_op=True)
            assert len(arena._pending) == 5
            arena.wait_pending()
        out[mode] = arena.grad.clone().numpy()
    np.savez(os.path.join(tmp, f"arena{rank}.npz"), **out)
    torch.distributed.destroy_process_group()


def test_bucketed_gradient_exchange_equals_the_monolithic_all_reduce(tmp_path):
    """FlatArena.reduce_range over a partition of the arena (asynchronous, in reverse layer order: what the trunk's
    backward launches bucket by bucket) leaves exactly what ONE all-reduce of the whole arena leaves, on both ranks."""
    port = _free_port()
    mp.spawn(_arena_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), f"arena{i}.npz")) for i in range(2)]
    np.testing.assert_array_equal(r[0]["mono"], r[1]["mono"])
    np.testing.assert_array_equal(r[0]["bucketed"], r[0]["mono"])
    np.testing.assert_array_equal(r[1]["bucketed"], r[1]["mono"])


def test_swin_bucket_plan_partitions_the_arena_at_block_boundaries():
    """Video Swin-B under SGD over all parameters: the gradient arena (351 MB at full size) is cut at transformer-block /
    patch-merging boundaries; a bucket leaves when the signal of its LOWEST unit fires -- for a block that is not the first of
    its stage that is the signal of the block before it (its norm1 is differentiated in that block's closing pass).  Built
    from the model's structure alone, so every rank holds the same plan before any step has run."""
    from vitta_amd import scripts, tta
    from vitta_amd.swin import PatchMerging, SwinTransformerBlock3D
    model = tta.SingleDeviceParallel(H.build_swin(11, 0))
    args = scripts.swin_ucf101_args([])
    args.update_only_bn_affine = False

    class A(tta.ViTTAAdapter):  # the exchange plumbing alone: no hooks, no statistics files
        def __init__(self, model, args):
            self.model, self.args = model, args
            self.arena = tta.FlatArena(list(model.parameters()))
            self.bucket, self.grad_buckets, self._bucket_plan = self.arena, 4, None
            self._armed = self._armed_runner = None
            self._launched, self.n_from_backward = set(), 0

    ad = A(model, args)
    units = ad._bucket_units()
    kinds = [type(m) for m, _ in units]
    assert kinds.count(SwinTransformerBlock3D) == 24 and kinds.count(PatchMerging) == 3
    first_of_stage = {0, 3, 6, 25}  # blocks 2, 2, 18, 2 with a PatchMerging after each of the first three stages
    for i, (m, rel) in enumerate(units):
        assert rel == (i if (i in first_of_stage or isinstance(m, PatchMerging)) else i - 1), (i, rel)
    n = ad.arena.grad.numel()
    plan = None
    for nb in (8, 12, 16, 4):  # (8, 12 and 16 cuts used to give two buckets one signal: the armed table dropped one of them)
        ad.grad_buckets, ad._bucket_plan = nb, None
        plan = ad.bucket_plan()
        pieces = sorted([(lo, hi) for _, lo, hi in plan["buckets"]] + [r for r in plan["rest"] if r[1] > r[0]])
        assert pieces[0][0] == 0 and pieces[-1][1] == n and all(pieces[i][1] == pieces[i + 1][0] for i in range(len(pieces) - 1))
        assert 2 <= len(plan["buckets"]) <= nb + 1
        los = [lo for _, lo, _ in plan["buckets"]]
        sigs = [sig for sig, _, _ in plan["buckets"]]
        # launch order = backward order, and NO two buckets share a release signal (the armed table is keyed by it)
        assert los == sorted(los, reverse=True) and all(sigs[i] > sigs[i + 1] for i in range(len(sigs) - 1)), (nb, sigs)
        for sig, lo, hi in plan["buckets"]:  # the signal's unit starts at or before the bucket's first parameter
            assert ad.arena.span(list(units[sig][0].parameters()))[0] <= lo
    assert 2 <= len(plan["buckets"]) <= 5
    assert ad.bucket_plan() is plan
    # the TANet plan comes from structure too: a rank that never ran the trunk holds it
    t = A(tta.SingleDeviceParallel(H.build_tanet(11, 8, 0)), H.tanet_args("/tmp", update_only_bn_affine=False))
    tp = t.bucket_plan()
    assert tp is not None and len(tp["buckets"]) == 4 and [m for m, _ in t._bucket_units()] == tp["blocks"]


def _swin_rank(rank, world, port, tmp, buckets):
    sys.path.insert(0, HERE)
    os.environ.update(VITTA_GRAD_BUCKETS=str(buckets), VITTA_TEST_SWIN_SGD_ALL="1")
    import swin_dp_worker as W
    W.run(rank, world, port, tmp, dev_name="cpu", steps=1)


def test_swin_bucketed_gradient_exchange_equals_the_monolithic_all_reduce(tmp_path):
    """Two ranks over gloo, Video Swin-B (64^2 clips), SGD over all parameters, one full adaptation step through
    ViTTAAdapter: with the arena exchanged in four buckets both ranks hold bit-for-bit the reduced gradients, EMA state and
    weights that ONE all-reduce leaves."""
    res = {}
    for nb in (1, 4):
        sub = tmp_path / f"b{nb}"
        sub.mkdir()
        mp.spawn(_swin_rank, args=(2, _free_port(), str(sub), nb), nprocs=2, join=True)
        res[nb] = [np.load(os.path.join(str(sub), f"w2r{r}.npz")) for r in range(2)]
    assert int(res[1][0]["n_buckets"]) == 0 and int(res[4][0]["n_buckets"]) >= 2
    for r in range(2):
        np.testing.assert_array_equal(res[4][r]["step0_grad"], res[1][r]["step0_grad"])
        np.testing.assert_array_equal(res[4][r]["step0_ema"], res[1][r]["step0_ema"])
        assert float(res[4][r]["step0_param_sum"]) == float(res[1][r]["step0_param_sum"])
    np.testing.assert_array_equal(res[4][0]["step0_grad"], res[4][1]["step0_grad"])


def _rank4_main(rank, world, port, tmp):
    """One of FOUR ranks: six videos -> step 0 gives every rank a video (0..3), step 1 only ranks 0 and 1 (4, 5): ranks 2 and 3 run dry."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle_backend import OracleBackend
    from vitta_amd import tta
    adapter, tta_set = _ws4_adapter(os.path.join(tmp, f"r{rank}"), 1, OracleBackend, tta)
    assert adapter.world == 4 and adapter.bucket is not None and adapter.engine.distributed
    out = {}
    for step in range(2):
        vid = 4 * step + rank
        has_video = vid < 6
        adapter.set_adapt_mode()
        x = adapter.shape_tta_input(tta_set[vid][0].unsqueeze(0)) if has_video else None
        _, loss_reg, loss_consis = adapter.adapt_step(x, has_video)
        named = dict(adapter.model.named_parameters())
        out[f"step{step}_loss_reg"] = float(loss_reg)
        out[f"step{step}_loss_consis"] = float(loss_consis) if loss_consis is not None else 0.0
        out[f"step{step}_ema"] = adapter.engine.ema_mean.detach().cpu().numpy().copy()
        out[f"step{step}_params"] = np.concatenate([p.detach().cpu().numpy().ravel() for _, p in sorted(named.items()) if p.requires_grad])
        out[f"step{step}_grads"] = np.concatenate([p.grad.detach().cpu().numpy().ravel() for _, p in sorted(named.items()) if p.requires_grad])
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), **out)
    torch.distributed.destroy_process_group()


def _ws4_adapter(rdir, batch_size, backend, tta):
    from vitta_amd import data
    g = H.golden("tta3_bz2.npz")
    cfg = json.loads(str(g["config"]))
    Tn, size = cfg["T"], cfg["size"]
    os.makedirs(rdir, exist_ok=True)
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    mp_, vp_ = H.write_stat_files(rdir, [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                  [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    args = H.tanet_args(rdir, clip_length=Tn, input_size=size, batch_size=batch_size, spatiotemp_mean_clean_file=mp_,
                        spatiotemp_var_clean_file=vp_, update_only_bn_affine=True, lr=1e-5)
    model = H.build_tanet(101, Tn, 0)
    model.base_model.fc = torch.nn.Identity()  # no dropout: the four ranks and the one process see the same forward
    tta.BACKEND_FACTORY = backend
    try:
        adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model), args)
    finally:
        tta.BACKEND_FACTORY = None
    return adapter, data.SyntheticVideoDataset(6, 2, Tn, size, 101, "tanet", seed0=cfg["seed0"])


def test_four_ranks_with_two_dry_ranks_in_the_tail_equal_one_process_batches(tmp_path):
    """World size 4 (round 6): six videos on four ranks.  Step 0: every rank adapts to one video; step 1: ranks 0 / 1
    hold videos 4 / 5, ranks 2 and 3 run DRY -- they contribute n = 0 to the packed-moments all-reduce and zeros to the gradient
    all-reduce.  All four replicas end every step with bit-identical EMA state, gradients and parameters, and those equal ONE process
    adapting the batch of four and then the batch of two (the reference's semantics with batch_size = R, pinned against the reference
    at R = 2 by test_two_ranks_equal_reference_batch_of_two): norm_stats_utils.py:193,202-204,242-243 pool every clip of the batch,
    pred_consistency_utils.py:8,28-30 sums over it."""
    from oracle.oracle_backend import OracleBackend
    from vitta_amd import tta
    mp.spawn(_rank4_main, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    r = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(4)]
    for step in range(2):
        for k in range(1, 4):
            np.testing.assert_array_equal(r[0][f"step{step}_ema"], r[k][f"step{step}_ema"])
            np.testing.assert_array_equal(r[0][f"step{step}_grads"], r[k][f"step{step}_grads"])
            np.testing.assert_array_equal(r[0][f"step{step}_params"], r[k][f"step{step}_params"])
            assert float(r[0][f"step{step}_loss_reg"]) == pytest.approx(float(r[k][f"step{step}_loss_reg"]), rel=1e-7)
    assert float(r[2]["step1_loss_consis"]) == 0.0 and float(r[3]["step1_loss_consis"]) == 0.0
    # one process: batch of four, then batch of two
    torch.set_num_threads(8)
    adapter, tta_set = _ws4_adapter(str(tmp_path / "one"), 4, OracleBackend, tta)
    for step, vids in enumerate(([0, 1, 2, 3], [4, 5])):
        adapter.set_adapt_mode()
        x = torch.stack([tta_set[v][0] for v in vids])
        _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x))
        named = dict(adapter.model.named_parameters())
        grads = np.concatenate([p.grad.detach().numpy().ravel() for _, p in sorted(named.items()) if p.requires_grad])
        params = np.concatenate([p.detach().numpy().ravel() for _, p in sorted(named.items()) if p.requires_grad])
        assert float(r[0][f"step{step}_loss_reg"]) == pytest.approx(float(loss_reg), rel=2e-5 if step == 0 else 1e-4)
        assert sum(float(r[k][f"step{step}_loss_consis"]) for k in range(4)) == pytest.approx(float(loss_consis), rel=1e-4 if step == 0 else 2e-3, abs=1e-7)
        # step 1 runs on weights that Adam's FIRST update moved by lr * g / (|g| + eps): an element whose gradient sits at round-off
        # takes either sign, so the two runs' weights differ by up to 2 lr = 2e-5 there (hence the small lr) and the second step's statistics follow
        np.testing.assert_allclose(r[0][f"step{step}_ema"], adapter.engine.ema_mean.detach().numpy(), rtol=1e-4, atol=1e-6 if step == 0 else 2e-5)
        ga, gb = r[0][f"step{step}_grads"].astype(np.float64), grads.astype(np.float64)
        cos = float(np.dot(ga, gb) / (np.linalg.norm(ga) * np.linalg.norm(gb)))
        assert cos >= 0.9999, (step, cos)
        if step == 0:  # (Adam's first update is lr * sign-like: the parameters agree wherever the gradients do)
            assert np.abs(r[0]["step0_params"] - params).max() <= 2.5e-5
