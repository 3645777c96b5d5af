"""bench.py -- videos/sec of the ViTTA online TTA step on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Started without a launcher (`WORLD_SIZE` unset) and with --gpus N > 1 it starts the N ranks itself: it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` and passes
rank 0's JSON line through; `n_gpus` is the world size the ranks gathered, `ranks` lists every rank's device, and `dp_graph`
says which capture form the data-parallel step replayed ("one" graph with the RCCL all-reduces captured | "segments" with
the exchanges launched eagerly between three graphs | "eager").

Workload (BASELINE.json configs[1]): TANet-R50 (random init of the real architecture, BN statistics
calibrated), UCF101 head (101 classes), one video per GPU and step = 2 temporally augmented views x 8
frames x 3 x 224 x 224 fp32 synthetic N(0,1) clips resident in HBM, statistics alignment (l1_loss,
momentum 0.1) on the 29 BatchNorm2d outputs of layer3/layer4 + prediction consistency, Adam on the BN
affine parameters (--update_only_bn_affine).  A "step" is the reference's per-video iteration
(corpus/basics.py:516-738): adaptation forward + losses + backward + optimizer step, then the
evaluation forward of the same video (centre view).  `value` = videos of all ranks / max-over-ranks
wall time of the K timed steps (barrier + synchronize on both sides).

`ms_per_step`: the K timed steps form one BLOCK (barrier + synchronize on both sides, max over ranks); the block is
repeated until >= 1 s has been timed and the MEDIAN block is reported (`blocks` = how many, `ms_per_step_first_block` =
the contract's single block).

Extra objects on the JSON line:
  roofline      the dominant kernels of the timed step, the convolution family behind vitta_conv_f32 (every bottleneck
                convolution of the trunk, forward, data gradient and evaluation forward), bound "mfma": achieved =
                algorithmic flops (2 x positions x C x K x taps) of the step's convolution launches / the sum of their
                durations, each duration from a hipEvent pair attached to that launch's own dispatch
                (vitta_conv_timed_f32 -> hipExtLaunchKernelGGL; a kernel inside a replayed hipGraph cannot carry
                events, so the timed steps are repeated eagerly with the same kernels for this).  The family issues two
                instructions: conv_b3.hip v_mfma_f32_32x32x16_bf16 on split-bf16 operands, SIX products per fp32-grade
                multiply-add (peak 2500 / 6 = 416.7 TFLOP/s of algorithmic flops), the stride-2 gathers conv_sk / conv_pw
                v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s).  `peak` = the flop-weighted peak of the step's launches, `frac` =
                (sum over launches of flops / that launch's peak) / (sum of durations); `frac_of_fp32_matrix_peak` =
                achieved / 157.3, the yardstick of the exact-fp32 rounds.
  swin, swin_c5_bf16 (, swin_sgd_all, swin_c5_bf16_sgd_all)   the same per-video iteration on Video Swin-B: BASELINE config 3's shape (2 views x 16 frames x
                224^2, window (8,7,7), exact-fp32 kernels) and config 5's (4 views x 32 frames x 224^2, window (16,7,7), the
                bf16-operand attention + dense kernels), each with the achieved TFLOP/s of its dense (gemm.hip) and window
                attention launches against the matrix peak of the instruction they issue (157.3 / 2500).
                roofline.moments: the north-star statistics kernel, moments_nchw_partial_kernel (one launch over all
                29 hooked layers; in the TANet step its work rides in the convolution epilogues, so it is timed
                stand-alone): `one_video` = 178 MB (4 B x 44 556 288 hooked elements; fits the Infinity
                Cache), `streaming` = 2.85 GB that cannot be cache resident, against 8 TB/s.
  sgd_all       the same iteration under the reference's DEFAULT optimizer (SGD over all parameters,
                corpus/basics.py:547-560), timed in the same run after the headline configuration (single-GPU runs).
  ranks         what every rank saw: torch.distributed world size and its device (for the driver to verify N ranks).
  cpu_baseline  the CPU restatement of the reference path (oracle/: stock PyTorch CPU ops in the
                reference's op order), same workload, a few steps on the host cores of this box.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOOKED_ELEMENTS_PER_VIDEO = 44556288  # 29 BN2d outputs of layer3/4 at 2x8x224^2
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
VALU_LANE_OPS_PER_S = 78.6e12  # 256 CUs x 4 SIMD-32 x 2.4 GHz (one lane-operation per lane and cycle = the 157.3 TFLOP/s fp32 FMA rate / 2)
MFMA_F32_PEAK_TF = 157.3  # same guide: fp32 matrix peak (v_mfma_f32_32x32x2_f32 / 16x16x4), 155 TF measured
MFMA_BF16_PEAK_TF = 2500.0  # same guide: bf16 dense matrix peak (v_mfma_f32_32x32x16_bf16), 2495 TF measured
B3_PRODUCTS = 6             # conv_b3.hip: bf16 MFMA products per fp32-grade multiply-add


_T0 = time.time()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--optimizer", default="adam_affine", choices=["adam_affine", "sgd_all"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=24)
    p.add_argument("--no-streaming", action="store_true")
    p.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (PCIe-inclusive) repetition of the timed steps")
    p.add_argument("--no-sgd-all", action="store_true", help="skip the second (SGD over all parameters) timing")
    p.add_argument("--no-swin", action="store_true", help="skip the Video Swin-B legs (configs 3 and 5)")
    p.add_argument("--no-exact-fp32", action="store_true", help="skip the exact-fp32 convolution leg (VITTA_CONV_ARITH=f32)")
    p.add_argument("--no-forced-exchange-leg", action="store_true", help="skip the one-rank RCCL leg (child process)")
    p.add_argument("--min-seconds", type=float, default=1.0, help="repeat the K-step block until this much has been timed")
    p.add_argument("--timed-only", action="store_true",
                   help="profiling aid: stop after the timed region (no eager repeat / adapt-only / streaming legs), so "
                        "the tail of a rocprofv3 trace is the shipped hipGraph replay and nothing else")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo: rehearse the multi-rank code path with all ranks sharing one GPU (not a measurement)")
    p.add_argument("--sequential", action="store_true",
                   help="adapt(i); eval(i) back to back on one stream instead of eval(i-1) beside adapt(i)")
    p.add_argument("--segmented-graph", action="store_true",
                   help="single GPU: use the data-parallel capture (3 graph segments, exchanges outside) anyway")
    p.add_argument("--dense-bf16", action="store_true",
                   help="--arch swin: qkv / proj / MLP products with bf16 MFMA operands (gemm.hip); dtype is reported")
    p.add_argument("--wmsa-bf16", action="store_true",
                   help="--arch swin: window attention with bf16 MFMA operands (BASELINE config 5's recipe); dtype is reported")
    p.add_argument("--graph-collectives", action="store_true",
                   help="data-parallel step as ONE hipGraph with the two RCCL all-reduces captured inside it")
    p.add_argument("--force-exchanges", action="store_true",
                   help="single process: form a ONE-rank RCCL group and run the data-parallel step (segmented graphs, both "
                        "all-reduces) anyway -- exercises the RCCL calls on a one-GPU box")
    p.add_argument("--miopen-find", action="store_true", help="cudnn.benchmark=True (MIOpen find mode) like main_eval.py:77")
    p.add_argument("--rendezvous-only", action="store_true",
                   help="start / join the ranks, gather what every rank sees, print the line's `n_gpus` / `ranks` part and stop "
                        "(needs no GPU with --dist-backend gloo: the launch logic's own test)")
    p.add_argument("--size", type=int, default=224)
    p.add_argument("--clip-length", type=int, default=None, help="frames per view (default 8 TANet / 16 Swin)")
    p.add_argument("--arch", default="tanet", choices=["tanet", "swin"],
                   help="tanet: BASELINE.json's metric (default); swin: the same iteration on Video Swin-B "
                        "(BASELINE config 2 shape: 2 views x 16 frames x 224^2, LN-affine Adam)")
    opt = p.parse_args()
    if opt.clip_length is None:
        opt.clip_length = 8 if opt.arch == "tanet" else 16
    return opt


def make_args(tmp, size, clip_length, optimizer, device, n_videos):
    from vitta_amd.opts import get_opts
    a = get_opts([])
    a.arch, a.dataset, a.datatype, a.num_classes = "tanet", "ucf101", "synthetic", 101
    a.clip_length, a.input_size, a.batch_size, a.workers = clip_length, size, 1, 0
    a.n_augmented_views, a.verbose, a.result_dir, a.gpus = 2, False, tmp, [0]
    a.update_only_bn_affine = optimizer == "adam_affine"
    a.lr = 5e-5
    a.synthetic_n_videos, a.synthetic_device = n_videos, device
    return a


def build_model_and_stats(tmp, size, T, device):
    """Seeded TANet-R50 with BN calibrated on a 224^2 batch, source statistics = moments of a second
    calibration pass (seed 1000) on the 53 BatchNorm2d outputs (what compute_statistics would write)."""
    from vitta_amd import synthetic as S
    from vitta_amd.norm_stats import ComputeNormStatsHook
    from vitta_amd.tanet import TSN
    torch.manual_seed(0)
    model = TSN(101, T, "RGB", base_model="resnet50", consensus_type="avg", tam=True, partial_bn=False)
    with torch.no_grad():
        model.new_fc.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(1))
    S.perturb_affine(model, 2)
    model = model.to(device)
    S.calibrate_bn(model, S.seeded_randn((4, T, 3, size, size), 3, device))
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                m.running_var.clamp_(min=0.05)
    model.eval()
    bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    hooks = [ComputeNormStatsHook(m, clip_len=T, stat_type="spatiotemp", before_norm=False, batch_size=2) for m in bn2d]
    with torch.no_grad():
        model(S.seeded_randn((2, T, 3, size, size), 1000, device))
    means = [h.batch_mean.cpu().numpy() for h in hooks]
    vars_ = [h.batch_var.cpu().numpy() for h in hooks]
    for h in hooks:
        h.close()
    mp, vp = S.write_stat_files(tmp, means, vars_, tag="bench")
    return model, mp, vp


def build_swin_and_stats(tmp, size, T, device):
    """Seeded Video Swin-B; source statistics = moments of a calibration pass (seed 1000) on the 52 LayerNorm outputs
    after the first one (what compute_statistics writes for the arch)."""
    from vitta_amd import synthetic as S
    from vitta_amd.bns_utils import choose_layers
    from vitta_amd.norm_stats import ComputeNormStatsHook
    model = S.build_swin(101, 0).to(device)
    lns = [m for _, m in choose_layers(model, [nn.LayerNorm])][1:]
    hooks = [ComputeNormStatsHook(m, clip_len=T, stat_type="spatiotemp", before_norm=False, batch_size=1) for m in lns]
    with torch.no_grad():
        model(S.seeded_randn((1, 2, 3, T, size, size), 1000, device))
    means = [h.batch_mean.cpu().numpy() for h in hooks]
    vars_ = [h.batch_var.cpu().numpy() for h in hooks]
    for h in hooks:
        h.close()
    mp, vp = S.write_stat_files(tmp, means, vars_, tag="swin")
    return model, mp, vp


def make_swin_args(tmp, size, clip_length, optimizer, device, n_videos):
    from vitta_amd import scripts
    a = scripts.swin_ucf101_args([])
    a.datatype, a.input_size, a.scale_size, a.workers, a.verbose = "synthetic", size, size, 0, False
    a.clip_length, a.result_dir, a.num_classes, a.batch_size = clip_length, tmp, 101, 1
    a.update_only_bn_affine = optimizer == "adam_affine"
    a.synthetic_n_videos, a.synthetic_device = n_videos, device
    return a


def run_gpu(opt, rank, world, device):
    from vitta_amd import data, tta
    tmp = tempfile.mkdtemp(prefix="vitta_bench_")
    n_videos = 64  # a stream of >= 64 distinct videos per GPU (14.5 MB each, resident in HBM before the timed region)
    if opt.arch == "swin":
        n_videos = min(n_videos, 16)  # 77 MB per clip pair
        model, mp, vp = build_swin_and_stats(tmp, opt.size, opt.clip_length, device)
        args = make_swin_args(tmp, opt.size, opt.clip_length, opt.optimizer, device, n_videos)
    else:
        model, mp, vp = build_model_and_stats(tmp, opt.size, opt.clip_length, device)
        args = make_args(tmp, opt.size, opt.clip_length, opt.optimizer, device, n_videos)
    args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
    args.synthetic_seed = 10000 * rank  # every rank adapts to its own videos (weak scaling)
    log("model + source statistics ready")
    adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model), args)
    if opt.arch == "swin":
        tta_set = data.build_videoswin_dataset(args, "val", "tta")
        eval_set = data.build_videoswin_dataset(args, "val", "eval")
    else:
        tta_set = data.build_tanet_dataset(args, "val", "tta")
        eval_set = data.build_tanet_dataset(args, "val", "eval")
    torch.cuda.synchronize()

    # live timing of the moments kernel: one (start, stop) event pair per step, on the launch stream
    pairs = []

    from vitta_amd import ops

    def new_events():
        pair = ops.KernelEventPair()
        pairs.append(pair)
        return pair

    def one_step(i):
        x, _ = tta_set[i % n_videos]
        if not opt.sequential:
            # overlapped schedule (the shipped default, ViTTAAdapter.step): video i is adapted while video i-1 is
            # evaluated on a second stream under the same weights; every step still holds one adaptation step and
            # one evaluation forward
            ev = eval_set[(i - 1) % n_videos][0]
            adapter.set_adapt_mode()
            return adapter.step(adapter.shape_tta_input(x.unsqueeze(0)), adapter.shape_eval_input(ev.unsqueeze(0)))
        ev, _ = eval_set[i % n_videos]
        if adapter._graph is not None:  # hipGraph replay: mode switches / hook (de)registration are baked in
            adapter.adapt_step(adapter.shape_tta_input(x.unsqueeze(0)))
            return adapter.evaluate(adapter.shape_eval_input(ev.unsqueeze(0)))
        adapter.set_adapt_mode()
        adapter.adapt_step(adapter.shape_tta_input(x.unsqueeze(0)))
        adapter.close_hooks()
        out = adapter.evaluate(adapter.shape_eval_input(ev.unsqueeze(0)))
        adapter.add_hooks_back()
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(opt.warmup):
        one_step(i)
        if i == 0:
            torch.cuda.synchronize()
            log("first warm-up step done")
    torch.cuda.synchronize()
    use_graph = not opt.no_graph and opt.warmup >= 2
    run_gpu.split_graphs = False
    if use_graph:
        x, _ = tta_set[0]
        ev, _ = eval_set[0]
        try:
            adapter.capture_graphs(adapter.shape_tta_input(x.unsqueeze(0)), adapter.shape_eval_input(ev.unsqueeze(0)),
                                   segmented=opt.segmented_graph and not opt.graph_collectives, overlap_eval=not opt.sequential,
                                   collectives_in_graph=True if opt.graph_collectives else (False if opt.segmented_graph else None))
            one_step(opt.warmup)  # first replay outside the timed region
            torch.cuda.synchronize()
            run_gpu.split_graphs = adapter._graph is not None and adapter._graph.get("step") == "split"
            log("hipGraphs captured")
        except Exception as e:  # noqa: BLE001
            if world == 1 and not opt.force_exchanges:
                raise
            # data-parallel: a rank whose capture failed goes on eagerly -- the exchanges of the segmented form are eager launches in
            # the same order either way, so the ranks' collective sequences still match; the line says what ran
            log(f"graph capture failed on rank {rank} ({e!r}): this rank runs the timed steps eagerly")
            adapter._abandon_step()
            torch.cuda.synchronize()
            use_graph = False
            run_gpu.capture_error = repr(e)[:200]
    log("warm-up done")
    if not use_graph:
        adapter.engine.timing_events = new_events
    def timed_block(b):
        barrier()
        t0 = time.perf_counter()
        for i in range(opt.steps):
            one_step(opt.warmup + b * opt.steps + i)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:  # the block's time is the slowest rank's (and the line says which rank that was)
            mine = torch.tensor([dt], dtype=torch.float64, device=device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(every, mine)
            per_rank = [float(t.item()) for t in every]
            dt = max(per_rank)
            run_gpu.per_rank_blocks.append(per_rank)
        return dt

    run_gpu.per_rank_blocks = []

    blocks = [timed_block(0)]
    # every rank derives the same block count from the same (reduced) first block
    # (the secondary legs -- timed_only -- take the median of up to five blocks, the headline of up to forty)
    n_blocks = max(1, min(5 if opt.timed_only else 40, int(np.ceil((0.5 if opt.timed_only else opt.min_seconds) / max(blocks[0], 1e-6)))))
    for b in range(1, n_blocks):
        blocks.append(timed_block(b))
    elapsed = float(np.median(blocks))
    run_gpu.blocks = blocks
    log(f"timed region done: {len(blocks)} blocks of {opt.steps} steps, median {elapsed:.3f}s, first {blocks[0]:.3f}s "
        f"({'hipGraph replay' if use_graph else 'eager'})")
    eager_elapsed = float("nan")
    run_gpu.conv = None
    run_gpu.host_fed = None
    run_gpu.exchange = None
    if world > 1 or opt.force_exchanges:
        # where a data-parallel step's time goes: the two exchanges bracketed by stream events on every rank, a few steps behind the
        # timed region (vitta_amd/exchange_timing.py; collectives captured INSIDE a graph cannot carry events: the three-segment and
        # eager forms launch them eagerly)
        from vitta_amd import exchange_timing as XT
        n_x = min(opt.steps, 10)
        XT.RECORDS = []
        barrier()
        tx = time.perf_counter()
        for i in range(n_x):
            one_step(opt.warmup + i)
        barrier()
        tx = (time.perf_counter() - tx) / n_x
        mine = dict(rank=rank, ms_per_step_while_timing=1e3 * tx, **{k: v for k, v in XT.summary(n_x).items()})
        XT.RECORDS = None
        every = [mine]
        if world > 1:
            try:
                every = [None] * world
                torch.distributed.all_gather_object(every, mine)
            except Exception as e:  # noqa: BLE001
                log(f"exchange report not gathered: {e!r}")
                every = [mine]
        res = {"steps": n_x, "per_rank": every,
               "note": "stream-event time of each eagerly launched exchange on its launching stream (the gradient buckets issued blocking, one "
                       "event pair each, while this is measured: an upper bound of what the overlapped bucketed form exposes); empty where "
                       "the collectives are captured inside the step's graph"}
        for kind in ("moments", "gradients"):
            vals = [r[kind] for r in every if isinstance(r, dict) and kind in r]
            if vals:
                res[kind] = {"bytes_per_step": vals[0]["bytes_per_step"], "calls_per_step": vals[0]["calls_per_step"],
                             "ms_per_step_mean_over_ranks": float(np.mean([v["ms_per_step"] for v in vals])),
                             "ms_per_step_max_over_ranks": float(np.max([v["ms_per_step"] for v in vals])),
                             "slowest_rank": int(np.argmax([v["ms_per_step"] for v in vals]))}
        run_gpu.exchange = res
        log(f"exchange decomposition: { {k: v for k, v in res.items() if k in ('moments', 'gradients')} }")
    if rank == 0 and world == 1 and not opt.timed_only and not opt.sequential and not opt.no_host_fed:
        # the same K steps with every video coming from (pinned) HOST memory: `value` above has its inputs resident in HBM; this is the
        # PCIe-inclusive rate -- (a) vitta_amd/prefetch.py, the next video's copies on a copy stream beside the current step (what
        # tta_standard does), (b) the reference's way, the upload in the compute stream in front of the step (corpus/basics.py:612-623)
        from vitta_amd.prefetch import DevicePrefetcher
        nv = min(n_videos, 4)
        hx = [tta_set[i][0].unsqueeze(0).cpu().pin_memory() for i in range(nv)]
        he = [eval_set[i][0].unsqueeze(0).cpu().pin_memory() for i in range(nv)]
        res = {"mb_per_video": round((hx[0].numel() * hx[0].element_size() + he[0].numel() * he[0].element_size()) / 1e6, 2)}
        for mode in ("prefetch", "upload_in_stream"):
            times = []
            for rep in range(3):
                if mode == "prefetch":
                    pt = DevicePrefetcher(((hx[i % nv],) for i in range(opt.steps)), device)
                    pe = DevicePrefetcher(((he[(i - 1) % nv],) for i in range(opt.steps)), device)
                    pt.ahead(), pe.ahead()
                barrier()
                t0 = time.perf_counter()
                for i in range(opt.steps):
                    if mode == "prefetch":
                        (x,), (ev,) = next(pt), next(pe)
                    else:
                        x, ev = hx[i % nv].to(device, non_blocking=True), he[(i - 1) % nv].to(device, non_blocking=True)
                    adapter.set_adapt_mode()
                    adapter.step(adapter.shape_tta_input(x), adapter.shape_eval_input(ev))
                    if mode == "prefetch":
                        pt.ahead(), pe.ahead()
                barrier()
                times.append((time.perf_counter() - t0) / opt.steps)
            res["ms_per_step_" + mode] = round(1e3 * float(np.median(times)), 4)
        res["videos_per_s"] = round(1e3 / res["ms_per_step_prefetch"], 2)
        res["note"] = ("every video from pinned host memory; prefetch = next video's copies on a copy stream beside the current step "
                       "(vitta_amd/prefetch.py, what tta_standard does); upload_in_stream = copies in the compute stream in front of the step")
        run_gpu.host_fed = res
        log(f"host-fed steps: {res}")
    if opt.timed_only:
        run_gpu.mode, run_gpu.eager_ms = ("hipGraph replay" if use_graph else "eager launches"), None
        return elapsed, float("nan"), float("nan"), None, adapter
    conv_events = []
    if use_graph:
        # A kernel inside a replayed graph cannot be bracketed by events; its duration does not depend on how it was
        # launched, so the K steps are repeated eagerly (same videos, same kernels) with an event pair attached to the
        # dispatch of every convolution launch (TANet: the dominant kernel) / of the moments kernel (Swin).
        from vitta_amd import conv as CV, fused_bn, fused_ln
        graph, adapter._graph = adapter._graph, None
        plan_of_graph = adapter.engine.plan  # the captured graphs write this plan's buffers (the eager exchanges read them)
        if opt.arch == "swin":
            fused_bn.ENABLED = fused_ln.ENABLED = False  # stand-alone moments kernel on the hooked LayerNorm outputs
            adapter.engine.timing_events = new_events
        else:
            def conv_timing(flops, key):
                ev = ops.KernelEventPair()
                conv_events.append((flops, key, ev))
                return ev
            CV.TIMING = conv_timing
            CV.KERNEL_TRACE = []  # kernel family of every launch, in launch order
        barrier()
        te = time.perf_counter()
        n_rep = opt.steps if opt.arch == "swin" else min(opt.steps, 12)
        for i in range(n_rep):
            one_step(opt.warmup + i)
        barrier()
        eager_elapsed = (time.perf_counter() - te) * opt.steps / n_rep
        CV.TIMING = None
        families, CV.KERNEL_TRACE = (CV.KERNEL_TRACE or []), None
        adapter._graph = graph
        adapter.engine.plan = plan_of_graph
        fused_bn.ENABLED = fused_ln.ENABLED = True
        log(f"eager repeat of {n_rep} timed steps (kernel events): {eager_elapsed:.3f}s per {opt.steps}")
        run_gpu.conv = None
        if conv_events:
            torch.cuda.synchronize()
            fl = np.array([f for f, _, _ in conv_events], dtype=np.float64)
            ms = np.array([ev.elapsed_ms() for _, _, ev in conv_events], dtype=np.float64)
            by_shape = {}
            fam = families if len(families) == len(conv_events) else [None] * len(conv_events)
            b3 = np.array([k == 3 for k in fam])  # _lib.CONV_KERNEL_B3
            peak = np.where(b3, MFMA_BF16_PEAK_TF / B3_PRODUCTS, MFMA_F32_PEAK_TF)
            # per launch: the composite bound t_roof = max(flops / the instruction's matrix peak, algorithmic bytes / 8 TB/s) --
            # the 64 -> 256 layer at 56 x 56 is bandwidth-bound, layer 4 matrix-bound; frac_roof = t_roof / duration
            by = np.array([key[4] for _, key, _ in conv_events], dtype=np.float64)
            by_min = np.array([key[7] if len(key) > 7 else key[4] for _, key, _ in conv_events], dtype=np.float64)  # x + weights + y only
            t_roof = np.maximum(fl / (peak * 1e9), by / (HBM_PEAK_GBS * 1e6))  # ms
            for (f, key, _), t, kid, tr in zip(conv_events, ms, fam, t_roof):
                r = by_shape.setdefault(key[:4] + (kid,), [0, 0.0, 0.0, 0.0, 0.0])
                r[0] += 1
                r[1] += f
                r[2] += t
                r[3] += key[4]
                r[4] += tr
            names = {0: "conv_igemm (fp32)", 1: "conv_sk (fp32)", 2: "conv_pw (fp32)", 3: "conv_b3 (split bf16)", None: "?"}
            pooled = np.array([bool(key[5]) if len(key) > 5 else False for _, key, _ in conv_events])  # launches that also pool (VITTA_CONV_POOL)
            folded = np.array([bool(key[6]) if len(key) > 6 else False for _, key, _ in conv_events])  # ... that carry a bn3 + add + ReLU backward
            n_folded = int(folded.sum())
            pooled = pooled | folded  # "plain" below = convolution + its own BatchNorm epilogue only
            run_gpu.conv = dict(launches=len(fl), steps=n_rep, flops=float(fl.sum()), ms=float(ms.sum()),
                                pooled_launches=int(pooled.sum()) - n_folded, folded_launches=n_folded, plain_flops=float(fl[~pooled].sum()), plain_ms=float(ms[~pooled].sum()),
                                plain_ms_at_peak=float((fl / (peak * 1e9))[~pooled].sum()),
                                ms_at_peak=float((fl / (peak * 1e9)).sum()), b3_launches=int(b3.sum()), b3_flops=float(fl[b3].sum()),
                                bytes=float(by.sum()), bytes_min=float(by_min.sum()), ms_at_roof=float(t_roof.sum()),
                                hbm_bound_launches=int((by / (HBM_PEAK_GBS * 1e6) > fl / (peak * 1e9)).sum()),
                                by_shape=[dict(C=k[0], K=k[1], taps=k[2], positions=k[3], kernel=names.get(k[4], str(k[4])),
                                               launches_per_step=v[0] / n_rep, avg_us=1e3 * v[2] / v[0], tflops=v[1] / v[2] / 1e9,
                                               algorithmic_mb=v[3] / v[0] / 1e6, t_roof_us=1e3 * v[4] / v[0],
                                               bound="hbm" if v[3] / (HBM_PEAK_GBS * 1e6) > v[1] / (1e9 * (MFMA_BF16_PEAK_TF / B3_PRODUCTS if k[4] == 3 else MFMA_F32_PEAK_TF)) else "mfma",
                                               frac_roof=v[4] / v[2])
                                          for k, v in sorted(by_shape.items(), key=lambda kv: -kv[1][2])])
            conv_events.clear()
    adapter.engine.timing_events = None
    torch.cuda.synchronize()
    kern_ms = float(np.mean([p.elapsed_ms() for p in pairs])) if pairs else float("nan")

    # adapt-only timing (no evaluation forward), same videos, for the report
    adapt_only = float("nan")
    if not opt.sequential and use_graph and world == 1 and not opt.force_exchanges:
        # the overlapped capture is one graph holding adaptation and evaluation: for SURVEY 8d's "(i) adapt step only"
        # capture the plain adapt / eval graphs as well (after the timed region; a failure only drops this figure)
        try:
            x, _ = tta_set[0]
            ev, _ = eval_set[0]
            adapter._graph = None
            adapter.capture_graphs(adapter.shape_tta_input(x.unsqueeze(0)), adapter.shape_eval_input(ev.unsqueeze(0)),
                                   segmented=False, overlap_eval=False)
            opt_sequential_for_adapt_only = True
        except Exception as e:  # noqa: BLE001
            log(f"adapt-only graphs not captured: {e!r}")
            opt_sequential_for_adapt_only = False
    else:
        opt_sequential_for_adapt_only = opt.sequential
    if opt_sequential_for_adapt_only:
        for i in range(2):
            x, _ = tta_set[i % n_videos]
            adapter.set_adapt_mode()
            adapter.adapt_step(adapter.shape_tta_input(x.unsqueeze(0)))
        barrier()
        t1 = time.perf_counter()
        for i in range(max(4, opt.steps // 3)):
            x, _ = tta_set[i % n_videos]
            adapter.set_adapt_mode()
            adapter.adapt_step(adapter.shape_tta_input(x.unsqueeze(0)))
        barrier()
        adapt_only = (time.perf_counter() - t1) / max(4, opt.steps // 3)

    streaming = None
    run_gpu.one_video = None
    if rank == 0 and not opt.no_streaming:
        streaming = streaming_moments(adapter, device)
        if opt.arch == "tanet":  # the in-step size (one video's hooked features, Infinity-Cache resident)
            run_gpu.one_video = streaming_moments(adapter, device, copies=1, reps=30, target_blocks=None)
        log("streaming-size moments done")
    one_graph = use_graph and adapter._graph is not None and ("step" in adapter._graph and adapter._graph["step"] is not None
                                                              or "adapt" in adapter._graph)
    split = use_graph and getattr(run_gpu, "split_graphs", False)  # (the form the TIMED steps ran in; the adapt-only graphs above replaced it)
    run_gpu.mode = "hipGraph replay (adaptation and evaluation as separate graphs on two streams)" if split else ("hipGraph replay" + (" (one graph, RCCL all-reduces captured)" if (one_graph and adapter.bucket is not None) else
                                         " (3 segments, exchanges eager)" if (world > 1 or opt.segmented_graph or adapter.bucket is not None) else "")) \
        if use_graph else "eager launches"
    run_gpu.eager_ms = (1e3 * eager_elapsed / opt.steps) if use_graph else None
    return elapsed, kern_ms, adapt_only, streaming, adapter


def streaming_moments(adapter, device, copies=16, reps=20, target_blocks=4096):
    """The same batched launch on features that cannot be cache resident: every hooked layer with
    `copies` videos' worth of frames (16 x 178 MB = 2.85 GB >> 256 MiB Infinity Cache)."""
    from vitta_amd import ops
    base = adapter.engine.plan.shapes
    shapes = [(outer * copies, c, inner, layout) for outer, c, inner, layout in base]
    # twice the workgroups of the in-step launch (same speed at this size): the two show up as separate lines of a
    # rocprofv3 summary split by launch geometry
    plan = ops.StatPlan(shapes, device, target_blocks=target_blocks) if target_blocks else ops.StatPlan(shapes, device)
    feats = [torch.randn(outer * c * inner, device=device) for outer, c, inner, _ in shapes]
    nbytes = 4 * sum(f.numel() for f in feats)
    shift = torch.zeros(plan.total_channels, device=device)
    times = []
    for r in range(reps + 3):
        ev = ops.KernelEventPair()
        plan.moments(feats, shift, events=ev)
        torch.cuda.synchronize()
        if r >= 3:
            times.append(ev.elapsed_ms())
    ms = float(np.mean(times))
    return dict(bytes=nbytes, ms=ms, achieved=nbytes / ms / 1e6, frac=nbytes / ms / 1e6 / HBM_PEAK_GBS,
                workgroups=plan.num_blocks)



class _StreamTimer:
    """start / stop events on the current stream around one launch (ops.KTIMING): the launch's duration plus the few
    microseconds of the event records -- small against the 50-500 us dense and attention launches of Video Swin-B."""
    records = None

    def __init__(self, kind, flops, valu_ops=0.0):
        self.kind, self.flops, self.valu_ops = kind, flops, valu_ops
        self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.a.record()

    def stop(self):
        self.b.record()
        _StreamTimer.records.append(self)


def swin_leg(device, views, frames, window_depth, classes, bf16, steps, warmup=3, sgd_all=False):
    """One Video Swin-B configuration: per-video iteration (adapt step + evaluation forward, overlapped schedule, hipGraph
    replay) timed over `steps` videos, then two eager steps with every dense (gemm.hip) and window attention launch
    bracketed by stream events."""
    from vitta_amd import data, ops, scripts, tta
    from vitta_amd import synthetic as S
    from vitta_amd.bns_utils import choose_layers
    from vitta_amd.norm_stats import ComputeNormStatsHook
    tmp = tempfile.mkdtemp(prefix="vitta_bench_swin_")
    old = (ops.WMSA_BF16, ops.DENSE_BF16)
    ops.WMSA_BF16 = ops.DENSE_BF16 = bool(bf16)
    try:
        size = 224
        model = S.build_swin(classes, 0, window_size=(window_depth, 7, 7)).to(device)
        lns = [m for _, m in choose_layers(model, [nn.LayerNorm])][1:]
        hooks = [ComputeNormStatsHook(m, clip_len=frames, stat_type="spatiotemp", before_norm=False, batch_size=1) for m in lns]
        with torch.no_grad():
            model(S.seeded_randn((1, 1, 3, frames, size, size), 1000, device))
        means, vars_ = [h.batch_mean.cpu().numpy() for h in hooks], [h.batch_var.cpu().numpy() for h in hooks]
        for h in hooks:
            h.close()
        mp, vp = S.write_stat_files(tmp, means, vars_, tag="swin")
        args = scripts.swin_ucf101_args([])
        args.datatype, args.input_size, args.scale_size, args.workers, args.verbose = "synthetic", size, size, 0, False
        args.clip_length, args.result_dir, args.num_classes = frames, tmp, classes
        args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
        args.update_only_bn_affine = not sgd_all
        args.n_augmented_views, args.window_size = views, (window_depth, 7, 7)
        n_videos = 4
        args.synthetic_n_videos, args.synthetic_device = n_videos, device
        adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model), args)
        tta_set = data.build_videoswin_dataset(args, "val", "tta")
        eval_set = data.build_videoswin_dataset(args, "val", "eval")

        def one(i):
            adapter.set_adapt_mode()
            return adapter.step(tta_set[i % n_videos][0].unsqueeze(0), eval_set[(i - 1) % n_videos][0].unsqueeze(0))

        for i in range(warmup):
            one(i)
        torch.cuda.synchronize()
        adapter.capture_graphs(tta_set[0][0].unsqueeze(0), eval_set[0][0].unsqueeze(0), overlap_eval=True)
        one(0)
        torch.cuda.synchronize()
        blocks = []  # three timed blocks of `steps` videos, the median reported (a 4-step block is 0.15 s: one slow step shows)
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                one(i)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / steps)
        dt = sorted(blocks)[1]
        # per-kernel figures: two eager steps with stream events around the dense and attention launches
        graph, adapter._graph = adapter._graph, None
        _StreamTimer.records, ops.KTIMING = [], _StreamTimer
        for i in range(2):
            one(i)
        torch.cuda.synchronize()
        ops.KTIMING = None
        adapter._graph = graph
        kinds = {}
        for r in _StreamTimer.records:
            k = kinds.setdefault(r.kind, [0, 0.0, 0.0, 0.0])
            k[0] += 1
            k[1] += r.flops
            k[2] += r.a.elapsed_time(r.b)
            k[3] += getattr(r, "valu_ops", 0.0)
        _StreamTimer.records = None
        roof = {}
        for kind, (n, fl, ms, vops) in sorted(kinds.items()):
            peak = MFMA_BF16_PEAK_TF if kind.endswith("bf16") else MFMA_F32_PEAK_TF
            tf = fl / ms / 1e9
            roof[kind] = {"launches_per_step": n / 2, "gflop_per_step": fl / 2e9, "ms_per_step": ms / 2, "achieved": tf, "peak": peak,
                          "unit": "TFLOP/s", "frac": tf / peak}
            if vops > 0:
                # composite bound of the attention family: with head dim 32 a score is 128 (forward) / 320 (backward) matrix flops but ~8 /
                # ~10 vector lane-operations (bias + mask terms, exp, dS, packing), and the vector pipe does 78.6e12 lane-operations a
                # second (157.3 TFLOP/s of fp32 FMA / 2): t_roof = max(matrix time, vector time)
                t_m, t_v = fl / (peak * 1e9), vops / (VALU_LANE_OPS_PER_S * 1e-3)
                roof[kind].update({"valu_lane_ops_per_step": vops / 2, "t_roof_ms_per_step": max(t_m, t_v) / 2,
                                   "bound": "valu" if t_v > t_m else "mfma", "frac_composite": max(t_m, t_v) / ms})
        out = {"value": 1.0 / dt, "unit": "videos/s", "ms_per_step": 1e3 * dt, "steps": steps, "blocks_ms": [round(1e3 * b, 3) for b in blocks],
               "launch_mode": "hipGraph replay",
               "dtype": "f32 residual stream / statistics / softmax / accumulation; bf16 MFMA operands AND bf16 activations between the "
                        "LayerNorms, dense layers and window attention (the bf16 recipe's data flow)" if bf16 else "f32",
               "optimizer": "SGD all parameters (reference default)" if sgd_all else "Adam on LN affine (update_only_bn_affine)",
               "config": {"workload": f"Video Swin-B ViTTA online TTA, per-video iteration = adapt step ({views} views x {frames} frames x "
                                      f"224^2, window ({window_depth},7,7), {len(adapter.engine.hooks)} hooked LayerNorm layers, l1 stat "
                                      f"alignment + prediction consistency, backward, optimizer) + eval forward (1 view), "
                                      f"{classes} classes",
                          "schedule": "overlapped"},
               "roofline": roof,
               "roofline_note": "dense (gemm.hip) and window attention launches of two eager steps bracketed by stream events; flops = "
                                "2 M N K per product, 4 / 10 x tokens^2 x head_dim per (window, head) forward / backward (ALGORITHMIC: "
                                "S, dP, dV, dK, dQ once each; rounds 1-4 counted the 14 x the two-kernel backward executes); peak = the "
                                "matrix rate of the instruction the kernel issues (157.3 fp32, 2500 bf16 dense); attention also carries "
                                "the composite bound max(matrix time, vector-ALU time at ~8 / ~10 lane-operations per score)",
               "max_mem_GB": torch.cuda.max_memory_allocated() / 1e9}
        del adapter, model
        torch.cuda.empty_cache()
        return out
    finally:
        ops.WMSA_BF16, ops.DENSE_BF16 = old
        ops.KTIMING = None


def run_cpu_baseline(opt):
    """The reference path restated on the CPU (oracle/), same workload, bounded sample."""
    from oracle import cpu_path
    # threads = the CPUs this process may actually run on (cgroup / affinity), not every core of the host
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota ("<quota> <period>"), e.g. 16 CPUs on a 256-thread host
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
    sec, steps = cpu_path.time_tta_steps(size=opt.size, clip_length=opt.clip_length, optimizer=opt.optimizer,
                                         warmup=1, steps=opt.cpu_steps, budget_s=25.0, log=log)
    return dict(value=steps / sec, unit="videos/s", cores=cores, kind="port",
                sample=f"{steps} full per-video iterations (adapt step + eval forward) after 1 warm-up, TANet-R50 "
                       f"2x{opt.clip_length}x{opt.size}^2 fp32, torch CPU ops in the reference's op order, "
                       f"{cores} threads")


def _rocprof_moments():
    """The moments kernel's bandwidth as rocprofv3 measured it (profiles/r6_moments_rocprof.json, written by tools/prof_summary.py from the
    kernel trace of this very command; bench.py cannot run the profiler on itself): the bench's own figure above is from stream events."""
    f = os.path.join(ROOT, "profiles", "r6_moments_rocprof.json")
    try:
        return dict(json.load(open(f)), source=os.path.relpath(f, ROOT))
    except (OSError, ValueError):
        return None


def _trunk_pool_fold():
    from vitta_amd import conv as _cv, trunk as _tr
    return _tr.POOL_FOLD and _cv.ARITH == "b3"


def launch_ranks(opt):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) under torch.distributed.run on
    this node and hand rank 0's line through.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={opt.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {opt.gpus} without a launcher: starting {opt.gpus} ranks: {' '.join(cmd[1:9])} ...")

    def attempt(extra_env, extra_args):
        """One launch of the N ranks; returns (exit code, rank 0's JSON line or None).  stderr passes through."""
        p = subprocess.run(cmd + extra_args, env=dict(env, **extra_env), stdout=subprocess.PIPE)
        line = None
        for ln in p.stdout.decode(errors="replace").splitlines():
            ln = ln.strip()
            if ln.startswith("{") and ln.endswith("}"):
                try:
                    json.loads(ln)
                    line = ln
                except ValueError:
                    pass
        return p.returncode, line

    # The first multi-GPU contact must not come back empty: a rank that dies (e.g. the process-group watchdog aborting the
    # process, which no `except` in the rank can catch) costs this attempt only.  Fallbacks, each said in the line's `dp_graph`:
    # (1) as asked; (2) collectives outside every capture (three graph segments); (3) eager launches.
    ladder = [({}, []),
              ({"VITTA_GRAPH_COLLECTIVES": "0", "VITTA_BENCH_NOTE": "segments forced after a failed attempt (rc={rc})"}, ["--segmented-graph"]),
              ({"VITTA_GRAPH_COLLECTIVES": "0", "VITTA_BENCH_NOTE": "eager launches after two failed attempts (rc={rc})"}, ["--no-graph"])]
    rc = 0
    for k, (e, extra) in enumerate(ladder):
        if opt.graph_collectives and "--segmented-graph" in extra:
            extra = [x for x in extra if x != "--segmented-graph"]
        e = {kk: vv.format(rc=rc) for kk, vv in e.items()}
        rc, line = attempt(e, extra)
        if rc == 0 and line is not None:
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
            return 0
        log(f"attempt {k + 1} of {len(ladder)} gave rc={rc}, line={'yes' if line else 'none'}" + ("; retrying" if k + 1 < len(ladder) else ""))
    return rc or 1


def rank_report(rank, local, device):
    """What this rank sees (the driver verifies N ranks on N distinct devices)."""
    mine = dict(rank=rank, local_rank=local, pid=os.getpid(),
                dist_world_size=torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                dist_backend=torch.distributed.get_backend() if torch.distributed.is_initialized() else None)
    if device is not None:
        props = torch.cuda.get_device_properties(device)
        mine.update(device_index=device.index, device=props.name, pci_bus_id=getattr(props, "pci_bus_id", None),
                    uuid=str(getattr(props, "uuid", "")) or None)
    ranks = [mine]
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        try:
            gathered = [None] * torch.distributed.get_world_size()
            torch.distributed.all_gather_object(gathered, mine)
            ranks = gathered
        except Exception as e:  # noqa: BLE001  (the report must never cost the benchmark line)
            log(f"rank report not gathered: {e!r}")
    return ranks


def main():
    opt = parse()
    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ and not opt.force_exchanges:
        raise SystemExit(launch_ranks(opt))
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner through
    # C stdio when a communicator is created, flushed at exit -- after the JSON line): hand file descriptor 1 to stderr
    # for the run and keep the real stdout for the line alone.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(opt.gpus, 1):
        log(f"WORLD_SIZE={world} but --gpus {opt.gpus}: the line reports the world size the ranks actually form")
    if opt.rendezvous_only:
        device = None
        if torch.cuda.is_available():
            device = torch.device("cuda", local % torch.cuda.device_count())
            torch.cuda.set_device(device)
        if world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.distributed.init_process_group(opt.dist_backend if device is not None else "gloo")
        ranks = rank_report(rank, local, device)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if rank == 0:
            os.write(real_stdout, (json.dumps({"rendezvous_only": True, "n_gpus": len(ranks), "gpus_requested": opt.gpus,
                                               "ranks": ranks}) + "\n").encode())
        os.close(real_stdout)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world == 1 and opt.force_exchanges:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        from vitta_amd import tta as _tta
        _tta.FORCE_EXCHANGES = True  # (the data-parallel default: one graph with the RCCL calls captured; --segmented-graph: three)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if opt.dist_backend == "nccl" and world > torch.cuda.device_count():
            # more ranks than devices (e.g. `--gpus 2` typed on a one-GPU box): RCCL refuses two ranks on one device ("Duplicate GPU
            # detected").  Say so and rehearse the data-parallel path over gloo -- the line is marked, it is NOT a scaling figure.
            if rank == 0:
                print(f"[bench] WARNING: {world} ranks on {torch.cuda.device_count()} device(s): RCCL refuses duplicate devices -> the "
                      "exchanges run over gloo and the ranks time-share the device; this line is a rehearsal, not a scaling figure",
                      file=sys.stderr, flush=True)
            opt.dist_backend = "gloo"
        if opt.dist_backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:  # rehearsal of the data-parallel path with several ranks on ONE GPU (RCCL refuses duplicate devices)
            torch.distributed.init_process_group(opt.dist_backend)
    if opt.dense_bf16:
        from vitta_amd import ops as _ops
        _ops.DENSE_BF16 = True
    if opt.wmsa_bf16:
        from vitta_amd import ops as _ops
        _ops.WMSA_BF16 = True
    if opt.miopen_find:
        torch.backends.cudnn.benchmark = True
    # corpus/main_eval.py:77 sets cudnn.benchmark (an exhaustive MIOpen find on ROCm: minutes of search
    # per conv shape); the bench keeps MIOpen's default immediate mode

    elapsed, kern_ms, adapt_only, streaming, adapter = run_gpu(opt, rank, world, device)
    mode, eager_ms, conv, one_video = run_gpu.mode, run_gpu.eager_ms, run_gpu.conv, getattr(run_gpu, "one_video", None)
    videos = opt.steps * world
    value = videos / elapsed

    # what every rank saw (the driver can verify N ranks on N devices)
    ranks = rank_report(rank, local, device)
    dp_graph = getattr(adapter, "dp_graph", "eager") if "hipGraph" in mode else "eager"
    if getattr(run_gpu, "capture_error", None):
        dp_graph += f" (capture failed on this rank: {run_gpu.capture_error})"
    if os.environ.get("VITTA_BENCH_NOTE"):
        dp_graph += f" ({os.environ['VITTA_BENCH_NOTE']})"

    if opt.arch == "swin":
        algo_bytes = 4.0 * sum(o * c * i for o, c, i, _ in adapter.engine.plan.shapes)  # 253.7 MB at 2x16x224^2
    else:
        algo_bytes = 4 * HOOKED_ELEMENTS_PER_VIDEO * (opt.size / 224.0) ** 2 * (opt.clip_length / 8.0)

    # HBM traffic of the moments launch from the PMC passes committed under profiles/ (bench.py itself
    # cannot run rocprofv3 --pmc): corrected read bytes + write bytes of the in-step (1 video) launch
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r1_moments_pmc.json")
    if os.path.exists(pmc_file) and opt.size == 224 and opt.clip_length == 8 and opt.arch == "tanet":
        pmc = json.load(open(pmc_file))["in_step_1_video"]
        traffic = pmc["hbm_read_bytes_corrected"] + pmc["hbm_write_bytes"]

    if opt.arch == "swin":
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms == kern_ms else None
        roofline = {"kernel": "moments_nhwc_partial_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
                    "algorithmic_bytes": algo_bytes, "avg_ms": kern_ms, "streaming": streaming}
    else:
        moments = {"kernel": "moments_nchw_partial_kernel (29 layers, 1 launch; stand-alone: in the TANet step the "
                             "hooked moments ride in the convolution epilogues)",
                   "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "one_video": one_video, "streaming": streaming,
                   "traffic_one_video": traffic, "rocprof": _rocprof_moments(),
                   "traffic_source": "profiles/r1_moments_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 per "
                                     "the gfx950 note)"}
        if conv:
            tf = conv["flops"] / conv["ms"] / 1e9
            conv_pmc = None
            cpf = os.path.join(ROOT, "profiles", "r5_conv_traffic_pmc.json")
            if not os.path.exists(cpf):
                cpf = os.path.join(ROOT, "profiles", "r4_conv_traffic_pmc.json")
            if os.path.exists(cpf) and opt.size == 224 and opt.clip_length == 8:
                conv_pmc = json.load(open(cpf)).get("hbm_bytes_per_launch")
            frac = conv["ms_at_peak"] / conv["ms"]
            roofline = {"kernel": "conv_b3_kernel (+ conv_sk / conv_pw for the stride-2 gathers) behind vitta_conv_f32: every "
                                  "bottleneck convolution of the trunk, forward + data gradient + evaluation forward",
                        "bound": "mfma", "achieved": tf, "peak": tf / frac, "unit": "TFLOP/s", "frac": frac,
                        "frac_of_fp32_matrix_peak": tf / MFMA_F32_PEAK_TF,
                        # composite: every launch against max(flops / its matrix peak, algorithmic bytes / 8 TB/s); time-weighted
                        "frac_composite": conv["ms_at_roof"] / conv["ms"],
                        "algorithmic_bytes_per_step": conv["bytes"] / conv["steps"],
                        # the round-3 definition (x + fp32 weights + y, no epilogue input streams), kept so that rounds stay comparable
                        "algorithmic_bytes_per_step_x_w_y_only": conv["bytes_min"] / conv["steps"],
                        "hbm_bound_launches_per_step": conv["hbm_bound_launches"] / conv["steps"],
                        "peaks": {"conv_b3 (v_mfma_f32_32x32x16_bf16, 6 split products per multiply-add)": MFMA_BF16_PEAK_TF / B3_PRODUCTS,
                                  "fp32 kernels (v_mfma_f32_32x32x2_f32)": MFMA_F32_PEAK_TF},
                        "split_bf16_share_of_flops": conv["b3_flops"] / conv["flops"],
                        "traffic": conv_pmc,
                        "traffic_source": (os.path.relpath(cpf, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                           "passes, FETCH x2 per the gfx950 note; per launch)") if conv_pmc else None,
                        "launches_per_step": conv["launches"] / conv["steps"],
                        "algorithmic_flops_per_launch": conv["flops"] / conv["launches"],
                        "avg_launch_us": 1e3 * conv["ms"] / conv["launches"],
                        "algorithmic_flops_per_step": conv["flops"] / conv["steps"],
                        "kernel_ms_per_step": conv["ms"] / conv["steps"],
                        "share_of_step": conv["ms"] / conv["steps"] / (1e3 * elapsed / opt.steps),
                        "pooled_means_in_epilogue": bool(_trunk_pool_fold()),
                        # the same two fractions over the launches that are convolutions only (the conv1 launches that also pool left out)
                        "pooling_launches_per_step": conv["pooled_launches"] / conv["steps"],
                        "bn3_fold_launches_per_step": conv["folded_launches"] / conv["steps"],
                        "frac_plain_launches": (conv["plain_ms_at_peak"] / conv["plain_ms"]) if conv["plain_ms"] > 0 else None,
                        "frac_of_fp32_matrix_peak_plain_launches": (conv["plain_flops"] / conv["plain_ms"] / 1e9 / MFMA_F32_PEAK_TF)
                        if conv["plain_ms"] > 0 else None,
                        "note": ("round 5: the conv1 data-gradient launches (15 per step) also carry the previous block's bn3 + add + ReLU "
                                 "backward in their epilogue (three more input streams, a second output, d gamma / d beta: 15 stand-alone "
                                 "passes of 4p-channel tensors less per step; VITTA_TRUNK_BN3_FOLD=0 measures frac_of_fp32_matrix_peak 0.02 "
                                 "higher and the step 0.13 ms slower); frac_plain_launches leaves those and the pooling launches out.  "
                                 "the conv1 launches of the bottlenecks (32 per step) also carry TAM's spatial average pooling "
                                 "(VITTA_CONV_POOL, +1.5-3 us each in place of 32 pooling launches: the family's time, hence frac, "
                                 "includes it; VITTA_TRUNK_POOL_FOLD=0 measures 0.015 higher on the same box).  "
                                 if _trunk_pool_fold() else "") + "per-launch durations from hipEvent pairs attached to each dispatch in an eager repeat of "
                                "the timed steps (a replayed hipGraph cannot carry events); flops = 2 x output positions "
                                "x C x K x taps per launch (vitta_conv_flops), whatever the instruction; peak = flops / "
                                "(sum over launches of flops / the peak of the instruction that launch issues); "
                                "share_of_step > what a serial schedule would allow where the evaluation stream overlaps "
                                "the adaptation stream; frac_composite / by_shape[*].frac_roof = (sum of per-launch "
                                "max(flops / matrix peak, algorithmic bytes / 8 TB/s)) / (sum of durations), algorithmic bytes "
                                "= x + fp32 weights + y + the epilogue's input streams, each once",
                        "by_shape": conv["by_shape"][:14], "moments": moments, "streaming": streaming}
        else:
            ov = one_video or {}
            roofline = {"kernel": moments["kernel"], "bound": "hbm", "achieved": ov.get("achieved"), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ov.get("frac"), "traffic": traffic, "algorithmic_bytes": algo_bytes,
                        "avg_ms": ov.get("ms"), "streaming": streaming, "moments": moments}

    line = {
        "metric": f"videos/sec TTA step (TANet-R50, 2x{opt.clip_length}x{opt.size}^2), whole job", "value": value,
        "unit": "videos/s",
        "n_gpus": len(ranks) if world > 1 else 1, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": 1e3 * elapsed / opt.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if (opt.arch == "swin" or os.environ.get("VITTA_CONV_ARITH", "b3") != "b3") else
                 "f32 (convolutions: operands split into three bf16 terms, six bf16-MFMA products per multiply-add, fp32 accumulation)",
        "data": "synthetic", "blocks": len(getattr(run_gpu, "blocks", [])) or 1,
        "ms_per_step_first_block": 1e3 * getattr(run_gpu, "blocks", [elapsed])[0] / opt.steps,
        "config": {"workload": "TANet-R50 UCF101 ViTTA online TTA, per-video iteration = adapt step (2 views x "
                               f"{opt.clip_length} frames x {opt.size}^2, 29 hooked BN2d layers, l1 stat alignment + "
                               "prediction consistency, backward, optimizer) + eval forward (1 view)",
                   "optimizer": "Adam on BN affine (update_only_bn_affine)" if opt.optimizer == "adam_affine" else "SGD all parameters",
                   "videos_per_gpu_per_step": 1, "parallelism": f"dp{world}",
                   "exchanges": "moments all-reduce (43k floats) + gradient all-reduce" if (world > 1 or opt.force_exchanges) else "none",
                   "schedule": "sequential: adapt(i); eval(i)" if opt.sequential else
                               "overlapped: eval(i-1) on a second stream beside adapt(i), optimizer update after both "
                               "(same weights and results as the sequential order)"},
        "adapt_only_ms": (1e3 * adapt_only if adapt_only == adapt_only else None), "host_fed": getattr(run_gpu, "host_fed", None), "launch_mode": mode, "eager_ms_per_step": eager_ms,
        "roofline": roofline, "ranks": ranks, "dp_graph": dp_graph,
    }
    # first-contact diagnostics of a data-parallel run: how many distinct devices the ranks really sit on, which
    # rank was the slowest in how many of the timed blocks, and the two exchanges timed separately
    line["rccl_ranks_seen"] = len({(r.get("pci_bus_id"), r.get("uuid"), r.get("device_index")) for r in ranks if isinstance(r, dict)}) \
        if (world > 1 and opt.dist_backend == "nccl") else (1 if opt.force_exchanges else None)
    if world > 1:
        line["ranks_per_device"] = -(-world // max(1, torch.cuda.device_count()))  # > 1: time-shared devices (a rehearsal)
    prb = getattr(run_gpu, "per_rank_blocks", [])
    if prb:
        slow = [int(np.argmax(b)) for b in prb]
        line["slowest_rank"] = {"per_block": slow, "most_often": int(np.bincount(slow).argmax()),
                                "spread_ms_per_step": 1e3 * float(np.median([max(b) - min(b) for b in prb])) / opt.steps}
    if getattr(run_gpu, "exchange", None):
        line["exchange"] = run_gpu.exchange
    if opt.arch == "swin":
        n_ln = len(adapter.engine.hooks)
        line["metric"] = f"videos/sec TTA step (Video Swin-B, 2x{opt.clip_length}x{opt.size}^2), whole job"
        line["config"]["workload"] = (f"Video Swin-B UCF101 ViTTA online TTA, per-video iteration = adapt step (2 views x "
                                      f"{opt.clip_length} frames x {opt.size}^2, {n_ln} hooked LayerNorm layers, l1 stat "
                                      "alignment + prediction consistency, backward, optimizer) + eval forward (1 view)")
        line["config"]["optimizer"] = "Adam on LN affine (update_only_bn_affine)" if opt.optimizer == "adam_affine" \
            else "SGD all parameters"
        line["config"]["exchanges"] = "moments all-reduce + gradient all-reduce" if world > 1 else "none"
        line["roofline"]["kernel"] = f"moments_nhwc_partial_kernel ({n_ln} layers, 1 launch)"
        line["config"]["window_attention"] = "bf16 operands, fp32 softmax / accumulation" if opt.wmsa_bf16 else "fp32"
        line["config"]["dense_layers"] = ("gemm.hip, bf16 operands, fp32 accumulation" if opt.dense_bf16
                                          else "gemm.hip, exact fp32 MFMA")
        if opt.wmsa_bf16 or opt.dense_bf16:
            line["dtype"] = "f32 (bf16 MFMA operands: " + " + ".join(
                n for n, on in (("window attention", opt.wmsa_bf16), ("dense layers", opt.dense_bf16)) if on) + ")"
        line["roofline"]["note"] = ("stand-alone batched kernel timed on the step's own hooked LayerNorm outputs; in the "
                                    "shipped step these moments ride on the fused LayerNorm pass (ln_fwd_kernel)")
    del adapter
    torch.cuda.empty_cache()
    if opt.arch == "tanet" and opt.optimizer == "adam_affine" and not opt.no_sgd_all and not opt.timed_only and world == 1:
        # report both optimizer modes -- the reference's default (SGD over every parameter) in the same run
        import copy
        o2 = copy.copy(opt)
        o2.optimizer, o2.timed_only = "sgd_all", True
        log("second timing: SGD over all parameters ...")
        e2 = run_gpu(o2, rank, world, device)[0]
        line["sgd_all"] = {"value": videos / e2, "unit": "videos/s", "ms_per_step": 1e3 * e2 / opt.steps, "steps": opt.steps,
                           "optimizer": "SGD all parameters (reference default, corpus/basics.py:547-560)",
                           "launch_mode": run_gpu.mode}
        torch.cuda.empty_cache()
    if opt.arch == "tanet" and opt.optimizer == "adam_affine" and not opt.no_exact_fp32 and not opt.timed_only and world == 1 \
            and not opt.force_exchanges and os.environ.get("VITTA_CONV_ARITH", "b3") == "b3":
        # the strictly exact-fp32 figure in the SAME run: every convolution on v_mfma_f32_32x32x2_f32 (conv_sk / conv_pw /
        # conv.hip, VITTA_CONV_ARITH=f32) -- no split operands anywhere
        import copy
        from vitta_amd import conv as _cv
        o3 = copy.copy(opt)
        o3.timed_only = True
        old_arith, _cv.ARITH = _cv.ARITH, "f32"
        try:
            log("third timing: exact-fp32 convolutions (VITTA_CONV_ARITH=f32) ...")
            e3 = run_gpu(o3, rank, world, device)[0]
            line["exact_fp32"] = {"value": videos / e3, "unit": "videos/s", "ms_per_step": 1e3 * e3 / opt.steps, "steps": opt.steps,
                                  "dtype": "f32 (v_mfma_f32_32x32x2_f32 in every convolution; no split-bf16 operands)",
                                  "launch_mode": run_gpu.mode}
        except Exception as e:  # noqa: BLE001  (a leg must never cost the headline line)
            log(f"exact-fp32 leg failed: {e!r}")
            line["exact_fp32"] = {"error": repr(e)}
        finally:
            _cv.ARITH = old_arith
        torch.cuda.empty_cache()
    if opt.arch == "tanet" and opt.optimizer == "adam_affine" and not opt.no_forced_exchange_leg and not opt.timed_only and world == 1 \
            and not opt.force_exchanges and opt.size == 224:
        # N = 1 WITH both exchanges live (a one-rank RCCL group): what a SCALE run's N = 1 point pays beyond this line's headline; its own
        # process (a process-group watchdog abort cannot be caught in-process and must not cost the line)
        import subprocess
        try:
            log("one-rank RCCL leg (--gpus 1 --force-exchanges) in a child process ...")
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force-exchanges", "--timed-only", "--no-cpu-baseline",
                   "--steps", str(opt.steps), "--warmup", str(opt.warmup)]
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            free_port = sk.getsockname()[1]
            sk.close()
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port))
            pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
            sub = None
            for ln in pr.stdout.decode(errors="replace").splitlines():
                if ln.strip().startswith("{"):
                    sub = json.loads(ln)
            if pr.returncode == 0 and sub:
                line["forced_exchanges_n1"] = {"value": sub["value"], "unit": "videos/s", "ms_per_step": sub["ms_per_step"],
                                               "dp_graph": sub.get("dp_graph"), "exchange": sub.get("exchange"),
                                               "launch_mode": sub.get("launch_mode"),
                                               "note": "the same step with the moments and gradient all-reduces live on a one-rank RCCL group "
                                                       "(--gpus 1 --force-exchanges): the N = 1 point of the data-parallel form"}
            else:
                line["forced_exchanges_n1"] = {"error": f"rc={pr.returncode}"}
        except Exception as e:  # noqa: BLE001
            log(f"one-rank RCCL leg failed: {e!r}")
            line["forced_exchanges_n1"] = {"error": repr(e)}
    if opt.arch == "tanet" and opt.optimizer == "adam_affine" and not opt.no_swin and not opt.timed_only and world == 1 \
            and opt.size == 224:
        # the other half of north_star: Video Swin-B at BASELINE config 3's and config 5's shapes, in the same run
        for key, cfg in (("swin", dict(views=2, frames=16, window_depth=8, classes=101, bf16=False, steps=8)),
                         ("swin_c5_bf16", dict(views=4, frames=32, window_depth=16, classes=174, bf16=True, steps=4)),
                         ("swin_sgd_all", dict(views=2, frames=16, window_depth=8, classes=101, bf16=False, steps=6, sgd_all=True)),
                         # config 5's shape under the reference's DEFAULT optimizer: every table, weight and bias trains -- the bf16
                         # attention bins the relative-position table's gradient in LDS (round 5; before: the fp32 kernels)
                         ("swin_c5_bf16_sgd_all", dict(views=4, frames=32, window_depth=16, classes=174, bf16=True, steps=3, sgd_all=True))):
            try:
                log(f"Video Swin-B leg {key} ...")
                line[key] = swin_leg(device, **cfg)
                log(f"  {line[key]['ms_per_step']:.2f} ms per video")
            except Exception as e:  # noqa: BLE001  (a leg must never cost the headline line)
                log(f"leg {key} failed: {e!r}")
                line[key] = {"error": repr(e)}
            torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not opt.no_cpu_baseline and opt.arch == "tanet":
        log("cpu baseline ...")
        line["cpu_baseline"] = run_cpu_baseline(opt)
        log("cpu baseline done")
    if world > 1 or opt.force_exchanges:
        torch.distributed.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
