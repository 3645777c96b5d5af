"""Step-level parity on the MI355X: the product path (HIP kernels, batched engine or stand-alone hooks)
against golden vectors captured from the reference's own tta_standard (tools/refgen/gen_golden.py)."""
import pytest
import torch
import torch.nn as nn

import helpers as H
from test_host_cpu import BASE, assert_logits_close, check_tta_records, run_product_tta

pytestmark = pytest.mark.gpu

# First-step bounds on the GPU = the CPU suite's (tests/test_host_cpu.py::BASE) for the gradients and logits now that every
# TANet convolution is our own fp32-MFMA kernel (round 1 ran them in MIOpen / rocBLAS with their own reduction orders and
# needed grad_frac 2e-2); later steps are bounded by the reference's own noise floor exactly as on the CPU.
BASE_GPU = dict(loss_rel=5e-5, logit_frac=2e-3, grad_frac=5e-3, param_lr_mult=0.1)
# After the first step at most ONE sampled gradient tensor per step may exceed its own strict bound, by at most 30 % of max|g|
# (one sign(ema - source) flip of an L1 term among the sampled channels: tests/test_host_cpu.py::check_tta_records); the
# exact-fp32 convolution arithmetic runs without the allowance (strict per-tensor floors everywhere) so that a path bug
# cannot hide behind what the split arithmetic's other summation order needs.
# Calibration, round 6: with NO allowance the fourteen step tests of this file pass in three of four consecutive runs on one box; the fourth
# draws one flip (test_batch_of_two_matches_reference_on_gpu[b3]) -- the allowance covers a run-to-run event, not a standing error.
OUTLIERS = (1, 0.30)


@pytest.fixture
def conv_arith():
    """Switch the convolution arithmetic of vitta_conv_f32 for one test: conv_arith("f32") / ("b3")."""
    from vitta_amd import conv as CV
    old = CV.ARITH

    def set_(name):
        CV.ARITH = name
    yield set_
    CV.ARITH = old


def _dev():
    return torch.device("cuda:0")


def test_tanet_forward_matches_reference_on_gpu():
    g = H.golden("tanet_fwd.npz")
    model = H.build_tanet(11, 8, 0).to(_dev())
    x = H.seeded_randn((2, 8, 3, 64, 64), 21).to(_dev())
    from vitta_amd.norm_stats import ComputeNormStatsHook
    bn2d = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    hooks = [ComputeNormStatsHook(m, clip_len=8, stat_type="spatiotemp", before_norm=False, batch_size=2) for m in bn2d]
    with torch.no_grad():
        logits = model(x)
    for h in hooks:
        h.close()
    assert_logits_close(logits.cpu(), torch.from_numpy(g["logits"]), 2e-3)
    means = torch.cat([h.batch_mean for h in hooks]).cpu()
    vars_ = torch.cat([h.batch_var for h in hooks]).cpu()
    torch.testing.assert_close(means, torch.from_numpy(g["means"]), rtol=2e-3, atol=5e-4)
    torch.testing.assert_close(vars_, torch.from_numpy(g["vars"]), rtol=5e-3, atol=1e-4)


@pytest.mark.parametrize("mode", ["sgd", "adam"])
@pytest.mark.parametrize("use_engine", [True, False])
@pytest.mark.parametrize("arith", ["b3", "f32"])
def test_three_tta_steps_match_reference_on_gpu(tmp_path, mode, use_engine, arith, abi_calls, conv_arith):
    """arith f32: the exact-fp32 MFMA convolutions under the STRICT per-tensor bounds (no common floor); b3 (the default
    split-bf16 arithmetic): the same with the capped common floor after the first step."""
    conv_arith(arith)
    g = H.golden("tta3.npz")
    recs = run_product_tta(g, mode, tmp_path, _dev(), None, use_engine=use_engine)
    report = check_tta_records(g, mode, recs, BASE_GPU, outliers=OUTLIERS if arith == "b3" else None)
    for row in report:
        print("step %d %-60s err %.3e bound %.3e" % row)
    if use_engine:  # the hand-written trunk (stand-alone hooks are foreign to it: that mode checks the module path)
        abi_calls.assert_tanet_trunk()


@pytest.mark.parametrize("arith", ["b3", "f32"])
def test_batch_of_two_matches_reference_on_gpu(tmp_path, arith, abi_calls, conv_arith):
    conv_arith(arith)
    g = H.golden("tta3_bz2.npz")
    recs = run_product_tta(g, "sgd", tmp_path, _dev(), None, batch_size=2)
    check_tta_records(g, "sgd", recs, BASE_GPU, outliers=OUTLIERS if arith == "b3" else None)
    abi_calls.assert_tanet_trunk()


def test_engine_equals_standalone_hooks_on_gpu(tmp_path):
    """The batched engine (1 moments launch + injected backward) and the per-hook path (one reference
    style hook per layer) are the same mathematics: same losses, same gradients."""
    g = H.golden("tta3.npz")
    a = run_product_tta(g, "sgd", tmp_path, _dev(), None, use_engine=True)
    b = run_product_tta(g, "sgd", tmp_path, _dev(), None, use_engine=False)
    assert a[0]["loss_reg"] == pytest.approx(b[0]["loss_reg"], rel=1e-5)
    for name in a[0]["grads"]:
        ga, gb = a[0]["grads"][name], b[0]["grads"][name]
        assert (ga - gb).abs().max().item() <= 1e-2 * gb.abs().max().item() + 1e-9, name  # GPU conv/BN backward kernels are not run-to-run deterministic


def test_hip_path_refuses_cpu_tensors():
    from vitta_amd import _lib, ops
    with pytest.raises(_lib.VittaHipError):
        ops.moments(torch.randn(4, 3, 2, 2), "bn2d")


def test_graph_replay_equals_eager_on_gpu(tmp_path):
    """hipGraph replay of the adapt step / eval forward == eager launches (dropout disabled so both
    arms see the same arithmetic): same losses, same adapted logits, same EMA state after 4 videos."""
    import json
    import numpy as np
    from vitta_amd import data, tta
    g = H.golden("tta3.npz")
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    means = [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    vars_ = [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))]
    mp, vp = H.write_stat_files(str(tmp_path), means, vars_)
    args = H.tanet_args(tmp_path, clip_length=T, input_size=size, spatiotemp_mean_clean_file=mp,
                        spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=1e-4)
    tta_set = data.SyntheticVideoDataset(5, 2, T, size, 101, "tanet", seed0=700)
    eval_set = data.SyntheticVideoDataset(5, 1, T, size, 101, "tanet", seed0=700)

    def run(capture_at):
        model = H.build_tanet(101, T, 0)
        model.base_model.fc = nn.Identity()  # no dropout: eager and graph arms must agree exactly
        adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(_dev()), args)
        out = []
        for i in range(5):
            x = adapter.shape_tta_input(tta_set[i][0].unsqueeze(0).to(_dev()))
            ev = adapter.shape_eval_input(eval_set[i][0].unsqueeze(0).to(_dev()))
            if capture_at is not None and i == capture_at:
                adapter.capture_graphs(x, ev)
            if adapter._graph is None:
                adapter.set_adapt_mode()
            _, lr_, lc_ = adapter.adapt_step(x)
            lr_, lc_ = lr_.detach().clone(), lc_.detach().clone()
            if adapter._graph is None:
                adapter.close_hooks()
            logits = adapter.evaluate(ev).clone()
            if adapter._graph is None:
                adapter.add_hooks_back()
            out.append((lr_.item(), lc_.item(), logits.cpu()))
        ema = adapter.engine.ema_mean.cpu().clone()
        torch.cuda.synchronize()
        return out, ema

    eager, ema_e = run(None)
    eager2, ema_e2 = run(None)
    eager3, ema_e3 = run(None)
    graphed, ema_g = run(2)
    # two GPU runs are not bit-reproducible (atomic arrival order in the backward kernels) and Adam's sign-like first updates amplify
    # the round-off: the yardstick is the spread of EAGER runs.  Round 6: THREE of them, the largest pairwise spread per step -- one pair
    # is a noisy estimate of a chaotic spread (a full-suite run drew 5.4e-6 / 2.8e-4 where three single runs drew 5.7e-6 .. 2.1e-5 /
    # 2.7e-4 .. 8.8e-4, and the graph arm then sat just outside 4 x the small draw)
    runs = [(eager, ema_e), (eager2, ema_e2), (eager3, ema_e3)]
    pairs = [(0, 1), (0, 2), (1, 2)]
    floor_ema = max((runs[i][1] - runs[j][1]).abs().max().item() for i, j in pairs)
    print("eager-vs-eager spread: ema %.3e logits %.3e" % (floor_ema, max((runs[i][0][k][2] - runs[j][0][k][2]).abs().max().item()
                                                                         for i, j in pairs for k in range(5))))
    for k, ((a, b, c), (d, e, f)) in enumerate(zip(eager, graphed)):
        fl = max((runs[i][0][k][2] - runs[j][0][k][2]).abs().max().item() for i, j in pairs)
        fa = max(abs(runs[i][0][k][0] - runs[j][0][k][0]) / abs(a) for i, j in pairs)
        fb = max(abs(runs[i][0][k][1] - runs[j][0][k][1]) / max(abs(b), 1e-12) for i, j in pairs)
        assert a == pytest.approx(d, rel=max(1e-4, 6 * fa)) and b == pytest.approx(e, rel=max(5e-3, 6 * fb)), (k, fa, fb)
        assert (f - c).abs().max().item() <= max(6 * fl, 2e-3 * c.abs().max().item()), (k, fl)
    assert (ema_g - ema_e).abs().max().item() <= max(6 * floor_ema, 2e-5)


# ---------------------------------------------------------------------------------------------------
# Video Swin-B: LayerNorm hooks -> NHWC moments / injection kernels
# ---------------------------------------------------------------------------------------------------
def test_swin_forward_matches_reference_on_gpu():
    from vitta_amd.bns_utils import choose_layers
    from vitta_amd.norm_stats import ComputeNormStatsHook
    g = H.golden("swin_fwd.npz")
    model = H.build_swin(11, 0).to(_dev())
    lns = [m for _, m in choose_layers(model, [nn.LayerNorm])][1:]
    hooks = [ComputeNormStatsHook(m, clip_len=16, stat_type="spatiotemp", before_norm=False, batch_size=1) for m in lns]
    with torch.no_grad():
        vid, view = model(H.seeded_randn((1, 2, 3, 16, 112, 112), 31).to(_dev()))
    for h in hooks:
        h.close()
    assert_logits_close(view.cpu(), torch.from_numpy(g["view"]), 2e-3)
    torch.testing.assert_close(torch.cat([h.batch_mean for h in hooks]).cpu(), torch.from_numpy(g["means"]), rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(torch.cat([h.batch_var for h in hooks]).cpu(), torch.from_numpy(g["vars"]), rtol=5e-3, atol=1e-5)


@pytest.mark.parametrize("mode,use_engine,strict", [("sgd", True, False), ("adam", True, False), ("sgd", False, False), ("adam", True, True)])
def test_three_swin_tta_steps_match_reference_on_gpu(tmp_path, mode, use_engine, strict, abi_calls):
    """strict: the exact-fp32 kernels of the Swin path with NO outlier allowance -- every sampled tensor under its own
    bound on every step, as the TANet test does with arith = f32."""
    from test_swin_cpu import run_product_tta_swin
    g = H.golden("tta3_swin.npz")
    recs = run_product_tta_swin(g, mode, tmp_path, _dev(), None, use_engine=use_engine)
    check_tta_records(g, mode, recs, BASE_GPU, outliers=None if strict else OUTLIERS)
    abi_calls.assert_swin_kernels()
    if use_engine:  # the column sums of the LayerNorm sites leave in batches (forward statistics, d gamma / d beta)
        assert abi_calls.abi.get("vitta_colsum2_multi_f32", 0) > 0, abi_calls.abi


def test_swin_fused_attention_equals_composed_ops_on_gpu():
    """Whole Swin-B forward + backward: fused W-MSA kernel vs the composed matmul/softmax ops."""
    from vitta_amd import swin
    model = H.build_swin(11, 0).to(_dev())
    x = H.seeded_randn((1, 2, 3, 16, 112, 112), 31).to(_dev())
    outs = []
    for fused in (True, False):
        swin.FUSED_ATTENTION = fused
        model.zero_grad()
        vid, view = model(x)
        view.square().sum().backward()
        outs.append((view.detach().cpu(), model.backbone.layers[2].blocks[3].attn.qkv.weight.grad.cpu().clone(),
                     model.backbone.layers[0].blocks[1].attn.relative_position_bias_table.grad.cpu().clone()))
    swin.FUSED_ATTENTION = True
    assert_logits_close(outs[0][0], outs[1][0], 1e-4)
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item() + 1e-9


def test_segmented_graph_capture_equals_single_graph_on_gpu(tmp_path):
    """The data-parallel capture (forward | backward | optimizer segments, exchanges outside any graph) must
    replay the same step as the single-graph capture; exercised here on one GPU (exchanges are no-ops)."""
    import json
    import numpy as np
    from vitta_amd import data, tta
    g = H.golden("tta3.npz")
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    mp, vp = H.write_stat_files(str(tmp_path), [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    args = H.tanet_args(tmp_path, clip_length=T, input_size=size, spatiotemp_mean_clean_file=mp,
                        spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=1e-4)
    tta_set = data.SyntheticVideoDataset(5, 2, T, size, 101, "tanet", seed0=700)
    eval_set = data.SyntheticVideoDataset(5, 1, T, size, 101, "tanet", seed0=700)

    def run(segmented):
        model = H.build_tanet(101, T, 0)
        model.base_model.fc = nn.Identity()
        adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(_dev()), args)
        out = []
        for i in range(5):
            x = adapter.shape_tta_input(tta_set[i][0].unsqueeze(0).to(_dev()))
            ev = adapter.shape_eval_input(eval_set[i][0].unsqueeze(0).to(_dev()))
            if i == 2:
                adapter.capture_graphs(x, ev, segmented=segmented)
            if adapter._graph is None:
                adapter.set_adapt_mode()
            _, lr_, lc_ = adapter.adapt_step(x)
            lr_, lc_ = lr_.clone(), lc_.clone()
            if adapter._graph is None:
                adapter.close_hooks()
            logits = adapter.evaluate(ev).clone()
            if adapter._graph is None:
                adapter.add_hooks_back()
            out.append((lr_.item(), lc_.item(), logits.cpu()))
        assert ("seg_fwd" in adapter._graph) == segmented
        torch.cuda.synchronize()
        return out

    single, single2, seg = run(False), run(False), run(True)
    floor = max((c - c2).abs().max().item() for (_, _, c), (_, _, c2) in zip(single, single2))
    for (a, b, c), (d, e, f) in zip(single, seg):
        assert a == pytest.approx(d, rel=1e-4) and b == pytest.approx(e, rel=5e-3)
        assert (f - c).abs().max().item() <= max(6 * floor, 2e-3 * c.abs().max().item())


@pytest.mark.parametrize("bf16,sgd", [(False, False), (True, False), ("dense", False), (True, True)])
def test_swin_config5_shape_gpu_equals_cpu_oracle_path(tmp_path, bf16, sgd):
    """BASELINE config 5 shape in small: Video Swin-B, window (16,7,7), 4 views x 32 frames.  N = 784 tokens per
    window at the first stages: the CHUNKED W-MSA kernels (keys / queries walked in chunks of 400, online softmax)
    run there -- asserted below by spying on the C-ABI entry --, the single-pass kernels at the clamped later stages;
    LN-affine Adam step on the GPU (HIP statistics path) == the same step on the CPU with the oracle backend.
    bf16: the bf16-OPERAND attention kernels (ops.WMSA_BF16, BASELINE config 5's "bf16 MFMA W-MSA"; 784 tokens in one
    pass) against the same fp32 CPU path at the tolerance 8-bit operand mantissas allow through 24 blocks: statistics loss
    rel 1e-4, consistency loss rel 2e-3, sampled gradients 3e-2 of their maximum (dense: 1e-4 / 5e-3 / 5e-2).  "dense": the bf16-operand dense layers
    as well (ops.DENSE_BF16, vitta_gemm_nt_bf16w_f32): every matrix product of the backbone on bf16 operands.
    sgd (round 5): SGD over ALL parameters, the reference's default optimizer (corpus/basics.py:547-560) -- the relative-position
    tables train, and the bf16 attention produces their gradient itself (one-pass backward, binned in LDS): the gradients of two
    784-token stages' tables against the CPU path at the attention bound, and no fp32 relative-position attention launch."""
    import numpy as np
    from oracle.oracle_backend import OracleBackend
    from vitta_amd import data, scripts, tta
    from vitta_amd.bns_utils import choose_layers
    T, size, views = 32, 112, 4

    def build():
        m = H.build_swin(174, 0, window_size=(16, 7, 7), drop_path_rate=0.0)
        m.cls_head.dropout = None
        return m

    model = build()
    lns = [m for _, m in choose_layers(model, [nn.LayerNorm])][1:]
    g = torch.Generator().manual_seed(3)
    mp, vp = H.write_stat_files(str(tmp_path), [torch.randn(m.normalized_shape[0], generator=g).numpy() * 0.1 for m in lns],
                                [torch.rand(m.normalized_shape[0], generator=g).numpy() + 0.5 for m in lns])
    args = scripts.swin_ucf101_args([])
    args.dataset, args.num_classes, args.datatype = "somethingv2", 174, "synthetic"
    args.clip_length, args.n_augmented_views, args.window_size = T, views, (16, 7, 7)
    args.input_size, args.scale_size, args.workers, args.verbose, args.result_dir = size, size, 0, False, str(tmp_path)
    args.spatiotemp_mean_clean_file, args.spatiotemp_var_clean_file = mp, vp
    args.update_only_bn_affine, args.lr = (not sgd), 1e-4
    x = data.SyntheticVideoDataset(1, views, T, size, 174, "swin", seed0=40)[0][0].unsqueeze(0)
    res = {}
    from vitta_amd import _lib, ops
    extra = ["module.backbone.layers.0.blocks.1.attn.relative_position_bias_table",
             "module.backbone.layers.2.blocks.5.attn.relative_position_bias_table"] if sgd else []
    calls = {}
    _lib.CALL_COUNTS = calls if sgd else None
    L = _lib.lib()
    entry = "vitta_wmsa_rel_fwd_bf16_io" if bf16 else "vitta_wmsa_rel_fwd_f32"
    orig, seen = getattr(L, entry), []

    def spy(*a):
        seen.append(int(a[8]))  # tokens per window of this launch
        return orig(*a)

    setattr(L, entry, spy)
    old_flag, ops.WMSA_BF16 = ops.WMSA_BF16, bool(bf16)
    old_dense, ops.DENSE_BF16 = ops.DENSE_BF16, bf16 == "dense"
    try:
        for dev, backend in ((torch.device("cpu"), OracleBackend()), (_dev(), None)):
            adapter = tta.ViTTAAdapter(tta.SingleDeviceParallel(build()).to(dev), args, engine_backend=backend)
            adapter.set_adapt_mode()
            _, loss_reg, loss_consis = adapter.adapt_step(adapter.shape_tta_input(x.to(dev)))
            named = dict(adapter.model.named_parameters())
            res[dev.type] = (float(loss_reg), float(loss_consis),
                             named["module.backbone.layers.2.blocks.4.norm1.weight"].grad.cpu().clone(),
                             named["module.backbone.norm.bias"].grad.cpu().clone()) + tuple(named[k].grad.cpu().clone() for k in extra)
    finally:
        _lib.CALL_COUNTS = None
        setattr(L, entry, orig)
        ops.WMSA_BF16 = old_flag
        ops.DENSE_BF16 = old_dense
    assert seen.count(784) == 22, sorted(set(seen))  # stages 1-3 (2 + 2 + 18 blocks): N = 784 (fp32: the chunked kernels)
    if sgd:  # every 784-token attention backward of the GPU step on the bf16 kernels (22 blocks; the last stage's window is clamped to
        # the 4 x 4 plane at this input size and takes the dense-bias kernels), none on the fp32 relative-position kernels
        assert calls.get("vitta_wmsa_rel_bwd_bf16", 0) == 22 and "vitta_wmsa_rel_bwd_f32" not in calls, {k: v for k, v in calls.items() if "wmsa" in k}
    c, gdev = res["cpu"], res["cuda"]
    # measured (r2k): fp32 9e-8 / 2e-7 / 4e-6; bf16 attention 9e-8 / 3e-5 / 3e-3; bf16 attention + dense 2e-6 / 1.3e-4 / 8e-3
    r0, r1, rg = (1e-4, 5e-3, 5e-2) if bf16 == "dense" else (1e-4, 2e-3, 3e-2) if bf16 else (2e-5, 1e-3, 2e-2)
    print("config-5 shape, mode", bf16, "loss_reg rel", abs(gdev[0] - c[0]) / abs(c[0]), "loss_consis rel", abs(gdev[1] - c[1]) / max(abs(c[1]), 1e-12),
          "grad rel", [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(gdev[2:], c[2:])])
    assert gdev[0] == pytest.approx(c[0], rel=r0) and gdev[1] == pytest.approx(c[1], rel=r1, abs=1e-6)
    for a, b in zip(gdev[2:], c[2:]):
        assert (a - b).abs().max().item() <= rg * b.abs().max().item() + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("capture", [None, "single", "segmented", "split", "split_sgd", "single_sgd"])
def test_overlapped_evaluation_equals_sequential_order_on_gpu(tmp_path, capture):
    """ViTTAAdapter.step (video i adapted while video i-1 is evaluated on a second stream, the optimizer update waiting for both)
    == adapt(i-1); eval(i-1); adapt(i): same losses and the same evaluation logits, eager and as one hipGraph (with a
    forked branch), as the data-parallel segments, and (round 6) as SEPARATE graphs on two streams -- forward + backward |
    optimizer on the step's stream, the evaluation graph captured on and replayed from the side stream; *_sgd: SGD over all
    parameters (split: the trunk's trainable convolutions are re-packed once per step by a graph of its own that both passes wait for)."""
    import json
    import numpy as np
    from vitta_amd import data, tta
    g = H.golden("tta3.npz")
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    mp, vp = H.write_stat_files(str(tmp_path), [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    sgd = capture is not None and capture.endswith("_sgd")
    args = H.tanet_args(tmp_path, clip_length=T, input_size=size, spatiotemp_mean_clean_file=mp,
                        spatiotemp_var_clean_file=vp, update_only_bn_affine=not sgd, lr=5e-5 if sgd else 1e-4)
    # (SGD over all parameters of this random-init model is chaotic: two runs of the SAME schedule drift apart ~30x per step -- by the
    # sixth video their logits differ by 2 % of the maximum, and ONE pair of runs is a poor estimate of a floor that grows that fast: the
    # fifth step failed the 8 x floor bound once in a full-suite run --, so those variants stop after four: two replayed steps behind the capture)
    n = 4 if sgd else 6
    tta_set = data.SyntheticVideoDataset(n, 2, T, size, 101, "tanet", seed0=700)
    eval_set = data.SyntheticVideoDataset(n, 1, T, size, 101, "tanet", seed0=700)

    def make():
        model = H.build_tanet(101, T, 0)
        model.base_model.fc = nn.Identity()
        return tta.ViTTAAdapter(tta.SingleDeviceParallel(model).to(_dev()), args)

    def clips(adapter, i):
        return (adapter.shape_tta_input(tta_set[i][0].unsqueeze(0).to(_dev())),
                adapter.shape_eval_input(eval_set[i][0].unsqueeze(0).to(_dev())))

    def sequential():
        adapter = make()
        out = []
        for i in range(n):
            x, ev = clips(adapter, i)
            adapter.set_adapt_mode()
            _, lr_, lc_ = adapter.adapt_step(x)
            adapter.close_hooks()
            logits = adapter.evaluate(ev).clone()
            adapter.add_hooks_back()
            out.append((lr_.item(), lc_.item(), logits.cpu()))
        torch.cuda.synchronize()
        return out

    def overlapped():
        adapter = make()
        losses, logits, prev = [], [], None
        for i in range(n):
            x, ev = clips(adapter, i)
            if capture is not None and i == 2:
                adapter.capture_graphs(x, ev, segmented=capture == "segmented", overlap_eval=True, split=capture.startswith("split"))
            adapter.set_adapt_mode()
            (_, lr_, lc_), ev_out = adapter.step(x, prev)
            losses.append((lr_.item(), lc_.item()))
            if ev_out is not None:
                logits.append(ev_out.clone().cpu())
            prev = ev
        if capture is not None:
            assert "step" in adapter._graph and (adapter._graph["step"] is None) == (capture == "segmented")
            assert (adapter._graph["step"] == "split") == capture.startswith("split")
            assert ("pre" in adapter._graph) == (capture == "split_sgd")
        adapter.close_hooks()
        logits.append(adapter.evaluate(prev).clone().cpu())
        torch.cuda.synchronize()
        return [(a, b, c) for (a, b), c in zip(losses, logits)]

    seq, seq2, ovl = sequential(), sequential(), overlapped()
    floor = max((c - c2).abs().max().item() for (_, _, c), (_, _, c2) in zip(seq, seq2))
    assert len(ovl) == n
    # per-step floors = two SEQUENTIAL runs against each other (atomic arrival order differs between any two runs; under SGD over all
    # parameters every weight moves each step and two runs drift apart ~30x per step: measured 0, 0, 1e-6, 3e-6, 9e-5, 7e-4)
    for (a, b, c), (d, e, f), (a2, b2, _) in zip(seq, ovl, seq2):
        fa, fb = abs(a - a2) / abs(a), abs(b - b2) / max(abs(b), 1e-12)
        assert a == pytest.approx(d, rel=max(1e-4, (30 if sgd else 8) * fa)) and b == pytest.approx(e, rel=max(5e-3, (30 if sgd else 8) * fb)), (sgd, fa, fb, a, d, b, e)
        assert (f - c).abs().max().item() <= max((10 if sgd else 6) * floor, (1e-2 if sgd else 2e-3) * c.abs().max().item())


def test_tta_loop_overlapped_schedule_logs_the_same_run_on_gpu(tmp_path):
    """tta_standard on the GPU, 8 synthetic videos (three eager steps, then hipGraph replay): the overlapped
    schedule (default) and the sequential schedule log the same per-video losses and accuracies; every video is
    evaluated exactly once and the lines come out in video order."""
    import json
    import re
    import numpy as np
    from vitta_amd import tta
    g = H.golden("tta3.npz")
    cfg = json.loads(str(g["config"]))
    T, size = cfg["T"], cfg["size"]
    ch = g["src_channels"]
    offs = np.concatenate([[0], np.cumsum(ch)])
    mp, vp = H.write_stat_files(str(tmp_path), [g["src_means"][offs[i]:offs[i + 1]] for i in range(len(ch))],
                                [g["src_vars"][offs[i]:offs[i + 1]] for i in range(len(ch))])
    model = H.build_tanet(101, T, 0)
    model.base_model.fc = nn.Identity()  # no dropout: both schedules see the same arithmetic

    def run(overlap):
        a = H.tanet_args(tmp_path, clip_length=T, input_size=size, spatiotemp_mean_clean_file=mp,
                         spatiotemp_var_clean_file=vp, update_only_bn_affine=True, lr=1e-4, synthetic_n_videos=8,
                         synthetic_seed=700, verbose=True, overlap_eval=overlap)
        lines = []

        class Log:
            def debug(self, msg):
                lines.append(msg)
        res = tta.tta_standard(tta.SingleDeviceParallel(model).to(_dev()), torch.nn.CrossEntropyLoss().to(_dev()),
                               args=a, logger=Log(), writer=None)
        rows, order = {}, []
        for l in lines:
            m = re.match(r"TTA Epoch1: \[(\d+)/8\].*Loss reg ([\d.]+) .*Loss consis ([\d.]+) .*Prec@1 ([\d.]+) ", l)
            if m:
                rows[int(m.group(1))] = tuple(float(v) for v in m.groups()[1:])
                order.append(int(m.group(1)))
        assert order == list(range(8))
        return res, rows

    (acc_o, rows_o), (acc_s, rows_s), (acc_s2, rows_s2) = run(True), run(False), run(False)
    # two GPU runs of the SAME schedule differ (library atomics, amplified by Adam's first sign-like updates):
    # their spread is the yardstick, as in the graph-vs-eager tests above
    for i in range(8):
        print(i, rows_o[i], rows_s[i], rows_s2[i])
        for k, (rel, floor) in enumerate(((2e-3, 2e-4), (2e-2, 2e-4))):
            spread = abs(rows_s[i][k] - rows_s2[i][k])
            assert abs(rows_o[i][k] - rows_s[i][k]) <= max(4 * spread, rel * abs(rows_s[i][k]) + floor), (i, k)
    assert abs(acc_o[0] - acc_s[0]) <= max(abs(acc_s[0] - acc_s2[0]), 100.0 / 8 + 1e-6)


def test_swin_row_mapped_attention_equals_roll_and_partition_copies_on_gpu():
    """Whole Swin-B forward + backward (224^2 and a 112^2 input, whose stage-3 windows clamp): shift + window
    partition folded into the attention kernel's addressing (natural token order end to end) vs torch.roll +
    window_partition / window_reverse copies around the same kernel."""
    from vitta_amd import swin
    model = H.build_swin(11, 0).to(_dev())
    for size in (224, 112):
        x = H.seeded_randn((1, 2, 3, 16, size, size), 31).to(_dev())
        outs = []
        for mapped in (True, False):
            swin.FUSED_PARTITION = mapped
            model.zero_grad()
            vid, view = model(x)
            view.square().sum().backward()
            outs.append((view.detach().cpu(), model.backbone.layers[2].blocks[3].attn.qkv.weight.grad.cpu().clone(),
                         model.backbone.layers[0].blocks[1].attn.relative_position_bias_table.grad.cpu().clone(),
                         model.backbone.layers[1].blocks[1].norm1.weight.grad.cpu().clone(),
                         model.backbone.patch_embed.proj.weight.grad.cpu().clone()))
        swin.FUSED_PARTITION = True
        assert_logits_close(outs[0][0], outs[1][0], 1e-4)
        for a, b in zip(outs[0][1:], outs[1][1:]):
            assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item() + 1e-9


def test_device_prefetcher_uploads_one_batch_ahead_on_its_own_stream():
    """vitta_amd/prefetch.py (the reference's loop uploads with a blocking .cuda() in front of each step, corpus/basics.py:612-623):
    the batches arrive on the device unchanged and in order, whether the loop calls `ahead()` after issuing its step or not, from pinned
    and from pageable host memory; the copy stream is not the consumer's stream, the consumer's stream waits for the copy (a kernel
    launched right after `next()` reads the uploaded values); StopIteration at the end, also after an `ahead()` that found the loader empty."""
    from vitta_amd.prefetch import DevicePrefetcher
    d = _dev()
    g = torch.Generator().manual_seed(3)
    host = [(torch.randn(2, 8, 3, 64, 64, generator=g), torch.tensor([i, 2 * i])) for i in range(5)]
    host[1] = (host[1][0].pin_memory(), host[1][1].pin_memory())
    for use_ahead in (True, False):
        pf = DevicePrefetcher(iter(host), d)
        assert pf.stream != torch.cuda.current_stream(d)
        sums = []
        for i in range(5):
            x, y = next(pf)
            assert x.is_cuda and y.is_cuda
            sums.append((x.double().sum() + y.sum()))  # consumer kernels on the current stream, no host synchronisation in between
            if use_ahead:
                pf.ahead()
                assert len(pf.queue) == (1 if i < 4 else 0)
        with pytest.raises(StopIteration):
            next(pf)
        assert pf.uploads == 5
        want = [float(a.double().sum() + b.sum()) for a, b in host]
        assert [float(s) for s in sums] == want


def test_role_streams_stay_distinct_when_the_stream_pool_wraps():
    """vitta_amd/streams.py (round 6): torch.cuda.Stream() hands out a per-device pool of 32 streams round-robin, so late in a long-lived
    process two requests share a handle -- and with it everything this package keys per stream (split-K workspaces, arrival tickets).
    The evaluation side stream, the trunk's helper streams, the copy stream and torch's graph-capture stream must be pairwise distinct
    whatever was created in between (the full GPU suite once put the side stream ON the capture stream: NaN losses in video 7)."""
    from vitta_amd import streams
    d = _dev()
    burn = [torch.cuda.Stream(d) for _ in range(45)]  # wrap the pool
    got = [streams.role(d, "eval"), streams.role(d, "copy")] + streams.roles(d, "trunk_helper", 8)
    burn += [torch.cuda.Stream(d) for _ in range(45)]
    again = [streams.role(d, "eval"), streams.role(d, "copy")] + streams.roles(d, "trunk_helper", 8)
    assert [s.cuda_stream for s in got] == [s.cuda_stream for s in again]  # a role keeps its stream
    handles = [s.cuda_stream for s in got] + [torch.cuda.graphs.graph.default_capture_stream.cuda_stream,
                                              torch.cuda.default_stream(d).cuda_stream, torch.cuda.current_stream(d).cuda_stream]
    assert len(set(handles[:11])) == 11 and len(set(handles)) >= 12, handles
    x = torch.ones(8, device=d)
    with torch.cuda.graph(torch.cuda.CUDAGraph()):  # a capture after the fact still finds its stream off every role's
        y = x * 2
    assert torch.cuda.graphs.graph.default_capture_stream.cuda_stream not in {s.cuda_stream for s in got}
