"""Parity of the HIP kernels (through the C ABI) against the CPU oracle -- GPU box only."""
import pytest
import torch

import helpers as H
from oracle import vitta_oracle as O

pytestmark = pytest.mark.gpu

# fp32 tolerances (stated per SURVEY section 7 "Variance numerics")
RTOL_MEAN, ATOL_MEAN = 1e-5, 1e-6
RTOL_VAR = 1e-4


def _dev():
    return torch.device("cuda:0")


def _feat(shape, seed, chan_dim, offset_scale=3.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    c = shape[chan_dim]
    off = torch.randn(c, generator=g) * offset_scale
    sc = torch.rand(c, generator=g) * 2 + 0.25
    view = [1] * len(shape)
    view[chan_dim] = c
    return x * sc.view(view) + off.view(view)


BN2D_SHAPES = [
    (16, 8, 7, 7),       # plane 392 (vec path, HW odd)
    (16, 7, 7, 7),       # plane 343 -> scalar path
    (16, 256, 28, 28),   # C2 layer3.0 conv1/bn1 shape
    (16, 1024, 14, 14),  # C2 largest hooked tensor
    (16, 2048, 7, 7),
    (16, 512, 7, 7),
    (3, 5, 2, 3),        # tiny ragged
    (64, 64, 56, 56),    # big HW (one channel spans many chunks)
]


@pytest.mark.parametrize("shape", BN2D_SHAPES)
def test_moments_nchw_single(shape):
    from vitta_amd import ops
    x = _feat(shape, 1, 1)
    clip = 8 if shape[0] % 8 == 0 else shape[0]
    m_ref, v_ref = O.moments(x.double(), "bn2d", clip)
    m, v = ops.moments(x.to(_dev()), "bn2d")
    torch.testing.assert_close(m.cpu().double(), m_ref, rtol=RTOL_MEAN, atol=ATOL_MEAN)
    torch.testing.assert_close(v.cpu().double(), v_ref, rtol=RTOL_VAR, atol=1e-7)


LN_SHAPES = [(2, 8, 14, 14, 512), (2, 8, 7, 7, 1024), (2, 8, 7, 7, 2048), (4, 4, 7, 7, 16), (1, 2, 3, 3, 6),
             (2, 4, 7, 7, 96), (1, 3, 5, 5, 7)]


@pytest.mark.parametrize("shape", LN_SHAPES)
def test_moments_nhwc_single(shape):
    from vitta_amd import ops
    x = _feat(shape, 2, 4)
    m_ref, v_ref = O.moments(x.double(), "ln")
    m, v = ops.moments(x.to(_dev()), "ln")
    torch.testing.assert_close(m.cpu().double(), m_ref, rtol=RTOL_MEAN, atol=ATOL_MEAN)
    torch.testing.assert_close(v.cpu().double(), v_ref, rtol=RTOL_VAR, atol=1e-7)


def test_moments_large_mean_small_var():
    """|mean| >> sigma: one-pass sum/sum-of-squares in fp32 would lose the variance."""
    from vitta_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 64, 14, 14, generator=g) * 1e-2 + 100.0
    m_ref, v_ref = O.moments(x.double(), "bn2d", 8)
    m, v = ops.moments(x.to(_dev()), "bn2d")
    torch.testing.assert_close(m.cpu().double(), m_ref, rtol=1e-6, atol=0)
    torch.testing.assert_close(v.cpu().double(), v_ref, rtol=1e-3, atol=0)


@pytest.mark.parametrize("shape", BN2D_SHAPES[:7])
def test_moments_nchw_single_bf16(shape):
    """bfloat16 features (`_bf16` input variants with fp32 accumulate): the result is the fp32 path's on the
    widened values -- checked against the oracle on exactly those values, same tolerances as fp32."""
    from vitta_amd import ops
    xb = _feat(shape, 1, 1).to(torch.bfloat16)
    clip = 8 if shape[0] % 8 == 0 else shape[0]
    m_ref, v_ref = O.moments(xb.double(), "bn2d", clip)
    m, v = ops.moments(xb.to(_dev()), "bn2d")
    assert m.dtype == torch.float32 and v.dtype == torch.float32
    torch.testing.assert_close(m.cpu().double(), m_ref, rtol=RTOL_MEAN, atol=ATOL_MEAN)
    torch.testing.assert_close(v.cpu().double(), v_ref, rtol=RTOL_VAR, atol=1e-7)


@pytest.mark.parametrize("shape", LN_SHAPES)
def test_moments_nhwc_single_bf16(shape):
    from vitta_amd import ops
    xb = _feat(shape, 2, 4).to(torch.bfloat16)
    m_ref, v_ref = O.moments(xb.double(), "ln")
    m, v = ops.moments(xb.to(_dev()), "ln")
    torch.testing.assert_close(m.cpu().double(), m_ref, rtol=RTOL_MEAN, atol=ATOL_MEAN)
    torch.testing.assert_close(v.cpu().double(), v_ref, rtol=RTOL_VAR, atol=1e-7)


def test_moments_batched_bf16_equals_fp32_on_widened_values():
    """One batched launch over mixed layouts with bfloat16 features: same (cnt, s1, s2) as the fp32 launch on the
    widened copies -- bit for bit (same plan, same order of operations) -- and mixed element types are refused."""
    from vitta_amd import _lib, ops
    feats = _mixed_plan_inputs()
    shapes = [ops.feature_layout(f, k) for f, k in feats]
    plan = ops.StatPlan(shapes, _dev())
    fb = [f.to(torch.bfloat16).to(_dev()) for f, _ in feats]
    ff = [t.float() for t in fb]
    shift = torch.cat([O.moments(f.double(), k, 8)[0].float() + 0.1 for f, k in feats]).to(_dev())
    plan.moments(fb, shift)
    got = [t.clone() for t in (plan.cnt, plan.s1, plan.s2)]
    plan.moments(ff, shift)
    for a, b in zip(got, (plan.cnt, plan.s1, plan.s2)):
        assert torch.equal(a, b)
    with pytest.raises(_lib.VittaHipError):
        plan.moments([fb[0]] + ff[1:], shift)
    with pytest.raises(_lib.VittaHipError):
        ops.moments(fb[0].half(), "bn2d")


def _mixed_plan_inputs():
    feats = [(_feat((16, 256, 14, 14), 10, 1), "bn2d"), (_feat((16, 512, 7, 7), 11, 1), "bn2d"),
             (_feat((2, 8, 7, 7, 128), 12, 4), "ln"), (_feat((16, 6, 5, 5), 13, 1), "bn2d"),
             (_feat((2, 4, 7, 7, 10), 14, 4), "ln")]
    return feats


@pytest.mark.parametrize("with_shift", [False, True])
def test_moments_batched_mixed_layouts(with_shift):
    from vitta_amd import ops
    feats = _mixed_plan_inputs()
    shapes = [ops.feature_layout(f, k) for f, k in feats]
    plan = ops.StatPlan(shapes, _dev())
    dfeats = [f.to(_dev()) for f, _ in feats]
    refs = [O.moments(f.double(), k, 8) for f, k in feats]
    shift = None
    if with_shift:
        shift = torch.cat([r[0].float() + 0.1 for r in refs]).to(_dev())
    plan.moments(dfeats, shift)
    mean, var = plan.mean_var(shift)
    m_ref = torch.cat([r[0] for r in refs])
    v_ref = torch.cat([r[1] for r in refs])
    torch.testing.assert_close(mean.cpu().double(), m_ref, rtol=RTOL_MEAN, atol=ATOL_MEAN)
    # without a shift the additive form s2/n - (s1/n)^2 cancels in fp32: looser bound there
    torch.testing.assert_close(var.cpu().double(), v_ref, rtol=RTOL_VAR if with_shift else 2e-3, atol=1e-6)
    n_ref = torch.tensor([f.numel() / s[1] for (f, _), s in zip(feats, shapes)])
    torch.testing.assert_close(plan.cnt.cpu(), n_ref.float())


@pytest.mark.parametrize("reg_type", ["l1_loss", "mse_loss", "kld"])
def test_align_three_steps_vs_oracle(reg_type):
    """EMA recurrence (zero init) + loss + injected gradient over 3 steps, all layers batched."""
    from vitta_amd import ops
    momentum = 0.1
    base = _mixed_plan_inputs()
    kinds = [k for _, k in base]
    shapes = [ops.feature_layout(f, k) for f, k in base]
    plan = ops.StatPlan(shapes, _dev())
    g = torch.Generator().manual_seed(77)
    src = []
    for f, k in base:
        m, v = O.moments(f, k, 8)
        src.append((m + 0.3 * torch.randn(m.shape, generator=g), v * (1.0 + 0.5 * torch.rand(v.shape, generator=g))))
    hooks = [O.StatHookOracle(sm, sv, reg_type, momentum, k, 8) for (sm, sv), k in zip(src, kinds)]
    src_mean = torch.cat([s[0] for s in src]).to(_dev())
    src_var = torch.cat([s[1] for s in src]).to(_dev())
    ema_mean = torch.zeros_like(src_mean)
    ema_var = torch.zeros_like(src_var)
    for step in range(3):
        feats = [(_feat(tuple(f.shape), 100 + 10 * step + i, 1 if k == "bn2d" else 4)) for i, (f, k) in enumerate(base)]
        leaves = [f.clone().requires_grad_(True) for f in feats]
        r = [h(x) for h, x in zip(hooks, leaves)]
        total_ref = sum(r)
        total_ref.backward()
        dfeats = [f.to(_dev()) for f in feats]
        plan.moments(dfeats, src_mean)
        total, layer = plan.align(src_mean, ema_mean, ema_var, src_mean, src_var, momentum, reg_type)
        torch.testing.assert_close(layer.cpu(), torch.stack([t.detach() for t in r]), rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(total.cpu()[0], total_ref.detach(), rtol=2e-5, atol=1e-6)
        for i, h in enumerate(hooks):
            sl = plan.channel_slice(i)
            torch.testing.assert_close(ema_mean[sl].cpu(), h.mean_avg.avg.detach(), rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(ema_var[sl].cpu(), h.var_avg.avg.detach(), rtol=1e-4, atol=1e-7)
            gx = ops.stat_align_bwd(dfeats[i], None, kinds[i], plan.mu[sl], plan.coef_a[sl], plan.coef_b[sl])
            ref = leaves[i].grad
            scale = ref.abs().max().item()
            assert (gx.cpu() - ref).abs().max().item() <= 2e-4 * scale + 1e-12


def test_align_bwd_accumulates_and_scales():
    from vitta_amd import ops
    x = _feat((16, 32, 7, 7), 3, 1).to(_dev())
    gout = torch.randn_like(x)
    mu = torch.randn(32, device=_dev())
    a = torch.randn(32, device=_dev())
    b = torch.randn(32, device=_dev())
    gs = torch.tensor([0.5], device=_dev())
    ref = gout + 0.5 * (a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1) * (x - mu.view(1, -1, 1, 1)))
    out = ops.stat_align_bwd(x, gout, "bn2d", mu, a, b, gs)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    g2 = gout.clone()
    ops.stat_align_bwd(x, g2, "bn2d", mu, a, b, gs, out=g2)  # in place
    torch.testing.assert_close(g2, ref, rtol=1e-5, atol=1e-5)
    xl = _feat((2, 4, 7, 7, 24), 4, 4).to(_dev())
    gl = torch.randn_like(xl)
    mu, a, b = (torch.randn(24, device=_dev()) for _ in range(3))
    refl = gl + (a + b * (xl - mu))
    torch.testing.assert_close(ops.stat_align_bwd(xl, gl, "ln", mu, a, b), refl, rtol=1e-5, atol=1e-5)


def test_feature_moments_autograd():
    from vitta_amd import ops
    x = _feat((16, 24, 7, 7), 8, 1)
    xr = x.clone().requires_grad_(True)
    m, v = O.moments(xr, "bn2d", 8)
    wm, wv = torch.randn(24), torch.randn(24)
    ((m * wm).sum() + (v * wv).sum()).backward()
    xd = x.to(_dev()).requires_grad_(True)
    md, vd = ops.FeatureMoments.apply(xd, "bn2d")
    ((md * wm.to(_dev())).sum() + (vd * wv.to(_dev())).sum()).backward()
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("shape", [(1, 2, 101), (3, 4, 174), (2, 2, 400), (5, 3, 7)])
def test_pred_consis(shape):
    from vitta_amd import ops
    g = torch.Generator().manual_seed(9)
    z = torch.randn(shape, generator=g) * 3
    zr = z.clone().requires_grad_(True)
    ref = O.compute_pred_consis(zr)
    ref.backward()
    zd = z.to(_dev()).requires_grad_(True)
    out = ops.pred_consis(zd)
    out.backward()
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(zd.grad.cpu(), zr.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("dims", [(2, 8, 64, 14, 14), (1, 8, 16, 7, 7), (2, 16, 8, 28, 28), (1, 4, 5, 3, 3)])
def test_tam_kernels(dims):
    from vitta_amd import ops
    n, t, c, h, w = dims
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n * t, c, h, w, generator=g)
    gate = torch.rand(n, c, t, generator=g)
    kern = torch.softmax(torch.randn(n * c, 3, generator=g), -1)
    gout = torch.randn(n * t, c, h, w, generator=g)
    gp = torch.randn(n, c, t, generator=g)
    xr, gr, kr = (v.clone().requires_grad_(True) for v in (x, gate, kern))
    pool_ref = O.tam_pool(xr, t)
    out_ref = O.tam_aggregate(xr, gr, kr, t)
    ((out_ref * gout).sum() + (pool_ref * gp).sum()).backward()
    d = _dev()
    xd, gd, kd = (v.to(d).requires_grad_(True) for v in (x, gate, kern))
    pool = ops.TamPool.apply(xd, t)
    out = ops.TamAggregate.apply(xd, gd, kd, t)
    ((out * gout.to(d)).sum() + (pool * gp.to(d)).sum()).backward()
    torch.testing.assert_close(pool.cpu(), pool_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out.cpu(), out_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gd.grad.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(kd.grad.cpu(), kr.grad, rtol=1e-4, atol=1e-3)


def _wmsa_reference(qkv, bias, mask, scale, nh):
    """swin_transformer.py:144-168 in fp64 on the CPU (composed ops)."""
    B_, N, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.view(B_, N, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, nh, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, nh, N, N)
    attn = attn.softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B_, N, C)


@pytest.mark.parametrize("B_,N,nh,nW", [(4, 392, 4, 2), (2, 392, 2, None), (6, 98, 3, 3), (2, 8, 1, None),
                                        (3, 100, 2, None), (2, 400, 1, 2), (64, 392, 4, 64)])
def test_wmsa_fused_forward_backward(B_, N, nh, nW):
    from vitta_amd import ops
    g = torch.Generator().manual_seed(17)
    C = nh * 32
    qkv = torch.randn(B_, N, 3 * C, generator=g)
    bias = torch.randn(nh, N, N, generator=g) * 0.5
    mask = None
    if nW is not None:
        region = torch.randint(0, 3, (nW, N), generator=g).float()
        mask = torch.where(region.unsqueeze(1) != region.unsqueeze(2), torch.tensor(-100.0), torch.tensor(0.0))
    gout = torch.randn(B_, N, C, generator=g)
    scale = 32 ** -0.5
    qr, br = qkv.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = _wmsa_reference(qr, br, mask.double() if mask is not None else None, scale, nh)
    ref.backward(gout.double())
    d = _dev()
    qd, bd = qkv.to(d).requires_grad_(True), bias.to(d).requires_grad_(True)
    out = ops.WindowAttention.apply(qd, bd, mask.to(d) if mask is not None else None, scale, nh)
    out.backward(gout.to(d))
    assert ops.wmsa_supported(N, 32)
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=2e-5)
    sc = qr.grad.abs().max().item()
    assert (qd.grad.cpu().double() - qr.grad).abs().max().item() <= 2e-4 * sc
    sb = br.grad.abs().max().item()
    assert (bd.grad.cpu().double() - br.grad).abs().max().item() <= 2e-4 * sb + 1e-6


def test_wmsa_unsupported_shapes_are_reported():
    from vitta_amd import ops
    assert not ops.wmsa_supported(784, 32) and not ops.wmsa_supported(392, 64)


@pytest.mark.parametrize("ws,clamp,B,nh,shift", [((8, 7, 7), (8, 7, 7), 2, 4, True), ((8, 7, 7), (4, 7, 7), 3, 2, False),
                                               ((8, 7, 7), (8, 4, 4), 2, 2, True), ((2, 3, 3), (2, 3, 3), 5, 1, True),
                                               ((16, 7, 7), (16, 7, 7), 1, 2, True), ((16, 7, 7), (16, 7, 7), 1, 1, False),
                                               ((16, 7, 7), (9, 7, 7), 1, 2, True), ((16, 7, 7), (16, 7, 5), 1, 1, True)])
def test_wmsa_relative_table_variant(ws, clamp, B, nh, shift):
    """On-chip bias/mask variant == dense variant semantics: table[index[:N,:N]] and compute_mask.  The (16,7,7)
    cases (N = 784, 441, 560 tokens; table of 5239 rows) run the chunked kernels (keys / queries walked in chunks of
    400 through LDS, online softmax)."""
    from vitta_amd import ops, swin
    g = torch.Generator().manual_seed(23)
    N = clamp[0] * clamp[1] * clamp[2]
    C = nh * 32
    T = (2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1)
    table = torch.randn(T, nh, generator=g) * 0.5
    index = swin.relative_position_index(ws)[:N, :N]
    code, off = swin.relative_position_code(ws)
    nW = 4
    region = torch.randint(0, 4, (nW, N), generator=g, dtype=torch.int32) if shift else None
    mask = None
    if shift:
        mask = torch.where(region.unsqueeze(1) != region.unsqueeze(2), torch.tensor(-100.0), torch.tensor(0.0))
    B_ = B * nW
    qkv = torch.randn(B_, N, 3 * C, generator=g)
    gout = torch.randn(B_, N, C, generator=g)
    scale = 32 ** -0.5
    qr, tr = qkv.double().requires_grad_(True), table.double().requires_grad_(True)
    bias = tr[index.reshape(-1)].view(N, N, nh).permute(2, 0, 1)
    ref = _wmsa_reference(qr, bias, mask.double() if mask is not None else None, scale, nh)
    ref.backward(gout.double())
    d = _dev()
    qd, td = qkv.to(d).requires_grad_(True), table.to(d).requires_grad_(True)
    out = ops.WindowAttentionRel.apply(qd, td, code[:N].to(d), off, region.to(d) if shift else None, scale, nh)
    out.backward(gout.to(d))
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=2e-5)
    assert (qd.grad.cpu().double() - qr.grad).abs().max().item() <= 2e-4 * qr.grad.abs().max().item()
    assert (td.grad.cpu().double() - tr.grad).abs().max().item() <= 5e-4 * tr.grad.abs().max().item() + 1e-6


@pytest.mark.parametrize("ws,clamp,B,nh,shift,rowmap", [((8, 7, 7), (8, 7, 7), 1, 4, False, False), ((8, 7, 7), (8, 7, 7), 2, 4, True, False),
                                                        ((16, 7, 7), (16, 7, 7), 1, 4, True, False), ((16, 7, 7), (16, 7, 7), 1, 8, True, True),
                                                        ((16, 7, 7), (9, 7, 7), 1, 4, True, False), ((8, 7, 7), (4, 7, 7), 1, 8, False, True)])
@pytest.mark.parametrize("io16", [False, True])
@pytest.mark.parametrize("bwd_form", ["one", "two"])
def test_wmsa_bf16_operand_variant(ws, clamp, B, nh, shift, rowmap, io16, bwd_form, monkeypatch):
    """vitta_wmsa_rel_{fwd,bwd}_bf16_io (BASELINE config 5: bf16 MFMA W-MSA, window (16,7,7) = 784 tokens in ONE pass) against the
    fp64 composed reference evaluated on bf16-ROUNDED q (x scale), k, v.  Tolerances (bf16 operands, fp32 accumulation: the
    probabilities and dS are rounded to 8 bits of mantissa before their GEMMs): output 1e-2, gradients 3e-2 of the tensor's
    maximum; the fp32 kernels on the same inputs are held to 1e-4.  io16: the bf16 data flow's form -- qkv and the upstream
    gradient ARE bfloat16 tensors, the context and the qkv gradient come back as bfloat16 (one more rounding each, inside the
    same bounds)."""
    from vitta_amd import ops, swin
    # bwd_form: the one-pass backward kernel of round 5 (dQ, dK, dV from ONE evaluation of every score tile; the default wherever a
    # workgroup per (window, head) fills the chip) / the two-kernel form (dQ query-major, dK / dV key-major); same bounds
    monkeypatch.setenv("VITTA_WMSA_BF16_BWD", bwd_form)
    g = torch.Generator().manual_seed(29)
    N = clamp[0] * clamp[1] * clamp[2]
    C = nh * 32
    T = (2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1)
    table = torch.randn(T, nh, generator=g) * 0.5
    index = swin.relative_position_index(ws)[:N, :N]
    code, off = swin.relative_position_code(ws)
    nW = 4
    region = torch.randint(0, 4, (nW, N), generator=g, dtype=torch.int32) if shift else None
    mask = torch.where(region.unsqueeze(1) != region.unsqueeze(2), torch.tensor(-100.0), torch.tensor(0.0)) if shift else None
    B_ = B * nW
    scale = 32 ** -0.5
    qkv = torch.randn(B_, N, 3 * C, generator=g)
    gout = torch.randn(B_, N, C, generator=g)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    if io16:  # what the kernels are handed is already 2 bytes wide
        qkv, gout = rb(qkv), rb(gout)
    # reference on the values the kernel multiplies: q is scaled in fp32 and then rounded
    q, k, v = qkv.view(B_, N, 3, C).unbind(2)
    qkv_r = torch.stack([rb(q * scale) / scale, rb(k), rb(v)], 2).reshape(B_, N, 3 * C)
    qr = qkv_r.double().requires_grad_(True)
    bias = table.double()[index.reshape(-1)].view(N, N, nh).permute(2, 0, 1)
    ref = _wmsa_reference(qr, bias, mask.double() if mask is not None else None, scale, nh)
    ref.backward(rb(gout).double())
    d = _dev()
    rm = None
    qd_in, gd_in = qkv, gout
    if rowmap:  # natural token order: window w of a sample gathers its rows through the map
        perm = torch.stack([torch.randperm(nW * N, generator=g) for _ in range(1)])[0].view(nW, N).to(torch.int32)
        rm = perm.to(d)
        nat = torch.empty(B, nW * N, 3 * C)
        gnat = torch.empty(B, nW * N, C)
        for bb in range(B):
            for w in range(nW):
                nat[bb, perm[w].long()] = qkv[bb * nW + w]
                gnat[bb, perm[w].long()] = gout[bb * nW + w]
        qd_in, gd_in = nat, gnat
    old = ops.WMSA_BF16
    ops.WMSA_BF16 = True
    try:
        io_t = torch.bfloat16 if io16 else torch.float32
        qd = qd_in.to(d, io_t).requires_grad_(True)
        out = ops.WindowAttentionRel.apply(qd, table.to(d), code[:N].to(d), off, region.to(d) if shift else None, scale, nh, rm)
        out.backward(gd_in.to(d, io_t))
    finally:
        ops.WMSA_BF16 = old
    assert out.dtype == io_t and qd.grad.dtype == io_t
    o, gq = out.detach().float().cpu(), qd.grad.float().cpu()
    if rowmap:
        o = torch.stack([o[bb, perm[w].long()] for bb in range(B) for w in range(nW)])
        gq = torch.stack([gq[bb, perm[w].long()] for bb in range(B) for w in range(nW)])
    assert (o.double() - ref.detach()).abs().max().item() <= 1e-2 * ref.abs().max().item()
    gr = qr.grad.view(B_, N, 3, C)
    gk = gq.double().view(B_, N, 3, C)
    for sel, name in enumerate(("dq", "dk", "dv")):
        err = (gk[:, :, sel] - gr[:, :, sel]).abs().max().item()
        assert err <= 3e-2 * gr[:, :, sel].abs().max().item(), (name, err, gr[:, :, sel].abs().max().item())


@pytest.mark.parametrize("ws,clamp,nh,shift", [((16, 7, 7), (16, 7, 7), 4, True), ((8, 7, 7), (8, 7, 7), 4, False), ((16, 7, 7), (9, 7, 7), 8, True)])
@pytest.mark.parametrize("io16", [False, True])
def test_wmsa_bf16_relative_position_table_gradient(ws, clamp, nh, shift, io16):
    """A TRAINABLE relative-position table (SGD over all parameters, the reference's default optimizer: corpus/basics.py:547-560,
    swin_transformer.py:110-151) on the bf16-operand attention: the one-pass backward bins d bias = dS by relative position in LDS
    and ADDS the (window, head) pair's column to table.grad.  Against the fp64 composed reference on the bf16-rounded operands: 3e-2
    of the gradient's maximum (the bound of dq / dk / dv); a gradient already in table.grad stays (accumulation); the qkv gradient
    keeps its bound.  Until round 5 a trainable table silently took the fp32 kernels."""
    from vitta_amd import _lib, ops, swin
    g = torch.Generator().manual_seed(31)
    N = clamp[0] * clamp[1] * clamp[2]
    C = nh * 32
    T = (2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1)
    assert _lib.lib().vitta_wmsa_bf16_dtable_supported(N, 32, T) == 1
    table = torch.randn(T, nh, generator=g) * 0.5
    index = swin.relative_position_index(ws)[:N, :N]
    code, off = swin.relative_position_code(ws)
    nW, B = 4, 1
    region = torch.randint(0, 4, (nW, N), generator=g, dtype=torch.int32) if shift else None
    mask = torch.where(region.unsqueeze(1) != region.unsqueeze(2), torch.tensor(-100.0), torch.tensor(0.0)) if shift else None
    B_ = B * nW
    scale = 32 ** -0.5
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    qkv = torch.randn(B_, N, 3 * C, generator=g)
    gout = torch.randn(B_, N, C, generator=g)
    if io16:
        qkv, gout = rb(qkv), rb(gout)
    q, k, v = qkv.view(B_, N, 3, C).unbind(2)
    qkv_r = torch.stack([rb(q * scale) / scale, rb(k), rb(v)], 2).reshape(B_, N, 3 * C)
    qr = qkv_r.double().requires_grad_(True)
    tabd = table.double().requires_grad_(True)
    bias = tabd[index.reshape(-1)].view(N, N, nh).permute(2, 0, 1)
    ref = _wmsa_reference(qr, bias, mask.double() if mask is not None else None, scale, nh)
    ref.backward(rb(gout).double())
    d = _dev()
    old = ops.WMSA_BF16
    ops.WMSA_BF16 = True
    try:
        io_t = torch.bfloat16 if io16 else torch.float32
        qd = qkv.to(d, io_t).requires_grad_(True)
        td = table.to(d).requires_grad_(True)
        td.grad = torch.full_like(td, 0.25)  # live storage: the kernel adds into it
        calls = {}
        _lib.CALL_COUNTS = calls
        out = ops.WindowAttentionRel.apply(qd, td, code[:N].to(d), off, region.to(d) if shift else None, scale, nh, None)
        out.backward(gout.to(d, io_t))
        torch.cuda.synchronize()
    finally:
        ops.WMSA_BF16 = old
        _lib.CALL_COUNTS = None
    assert calls.get("vitta_wmsa_rel_bwd_bf16", 0) == 1 and "vitta_wmsa_rel_bwd_f32" not in calls, calls
    gt = td.grad.cpu().double() - 0.25
    err = (gt - tabd.grad).abs().max().item()
    assert err <= 3e-2 * tabd.grad.abs().max().item(), (err, tabd.grad.abs().max().item())
    # entries no (query, key) pair of the clamped window reaches stay untouched
    reached = torch.zeros(T, dtype=torch.bool)
    reached[index.reshape(-1)] = True
    assert (gt[~reached] == 0).all()
    gr = qr.grad.view(B_, N, 3, C)
    gk = qd.grad.float().cpu().double().view(B_, N, 3, C)
    for sel in range(3):
        assert (gk[:, :, sel] - gr[:, :, sel]).abs().max().item() <= 3e-2 * gr[:, :, sel].abs().max().item()


@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("shape", [(16, 64, 14, 14), (16, 512, 7, 7), (8, 24, 5, 4)])
def test_fused_bn_act_matches_torch(shape, relu, res):
    """FusedBNAct (no statistics site) == relu(batch_norm_eval(x) + residual), forward and all gradients."""
    import torch.nn.functional as F
    from vitta_amd import ops
    g = torch.Generator().manual_seed(31)
    c = shape[1]
    x = torch.randn(shape, generator=g)
    r = torch.randn(shape, generator=g) if res else None
    w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.3
    gout = torch.randn(shape, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (x, w, b)] + ([r.double().requires_grad_(True)] if res else [])
    y = F.batch_norm(leaves[0], rm.double(), rv.double(), leaves[1], leaves[2], False, 0.0, 1e-5)
    if res:
        y = y + leaves[3]
    if relu:
        y = torch.relu(y)
    y.backward(gout.double())
    d = _dev()
    dl = [t.to(d).requires_grad_(True) for t in (x, w, b)] + ([r.to(d).requires_grad_(True)] if res else [])
    z = ops.FusedBNAct.apply(dl[0], dl[1], dl[2], rm.to(d), rv.to(d), 1e-5, dl[3] if res else None, relu, None)
    z.backward(gout.to(d))
    torch.testing.assert_close(z.detach().cpu().double(), y.detach(), rtol=1e-5, atol=1e-5)
    for a, bref in zip(dl, leaves):
        ref = bref.grad
        assert (a.grad.cpu().double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-6


def test_fused_bn_statistics_site_equals_recorded_hooks():
    """Hooked BN layers: fused pass (moments in the BN kernel, injection in the BN backward) == recorded
    features + batched moments kernel + separate injection, over three engine steps."""
    import torch.nn as nn
    from vitta_amd import fused_bn
    from vitta_amd.fused_bn import bn_act
    from vitta_amd.norm_stats import CombineNormStatsRegHook_onereg, StatAlignEngine

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.b1 = nn.Conv2d(3, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16)
            self.c2, self.b2 = nn.Conv2d(16, 16, 1, bias=False), nn.BatchNorm2d(16)
            self.c3, self.b3 = nn.Conv2d(16, 8, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(8)

        def forward(self, x):
            a = bn_act(self.b1, self.c1(x), relu=True)
            b = bn_act(self.b2, self.c2(a), residual=a, relu=True)
            return bn_act(self.b3, self.c3(b), relu=False).mean((2, 3))

    def run(enabled):
        fused_bn.ENABLED = enabled
        torch.manual_seed(0)
        net = Tiny().to(_dev())
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.3)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.2)
        net.eval()
        engine = StatAlignEngine("l1_loss", 0.1)
        hooks = [CombineNormStatsRegHook_onereg(m, clip_len=8, spatiotemp_stats_clean_tuple=(torch.zeros(c), torch.ones(c)),
                                                reg_type="l1_loss", moving_avg=True, momentum=0.1,
                                                stat_type_list=["spatiotemp"], before_norm=False,
                                                if_sample_tta_aug_views=True, n_augmented_views=2, engine=engine)
                 for m, c in ((net.b1, 16), (net.b2, 16), (net.b3, 8))]
        out = []
        for step in range(3):
            x = H.seeded_randn((16, 3, 12, 12), 50 + step).to(_dev())
            net.zero_grad()
            y = net(x)
            loss_reg = engine.finish()
            (loss_reg + 0.1 * y.square().sum()).backward()
            out.append((loss_reg.item(), [p.grad.detach().cpu().clone() for p in net.parameters()],
                        engine.ema_var.cpu().clone()))
        fused_bn.ENABLED = True
        return out

    fused, plain = run(True), run(False)
    for (la, ga, ea), (lb, gb, eb) in zip(fused, plain):
        assert la == pytest.approx(lb, rel=1e-5)
        torch.testing.assert_close(ea, eb, rtol=1e-4, atol=1e-7)
        for a, b in zip(ga, gb):
            assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-7


@pytest.mark.parametrize("name", ["c64_t8", "c16_t16"])
def test_tam_module_on_gpu_matches_reference_golden(name):
    """Whole TAM on the GPU (pool + fused G/L branches + aggregation kernels) vs the reference's TAM."""
    from test_oracle_golden import _golden_tam
    g, tam, (c, t, n, hw) = _golden_tam(name)
    tam = tam.to(_dev())
    x = H.seeded_randn((n * t, c, hw, hw), 9).to(_dev()).requires_grad_(True)
    gout = H.seeded_randn((n * t, c, hw, hw), 10).to(_dev())
    y = tam(x)
    grads = torch.autograd.grad(y, [x] + list(tam.parameters()), gout)
    torch.testing.assert_close(y.detach().cpu(), torch.from_numpy(g[f"{name}_y"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(grads[0].cpu(), torch.from_numpy(g[f"{name}_gx"]), rtol=1e-3, atol=1e-5)
    for (pn, _), gp in zip(tam.named_parameters(), grads[1:]):
        ref = torch.from_numpy(g[f"{name}_g_{pn}"])
        assert (gp.cpu() - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-6, pn


@pytest.mark.parametrize("c,t,n,hw", [(64, 8, 2, 14), (512, 8, 2, 7), (256, 16, 1, 7), (128, 8, 3, 5)])
def test_tam_fused_branches_equal_torch_modules(c, t, n, hw):
    import torch.nn as nn
    from vitta_amd import tanet
    torch.manual_seed(5)
    tam = tanet.TAM(c, t).to(_dev())
    with torch.no_grad():
        for m in tam.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    tam.eval()
    x0 = H.seeded_randn((n * t, c, hw, hw), 3).to(_dev())
    gout = H.seeded_randn((n * t, c, hw, hw), 4).to(_dev())
    res = []
    for fused in (True, False):
        tanet.FUSED_TAM_BRANCHES = fused
        x = x0.clone().requires_grad_(True)
        y = tam(x)
        res.append((y.detach(), torch.autograd.grad(y, [x] + list(tam.parameters()), gout)))
    tanet.FUSED_TAM_BRANCHES = True
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-5)
    names = ["x"] + [k for k, _ in tam.named_parameters()]
    for nm, a, b in zip(names, res[0][1], res[1][1]):
        assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item() + 1e-6, nm


def test_parameter_gradients_accumulate_into_live_grad_buffers():
    """With a live `.grad` (the views of tta.FlatArena) the backward kernels add into it and hand autograd None;
    the result must equal what AccumulateGrad would have produced: old grad + new grad."""
    import torch.nn as nn
    from vitta_amd import ops, tanet
    from vitta_amd.fused_bn import bn_act
    torch.manual_seed(11)
    c, t, n, hw = 64, 8, 2, 7
    tam = tanet.TAM(c, t).to(_dev()).eval()
    bn = nn.BatchNorm2d(c).to(_dev()).eval()
    with torch.no_grad():
        for m in list(tam.modules()) + [bn]:
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    params = list(tam.parameters()) + list(bn.parameters())
    x0 = H.seeded_randn((n * t, c, hw, hw), 3).to(_dev())
    gout = H.seeded_randn((n * t, c, hw, hw), 4).to(_dev())

    def run(direct, preset):
        ops.DIRECT_PARAM_GRAD = direct
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, 0.25 * (i + 1)) if preset else None
        x = x0.clone().requires_grad_(True)
        y = bn_act(bn, tam(x), relu=True)
        y.backward(gout)
        ops.DIRECT_PARAM_GRAD = True
        return [p.grad.clone() for p in params], x.grad.clone()

    (plain, gx0), (direct, gx1) = run(False, False), run(True, True)
    torch.testing.assert_close(gx0, gx1, rtol=1e-5, atol=1e-6)
    for i, (a, b) in enumerate(zip(direct, plain)):
        want = b + 0.25 * (i + 1)
        assert (a - want).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-5, i
    again, _ = run(True, False)  # no live buffer -> ordinary autograd outputs
    for a, b in zip(again, plain):
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-6


@pytest.mark.parametrize("kind", ["adam", "sgd", "sgd_plain"])
def test_flat_optimizers_match_torch_optim(kind):
    """vitta_adam_step_f32 / vitta_sgd_step_f32 on the arena == torch.optim.Adam / SGD (corpus/basics.py:547-560)
    stepping the same tensors, five steps with fresh gradients; arena gaps stay zero."""
    from vitta_amd import optim, tta
    torch.manual_seed(3)
    shapes = [(64,), (64,), (3, 5, 7), (1000,), (17,)]
    mine = [torch.nn.Parameter(torch.randn(s, device=_dev())) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone().double()) for p in mine]
    arena = tta.FlatArena(mine)
    if kind == "adam":
        opt = optim.FlatAdam(arena, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
        ropt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
    else:
        mu, wd = (0.9, 5e-4) if kind == "sgd" else (0.0, 0.0)
        opt = optim.FlatSGD(arena, lr=1e-2, momentum=mu, weight_decay=wd)
        ropt = torch.optim.SGD(ref, lr=1e-2, momentum=mu, weight_decay=wd)
    guard = None
    for step in range(5):
        if kind == "adam" and step == 2:
            #  a replaced / loaded one-element step tensor: the launch's arrival counter is a word of the optimizer's own,
            # nothing is written behind the step (the word after it keeps its sentinel)
            guard = torch.tensor([float(opt.state["step"]), 777.0], device=_dev())
            opt.state["step"] = guard[:1]
        opt.zero_grad()
        for p, r in zip(mine, ref):
            gr = torch.randn(p.shape, device=_dev())
            p.grad.copy_(gr)
            r.grad = gr.double()
        opt.step()
        ropt.step()
        for p, r in zip(mine, ref):
            assert (p.detach().double() - r.detach()).abs().max().item() <= 2e-6 * max(1.0, r.abs().max().item()), (kind, step)
    used = torch.zeros_like(arena.flat_param, dtype=torch.bool)
    for p in mine:
        off = (p.data_ptr() - arena.flat_param.data_ptr()) // 4
        used[off:off + p.numel()] = True
    assert float(arena.flat_param.detach()[~used].abs().max()) == 0.0
    if kind == "adam":
        assert float(opt.state["step"]) == 5.0 and float(guard[1]) == 777.0 and int(opt._ticket) == 0


def test_fused_bn_act_forked_output_sums_both_gradients():
    """bn_act(..., fork=True): the twin handle shares the storage of z, and the two upstream gradients (conv path,
    identity path of the next block) are summed inside the backward kernel; one of them may be absent."""
    import torch.nn.functional as F
    from vitta_amd import fused_bn
    from vitta_amd.fused_bn import bn_act, identity_source
    torch.manual_seed(8)
    shape = (8, 32, 7, 7)
    bn = torch.nn.BatchNorm2d(32).to(_dev()).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
    x0, r0 = torch.randn(shape, device=_dev()), torch.randn(shape, device=_dev())
    g1, g2 = torch.randn(shape, device=_dev()), torch.randn(shape, device=_dev())
    for use_twin in (True, False):
        res = []
        for fused in (True, False):
            fused_bn.ENABLED = fused
            x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            bn.zero_grad()
            z = bn_act(bn, x, residual=r, relu=True, fork=True)
            t = identity_source(z)
            assert (t is not z) == fused and t.data_ptr() == z.data_ptr()
            loss = (z * g1).sum() + ((t * g2).sum() if use_twin else 0.0)
            loss.backward()
            res.append((z.detach().clone(), x.grad.clone(), r.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
        fused_bn.ENABLED = True
        for a, b in zip(*res):
            assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-6


def test_residual_drop_path_one_pass_equals_torch_ops():
    """ops.ResidualDropPath (x + scale_b * branch, vitta_scale_add_f32) == x + branch * mask of timm's DropPath,
    forward and both gradients, with and without a scale; and the batched mask draw feeds each module's two uses."""
    from vitta_amd import ops, swin
    torch.manual_seed(4)
    x0 = torch.randn(3, 4, 7, 7, 32, device=_dev())
    b0 = torch.randn_like(x0)
    gout = torch.randn_like(x0)
    for scale in (torch.tensor([0.0, 1.25, 1.25], device=_dev()), None):
        x, b = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = ops.ResidualDropPath.apply(x, b, scale)
        out.backward(gout)
        xr, br = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        ref = xr + (br * scale.view(3, 1, 1, 1, 1) if scale is not None else br)
        ref.backward(gout)
        torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(x.grad, xr.grad, rtol=0, atol=0)
        torch.testing.assert_close(b.grad, br.grad, rtol=1e-6, atol=1e-6)
    mods = [swin.DropPath(0.2).train(), swin.DropPath(0.0).train(), swin.DropPath(0.5).train()]
    swin.draw_drop_path_masks(mods, 64, _dev())
    assert len(mods[0]._next) == 2 and len(mods[2]._next) == 2 and not getattr(mods[1], "_next", None)
    m0 = mods[0].sample(64, _dev())
    assert set(m0.unique().tolist()) <= {0.0, 1.25} and len(mods[0]._next) == 1
    m2 = mods[2].sample(64, _dev())
    assert set(m2.unique().tolist()) <= {0.0, 2.0}



@pytest.mark.parametrize("c", [128, 256, 512, 1024, 2048])
@pytest.mark.parametrize("with_branch", [False, True])
def test_fused_layernorm_matches_torch(c, with_branch):
    """ops.FusedLayerNorm (no statistics site) == F.layer_norm(x + scale_b * branch): outputs, saved x', and every
    gradient (x, branch, gamma, beta), including the residual path's own gradient arriving at x'."""
    import torch.nn.functional as F
    from vitta_amd import ops
    g = torch.Generator().manual_seed(c)
    shape = (3, 2, 5, 7, c)
    x, br = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    scale = torch.tensor([0.0, 1.25, 1.25])
    gy, gx2 = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (x, br, w, b)]
    xn = leaves[0] + leaves[1] * scale.double().view(3, 1, 1, 1, 1) if with_branch else leaves[0]
    y = F.layer_norm(xn, (c,), leaves[2], leaves[3], 1e-5)
    ((y * gy.double()).sum() + ((xn * gx2.double()).sum() if with_branch else 0.0)).backward()
    d = _dev()
    dl = [t.to(d).requires_grad_(True) for t in (x, br, w, b)]
    if with_branch:
        xn_d, y_d = ops.FusedLayerNorm.apply(dl[0], dl[1], scale.to(d), dl[2], dl[3], 1e-5, None)
        ((y_d * gy.to(d)).sum() + (xn_d * gx2.to(d)).sum()).backward()
        torch.testing.assert_close(xn_d.detach().cpu().double(), xn.detach(), rtol=1e-6, atol=1e-6)
    else:
        y_d = ops.FusedLayerNorm.apply(dl[0], None, None, dl[2], dl[3], 1e-5, None)
        (y_d * gy.to(d)).sum().backward()
    torch.testing.assert_close(y_d.detach().cpu().double(), y.detach(), rtol=1e-5, atol=1e-5)
    for i, (a, ref) in enumerate(zip(dl, leaves)):
        if i == 1 and not with_branch:
            continue
        assert (a.grad.cpu().double() - ref.grad).abs().max().item() <= 1e-4 * ref.grad.abs().max().item() + 1e-6, i


@pytest.mark.parametrize("c,nb,t,hw", [(64, 2, 8, 196), (256, 1, 8, 49), (32, 2, 4, 36)])
@pytest.mark.parametrize("relu,mask,rowadd,inject", [(True, False, True, "z"), (True, True, False, "raw"), (False, False, False, "raw"),
                                                      (True, False, True, "raw"), (True, False, False, None)])
def test_batchnorm_backward_on_channel_major_planes(c, nb, t, hw, relu, mask, rowadd, inject):
    """vitta_bn_bwd_cm_f32 (the element-wise piece between two data-gradient convolutions of the trunk: BatchNorm (+ReLU)
    backward with the TAM pooling gradient per (clip, channel, frame) row, the statistics-loss injection of a hooked layer --
    after the norm, or VITTA_BN_BWD_INJ_RAW for before_norm hooks (utils/norm_stats_utils.py:185) on its raw input -- and
    d gamma / d beta) against fp64 autograd of the composition it differentiates."""
    import ctypes as C
    import torch.nn.functional as F
    from vitta_amd import _lib
    from vitta_amd.ops import _ptr4
    g = torch.Generator().manual_seed(c + hw + t)
    n = nb * t
    P = n * hw
    x = torch.randn(c, P, generator=g)
    gam, bet = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rm, rv = torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5
    z0 = (x - rm[:, None]) * (gam * (rv + 1e-5).rsqrt())[:, None] + bet[:, None]
    x = torch.where(z0.abs() < 1e-3, x + 0.01, x)  # away from the ReLU kink
    other = torch.randn(c, P, generator=g)
    gy, g2 = torch.randn(c, P, generator=g), torch.randn(c, P, generator=g)
    radd = torch.randn(nb, c, t, generator=g)
    mu, ca, cb = torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 1e-2, torch.randn(c, generator=g) * 1e-2
    gscale = 0.6
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    z = (xd - rm.double()[:, None]) * (gd * (rv.double() + 1e-5).rsqrt())[:, None] + bd[:, None]
    a = (z * (other.double() > 0) if mask else torch.relu(z)) if relu else z
    gtot = gy.double() + g2.double()
    if rowadd:  # per (clip, channel, frame) constant / HW, the pooling path of TAM
        gtot = gtot + (radd.double() / hw).permute(1, 0, 2).reshape(c, n).repeat_interleave(hw, dim=1)
    loss = (a * gtot).sum()
    if inject:
        f = xd if inject == "raw" else z
        loss = loss + gscale * ((ca.double() - cb.double() * mu.double())[:, None] * f + 0.5 * cb.double()[:, None] * f * f).sum()
    loss.backward()
    d = _dev()
    dv = lambda v: v.to(d).contiguous()
    dx, gm = torch.full((c, P), float("nan"), device=d), torch.full((c, P), float("nan"), device=d)
    dgam, dbet = torch.full((c,), 0.5, device=d), torch.full((c,), -1.0, device=d)
    keep = [dv(v) for v in (gy, g2, x, other, radd, gam, bet, rm, rv, mu, ca, cb)]
    gs = torch.tensor([gscale], device=d)
    pp = lambda v: C.c_void_p(v.data_ptr())
    flag = int(relu) | (_lib.BN_BWD_INJ_RAW if inject == "raw" else 0)
    _lib.check(_lib.lib().vitta_bn_bwd_cm_f32(pp(keep[0]), pp(keep[1]), pp(keep[2]), pp(keep[3]) if mask else None, pp(keep[4]) if rowadd else None,
                                              1.0 / hw, _ptr4(*keep[5:9]), 1e-5, *( (pp(keep[9]), pp(keep[10]), pp(keep[11]), pp(gs)) if inject else (None,) * 4),
                                              flag, pp(dx), pp(gm), pp(dgam), pp(dbet), c, nb, t, hw, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "vitta_bn_bwd_cm_f32")
    assert (dx.cpu().double() - xd.grad).abs().max().item() <= 1e-5 * xd.grad.abs().max().item()
    assert ((dgam - 0.5).cpu().double() - gd.grad).abs().max().item() <= 1e-4 * gd.grad.abs().max().item() + 1e-5
    assert ((dbet + 1.0).cpu().double() - bd.grad).abs().max().item() <= 1e-4 * bd.grad.abs().max().item() + 1e-5
    mref = gtot * (((other.double() > 0) if mask else (z.detach() > 0)) if relu else 1.0)
    assert (gm.cpu().double() - mref).abs().max().item() <= 1e-6 * mref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 8, 56, 56, 128), (1, 4, 28, 28, 256), (2, 2, 14, 14, 512), (1, 3, 6, 10, 4)])
def test_patch_gather_equals_the_cat_of_strided_slices_both_ways(shape, abi_calls):
    """swin.PatchGather on vitta_patch_gather_f32 (one launch each way) == torch.cat of the four strided slices of PatchMerging
    (swin_transformer.py:281-286) and its autograd gradient, bit for bit (pure data movement)."""
    from vitta_amd import swin
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(_dev()).requires_grad_(True)
    out = swin.PatchGather.apply(x)
    ref_in = x.detach().clone().requires_grad_(True)
    ref = torch.cat([ref_in[:, :, 0::2, 0::2], ref_in[:, :, 1::2, 0::2], ref_in[:, :, 0::2, 1::2], ref_in[:, :, 1::2, 1::2]], -1)
    assert torch.equal(out, ref)
    go = torch.randn(ref.shape, generator=g).to(_dev())
    out.backward(go)
    ref.backward(go)
    assert torch.equal(x.grad, ref_in.grad)
    assert abi_calls.abi.get("vitta_patch_gather_f32", 0) == 2


def test_batched_column_sums_equal_the_per_site_launches():
    """vitta_colsum2_multi_f32 (the deferred column sums of a pass's LayerNorm sites: one launch per 32 items) == one
    vitta_colsum2_f32 per item: 40 items (two launches), ragged partial counts, every supported width, outputs that already hold
    values, two items sharing one output pair, the optional count word -- and through ops.ColsumQueue."""
    from vitta_amd import _lib, ops
    from vitta_amd.ops import _p, _stream
    L, d = _lib.lib(), _dev()
    gen = torch.Generator().manual_seed(5)
    items, ref = [], []
    shared = None
    for i in range(40):
        c = (128, 256, 512, 1024, 2048)[i % 5]
        nb = (1, 7, 32, 33, 196, 1568)[i % 6]
        part = torch.randn(nb, 2, c, generator=gen).to(d)
        if i == 11 and shared is not None:
            a, b = shared
            a0, b0 = None, None
        else:
            a, b = torch.randn(c, generator=gen).to(d), torch.randn(c, generator=gen).to(d)
            a0, b0 = a.clone(), b.clone()
        if i == 6:
            shared = (a, b)
        cnt = torch.zeros(1, device=d) if i % 3 == 0 else None
        items.append((part, nb, c, a, b, cnt, float(i + 1)))
        ref.append((a0, b0))
    # expected: fp64 sums on top of the initial values (item 11 adds to item 6's outputs)
    exp, mag = {}, {}
    for i, (part, nb, c, a, b, cnt, cv) in enumerate(items):
        key = 6 if i == 11 else i
        a0, b0 = ref[key]
        ea, eb = exp.get(key, (a0.double().cpu(), b0.double().cpu()))
        exp[key] = (ea + part[:, 0].double().sum(0).cpu(), eb + part[:, 1].double().sum(0).cpu())
        mag[key] = mag.get(key, 1.0) + float(part.abs().sum(0).max())  # (every contribution to a shared output counts)
    q = ops.ColsumQueue()
    for it in items:
        q.add(*it)
    q.flush()
    assert not q.items
    torch.cuda.synchronize()
    for i, (part, nb, c, a, b, cnt, cv) in enumerate(items):
        if i == 11:
            continue
        ea, eb = exp[i]
        tol = 1e-5 * mag[i]
        assert (a.double().cpu() - ea).abs().max().item() <= tol and (b.double().cpu() - eb).abs().max().item() <= tol, i
        if cnt is not None:
            assert cnt.item() == cv
    # the per-site entry point on the same partial rows gives the same sums (another order of the atomic adds)
    part, nb, c = items[5][0], items[5][1], items[5][2]
    a1, b1 = torch.zeros(c, device=d), torch.zeros(c, device=d)
    a2, b2 = torch.zeros(c, device=d), torch.zeros(c, device=d)
    _lib.check(L.vitta_colsum2_f32(_p(part), nb, c, _p(a1), _p(b1), None, 0.0, _stream()), "colsum2")
    q.add(part, nb, c, a2, b2)
    q.flush()
    torch.cuda.synchronize()
    torch.testing.assert_close(a1, a2, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(b1, b2, rtol=1e-5, atol=1e-4)
    arr = (_lib.ColsumItem * 1)()
    assert L.vitta_colsum2_multi_f32(arr, 1, _stream()) != 0  # null pointers: VITTA_ERR_INVALID_ARG
    assert L.vitta_colsum2_multi_f32(arr, 0, _stream()) != 0


@pytest.mark.parametrize("c", [128, 512, 1024])
def test_fused_layernorm_passthrough_joins_the_two_gradients_of_its_input(c):
    """FusedLayerNorm(..., passthrough=True) -> (x, y): the returned x stands for the block input's second reader (the residual
    update), so the gradient of x is gx(LayerNorm) + g(second reader) produced inside the backward pass -- equal to autograd's own
    accumulation over the two consumers; with y unused only the second reader's gradient comes back."""
    import torch.nn.functional as F
    from vitta_amd import ops
    g = torch.Generator().manual_seed(c + 1)
    shape = (2, 3, 5, 7, c)
    x = torch.randn(shape, generator=g)
    w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    gy, gx2 = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (x, w, b)]
    y = F.layer_norm(leaves[0], (c,), leaves[1], leaves[2], 1e-5)
    ((y * gy.double()).sum() + (leaves[0] * gx2.double()).sum()).backward()
    d = _dev()
    dl = [t.to(d).requires_grad_(True) for t in (x, w, b)]
    xa, y_d = ops.FusedLayerNorm.apply(dl[0], None, None, dl[1], dl[2], 1e-5, None, False, True)
    assert xa.data_ptr() == dl[0].data_ptr() and torch.equal(xa, dl[0])
    ((y_d * gy.to(d)).sum() + (xa * gx2.to(d)).sum()).backward()
    torch.testing.assert_close(y_d.detach().cpu().double(), y.detach(), rtol=1e-5, atol=1e-5)
    for i, (a_, ref) in enumerate(zip(dl, leaves)):
        assert (a_.grad.cpu().double() - ref.grad).abs().max().item() <= 1e-4 * ref.grad.abs().max().item() + 1e-6, i
    x2 = x.to(d).requires_grad_(True)
    xa, _ = ops.FusedLayerNorm.apply(x2, None, None, dl[1], dl[2], 1e-5, None, False, True)
    (xa * gx2.to(d)).sum().backward()
    assert torch.equal(x2.grad.cpu(), gx2)


@pytest.mark.parametrize("c", [128, 512, 1024])
@pytest.mark.parametrize("with_branch", [False, True])
def test_fused_layernorm_with_bfloat16_sides(c, with_branch):
    """The bf16 data flow's LayerNorm (vitta_ln_fwd_mixed / _bwd_mixed): y written as bfloat16, a bfloat16 branch, a bfloat16
    incoming gradient, a bfloat16 branch gradient -- against fp64 F.layer_norm evaluated on the SAME rounded inputs: y is the
    fp64 result rounded once (half a bf16 ulp), x' and d x, d gamma, d beta are fp32-accurate, d branch is d x' rounded once."""
    import torch.nn.functional as F
    from vitta_amd import ops
    g = torch.Generator().manual_seed(c + 7)
    shape = (3, 2, 5, 7, c)
    bf = lambda t: t.to(torch.bfloat16)
    x, br = torch.randn(shape, generator=g), bf(torch.randn(shape, generator=g))
    w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    scale = torch.tensor([0.0, 1.25, 1.25])
    gy, gx2 = bf(torch.randn(shape, generator=g)), torch.randn(shape, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (x, br.float(), w, b)]
    xn = leaves[0] + leaves[1] * scale.double().view(3, 1, 1, 1, 1) if with_branch else leaves[0]
    y = F.layer_norm(xn, (c,), leaves[2], leaves[3], 1e-5)
    ((y * gy.double()).sum() + ((xn * gx2.double()).sum() if with_branch else 0.0)).backward()
    d = _dev()
    dl = [x.to(d).requires_grad_(True), br.to(d).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)]
    if with_branch:
        xn_d, y_d = ops.FusedLayerNorm.apply(dl[0], dl[1], scale.to(d), dl[2], dl[3], 1e-5, None, True)
        assert xn_d.dtype == torch.float32
        torch.autograd.backward([y_d, xn_d], [gy.to(d), gx2.to(d)])
        torch.testing.assert_close(xn_d.detach().cpu().double(), xn.detach(), rtol=1e-6, atol=1e-6)
        assert dl[1].grad.dtype == torch.bfloat16
    else:
        y_d = ops.FusedLayerNorm.apply(dl[0], None, None, dl[2], dl[3], 1e-5, None, True)
        y_d.backward(gy.to(d))
    assert y_d.dtype == torch.bfloat16
    assert (y_d.detach().cpu().double() - y.detach()).abs().max().item() <= 2.0 ** -8 * y.detach().abs().max().item() + 1e-5
    for i, (a, ref) in enumerate(zip(dl, leaves)):
        if i == 1 and not with_branch:
            continue
        tol = 2.0 ** -8 if i == 1 else 1e-4   # the branch gradient is bfloat16
        assert (a.grad.cpu().double() - ref.grad).abs().max().item() <= tol * ref.grad.abs().max().item() + 1e-6, i


@pytest.mark.parametrize("shape", [(4, 64, 112, 112), (3, 16, 15, 23), (2, 8, 8, 8)])
def test_fused_stem_pool_matches_torch(shape):
    """ops.FusedStemPool == max_pool2d(relu(batch_norm_eval(x)), 3, 2, 1): output and the affine gradients (the
    max-pool scatter recomputed as a reduction), accumulated into live .grad storage; odd sizes exercise the padding."""
    import torch.nn.functional as F
    from vitta_amd import ops
    g = torch.Generator().manual_seed(shape[2])
    c = shape[1]
    x = torch.randn(shape, generator=g)
    w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rm, rv = torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.max_pool2d(torch.relu(F.batch_norm(x.double(), rm.double(), rv.double(), wd, bd, False, 0.0, 1e-5)), 3, 2, 1)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout.double())
    d = _dev()
    wg, bg = w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    wg.grad, bg.grad = torch.full_like(wg, 0.5), torch.full_like(bg, -0.25)  # live storage: the kernel adds into it
    out = ops.FusedStemPool.apply(x.to(d), wg, bg, rm.to(d), rv.to(d), 1e-5)
    out.backward(gout.to(d))
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=1e-5)
    assert (wg.grad.cpu().double() - 0.5 - wd.grad).abs().max().item() <= 1e-4 * wd.grad.abs().max().item() + 1e-5
    assert (bg.grad.cpu().double() + 0.25 - bd.grad).abs().max().item() <= 1e-4 * bd.grad.abs().max().item() + 1e-5


@pytest.mark.parametrize("c,t,n", [(64, 8, 2), (128, 8, 1), (256, 8, 2), (512, 8, 2), (512, 8, 3), (256, 16, 1), (96, 8, 2), (512, 16, 1),
                                   (320, 4, 2)])
def test_tam_branch_single_launch_forms_equal_the_two_launch_forms(c, t, n):
    """vitta_tam_branch_{fwd,bwd}_fused_f32 (F1 -> F2 / B1 -> B2 inside one launch, the clip's workgroups meeting on a
    device-scope counter) against the two-launch entry points: identical outputs (the arithmetic is the same code; only
    the atomically accumulated parameter gradients may differ in summation order), the counters back at zero, and the
    same again on a second and third launch.  Both also read a FRAME-major int64 fixed-point pooled tensor [N, T, C] (pooled_tc = 1:
    what a convolution's VITTA_CONV_POOL epilogue accumulates): same outputs as from the [N, C, T] form, bit for bit."""
    import ctypes as C
    from vitta_amd import _lib
    from vitta_amd.ops import _p, _ptr4, _stream
    L = _lib.lib()
    d = _dev()
    g = torch.Generator().manual_seed(c + t + n)
    o = c // 4
    r = lambda *s: torch.randn(*s, generator=g).to(d)
    pooled = torch.round(r(n, c, t) * 4096) / 4096  # (exactly representable with 32 fractional bits: the fixed-point form below is lossless)
    wg1, wg3, w0, w3 = r(2 * t, t) * 0.3, r(3, 2 * t) * 0.3, r(o, c, 3) * (3 * c) ** -0.5, r(c, o) * o ** -0.5
    bng = [torch.rand(2 * t, generator=g).to(d) + 0.5, r(2 * t) * 0.1, r(2 * t) * 0.1, torch.rand(2 * t, generator=g).to(d) + 0.5]
    bnl = [torch.rand(o, generator=g).to(d) + 0.5, r(o) * 0.1, r(o) * 0.1, torch.rand(o, generator=g).to(d) + 0.5]
    gkern, ggate = r(n * c, 3), r(n, c, t)
    sync = torch.zeros(256, dtype=torch.int32, device=d)

    pooled_tc = torch.round(pooled.permute(0, 2, 1).double() * 2.0 ** 32).to(torch.int64).contiguous()  # [N, T, C] fixed point

    def run(fused, tc=0):
        kern, gate, hpre = torch.empty(n * c, 3, device=d), torch.empty(n, c, t, device=d), torch.empty(2, n, o, t, device=d)
        args = (_p(pooled_tc if tc else pooled), _p(wg1), _ptr4(*bng), 1e-5, _p(wg3), _p(w0), _ptr4(*bnl), 1e-5, _p(w3), n, c, t)
        if fused:
            _lib.check(L.vitta_tam_branch_fwd_fused_f32(*args, _p(kern), _p(gate), _p(hpre), _p(sync), tc, _stream()), "fwd fused")
        else:
            _lib.check(L.vitta_tam_branch_fwd_f32(*args, _p(kern), _p(gate), _p(hpre), tc, _stream()), "fwd")
        gbuf = torch.empty(n * c * t + n * o * t, device=d)
        dbn = [torch.zeros(2 * t, device=d), torch.zeros(2 * t, device=d), torch.zeros(o, device=d), torch.zeros(o, device=d)]
        dw = [torch.zeros_like(wg1), torch.zeros_like(wg3), torch.zeros_like(w0), torch.zeros_like(w3)]
        bargs = args + (_p(kern), _p(gate), _p(hpre), _p(gkern), _p(ggate), _p(gbuf), _ptr4(*dbn), _ptr4(*dw))
        if fused:
            _lib.check(L.vitta_tam_branch_bwd_fused_f32(*bargs, _p(sync), tc, _stream()), "bwd fused")
        else:
            _lib.check(L.vitta_tam_branch_bwd_f32(*bargs, tc, _stream()), "bwd")
        torch.cuda.synchronize()
        return kern, gate, hpre, gbuf[:n * c * t].clone(), dbn, dw

    # round 5: shapes with C % 64 == 0, C <= 512, T % 4 == 0, C * T <= 4096 take the one-batch kernels (every operand of a workgroup
    # requested at once; a growing arrival counter + a generation word in the SECOND half of the meeting buffer instead of the
    # zero-at-rest pair); (96, 8), (512, 16) stay on the first fused kernels
    fast = c % 64 == 0 and c <= 512 and t % 4 == 0 and c * t <= 4096
    ref = run(False)
    for rep in range(3):
        got = run(True)
        assert int(sync[:128].abs().sum()) == 0, "meeting counters must be zero at rest"
        words = sync[128:128 + 2 * n].view(n, 2).cpu()
        if fast:  # arrivals == base at rest == the workgroups of a clip over all launches so far (forward C / 8, backward C / 16)
            assert (words[:, 1] == words[:, 0]).all() and (words[:, 0] == (rep + 1) * (c // 8 + c // 16)).all(), words
        else:
            assert int(words.abs().sum()) == 0
        for a, b in zip(got[:4], ref[:4]):
            assert torch.equal(a, b), rep
        for a, b in zip(got[4] + got[5], ref[4] + ref[5]):
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-7
    for fused in (False, True):
        got = run(fused, tc=1)
        for a, b in zip(got[:4], ref[:4]):
            assert torch.equal(a, b), ("frame-major pooled", fused)
        for a, b in zip(got[4] + got[5], ref[4] + ref[5]):
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-7


def test_tam_branch_launches_of_different_grids_share_one_meeting_buffer():
    """The trunk issues the fused branch launches of its sixteen blocks back to back on one stream: 64 -> 128 -> 256 -> 512 channels
    forward, the reverse order backward, all with the SAME per-stream meeting buffer.  The one-batch kernels wait for
    `arrivals so far + workgroups of THIS launch` (a growing counter, never reset): launches with different grids must not read each
    other's arrival counts (round 5's first form counted generations x workgroups and dead-locked at the first change of grid).
    Outputs equal the two-launch entry points bit for bit at every stage."""
    from vitta_amd import _lib
    from vitta_amd.ops import _p, _ptr4, _stream
    L = _lib.lib()
    d = _dev()
    g = torch.Generator().manual_seed(11)
    n, t = 2, 8
    r = lambda *s: torch.randn(*s, generator=g).to(d)
    sync = torch.zeros(256, dtype=torch.int32, device=d)
    stages = []
    for c in (64, 128, 256, 512, 256, 64):
        o = c // 4
        pooled = torch.round(r(n, c, t) * 4096) / 4096
        st = dict(c=c, o=o, pooled=pooled, wg1=r(2 * t, t) * 0.3, wg3=r(3, 2 * t) * 0.3, w0=r(o, c, 3) * (3 * c) ** -0.5, w3=r(c, o) * o ** -0.5,
                  bng=[torch.rand(2 * t, generator=g).to(d) + 0.5, r(2 * t) * 0.1, r(2 * t) * 0.1, torch.rand(2 * t, generator=g).to(d) + 0.5],
                  bnl=[torch.rand(o, generator=g).to(d) + 0.5, r(o) * 0.1, r(o) * 0.1, torch.rand(o, generator=g).to(d) + 0.5],
                  gkern=r(n * c, 3), ggate=r(n, c, t))
        stages.append(st)

    def run(fused):
        outs = []
        for st in stages:  # forward chain
            c, o = st["c"], st["o"]
            st["kern"], st["gate"], st["hpre"] = torch.empty(n * c, 3, device=d), torch.empty(n, c, t, device=d), torch.empty(2, n, o, t, device=d)
            args = (_p(st["pooled"]), _p(st["wg1"]), _ptr4(*st["bng"]), 1e-5, _p(st["wg3"]), _p(st["w0"]), _ptr4(*st["bnl"]), 1e-5, _p(st["w3"]), n, c, t)
            if fused:
                _lib.check(L.vitta_tam_branch_fwd_fused_f32(*args, _p(st["kern"]), _p(st["gate"]), _p(st["hpre"]), _p(sync), 0, _stream()), "fwd fused")
            else:
                _lib.check(L.vitta_tam_branch_fwd_f32(*args, _p(st["kern"]), _p(st["gate"]), _p(st["hpre"]), 0, _stream()), "fwd")
        for st in reversed(stages):  # backward chain
            c, o = st["c"], st["o"]
            gbuf = torch.empty(n * c * t + n * o * t, device=d)
            dbn = [torch.zeros(2 * t, device=d), torch.zeros(2 * t, device=d), torch.zeros(o, device=d), torch.zeros(o, device=d)]
            args = (_p(st["pooled"]), _p(st["wg1"]), _ptr4(*st["bng"]), 1e-5, _p(st["wg3"]), _p(st["w0"]), _ptr4(*st["bnl"]), 1e-5, _p(st["w3"]), n, c, t)
            bargs = args + (_p(st["kern"]), _p(st["gate"]), _p(st["hpre"]), _p(st["gkern"]), _p(st["ggate"]), _p(gbuf), _ptr4(*dbn), None)
            if fused:
                _lib.check(L.vitta_tam_branch_bwd_fused_f32(*bargs, _p(sync), 0, _stream()), "bwd fused")
            else:
                _lib.check(L.vitta_tam_branch_bwd_f32(*bargs, 0, _stream()), "bwd")
            outs.append((st["kern"].clone(), st["gate"].clone(), gbuf[:n * c * t].clone()))
        torch.cuda.synchronize()
        return outs

    ref = run(False)
    for rep in range(2):
        got = run(True)
        for a, b in zip(got, ref):
            for x, y in zip(a, b):
                assert torch.equal(x, y), rep
    words = sync[128:128 + 2 * n].view(n, 2).cpu()
    assert (words[:, 0] == words[:, 1]).all() and int(words[0, 0]) == 2 * sum(st["c"] // 8 + st["c"] // 16 for st in stages)
