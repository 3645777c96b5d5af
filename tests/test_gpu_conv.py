"""vitta_conv_f32 against fp64 torch CPU convolutions, in BOTH arithmetic forms (the autouse `arith` fixture): the shipped
split-bf16 kernel (vitta_amd/csrc/conv_b3.hip: every fp32 operand as three bf16 terms, six v_mfma_f32_32x32x16_bf16 products per
multiply-add, fp32 accumulation; round 5: the epilogue of contiguous outputs through an LDS turn-around) and the exact-fp32 MFMA
kernels (conv_sk.hip / conv_pw.hip / conv.hip) -- every geometry the TANet trunk uses (pointwise, 3x3, stride 2, their data
gradients, the parity-merged stride-2 data gradient) and every epilogue (eval BN / residual / ReLU / hooked moments / raw copy /
pooled means; BatchNorm backward with mask, residual, injection, d gamma / d beta).

Tolerance, the same for both forms: |err| <= 2e-5 * max|ref| against the fp64 reference for K up to 4608 terms per output (measured
~3e-7 for either form); element-wise bounds under 16 decades of dynamic range, large-mean inputs and non-finite inputs have their own
tests below."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(params=["b3", "f32"], autouse=True)
def arith(request):
    """Every test of this file runs in both arithmetic forms of vitta_conv_f32: split-bf16 operands on the bf16 matrix pipe
    (conv_b3.hip, the default) and exact-fp32 MFMA; same bounds."""
    from vitta_amd import conv as CV
    old, CV.ARITH = CV.ARITH, request.param
    yield request.param
    CV.ARITH = old


def _dev():
    return torch.device("cuda:0")


def _close(got, ref, tol=TOL, what=""):
    err = (got.detach().cpu().double() - ref).abs().max().item()
    bound = tol * max(ref.abs().max().item(), 1e-6)
    assert err <= bound, (what, err, bound)


def _bn(c, g):
    return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.2,
            torch.rand(c, generator=g) + 0.5)


def _bn_apply(x, bn, eps=1e-5):
    gam, bet, rm, rv = (t.double() for t in bn)
    return F.batch_norm(x, rm, rv, gam, bet, False, 0.0, eps)


@pytest.mark.parametrize("n,c,k,h,ksz,stride,tile", [
    (4, 64, 64, 56, 1, 1, 0), (2, 256, 64, 56, 1, 1, 0), (16, 64, 256, 56, 1, 1, 0), (8, 512, 128, 28, 1, 1, 0),
    (16, 1024, 256, 14, 1, 1, 0), (16, 2048, 512, 7, 1, 1, 0), (8, 512, 2048, 7, 1, 1, 0), (8, 48, 96, 10, 1, 1, 0),
    (4, 64, 64, 56, 3, 1, 0), (8, 128, 128, 28, 3, 1, 0), (16, 256, 256, 14, 3, 1, 0), (16, 512, 512, 7, 3, 1, 0),
    (4, 128, 128, 56, 3, 2, 0), (8, 256, 256, 28, 3, 2, 0), (8, 512, 512, 14, 3, 2, 0), (4, 256, 512, 56, 1, 2, 0),
    (8, 1024, 2048, 14, 1, 2, 0), (4, 64, 64, 13, 3, 1, 0), (4, 32, 32, 9, 3, 2, 0),
    (4, 128, 128, 28, 1, 1, (128 << 16) | 128), (4, 128, 128, 28, 3, 1, (128 << 16) | 64), (4, 128, 128, 28, 3, 1, (64 << 16) | 64),
    (4, 128, 128, 28, 1, 1, (64 << 16) | 32), (4, 128, 128, 28, 3, 1, (128 << 16) | 128),
    # the 64^2-input shapes of the golden step tests (planes of 16 / 8 / 4 / 2 pixels a side, 32 frames)
    (32, 64, 64, 16, 3, 1, 0), (32, 128, 128, 16, 3, 2, 0), (32, 256, 256, 8, 3, 2, 0), (32, 512, 512, 4, 3, 2, 0),
    (32, 512, 512, 2, 3, 1, 0), (32, 1024, 2048, 4, 1, 2, 0), (32, 2048, 512, 2, 1, 1, 0), (32, 256, 512, 16, 1, 2, 0)])
def test_forward_matches_fp64_conv(n, c, k, h, ksz, stride, tile):
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(n * 1000 + c + k + h + ksz)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, ksz, ksz, generator=g) * (c * ksz * ksz) ** -0.5
    pad = ksz // 2
    ref = F.conv2d(x.double(), w.double(), stride=stride, padding=pad)
    d = _dev()
    geom = CV.Geometry.forward(n, h, h, ksz, stride, pad)
    y = torch.full((k, n * geom.hy * geom.wy), float("nan"), device=d)
    CV.launch(geom, CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k, tile=tile)
    _close(CV.from_cm(y, n, geom.hy, geom.wy), ref)


@pytest.mark.parametrize("n,t,c,k,h,hooked,tile,ksplit", [
    (16, 8, 64, 64, 56, False, 0, 0), (16, 8, 256, 64, 28, True, 0, 0), (16, 8, 512, 128, 28, False, 0, 0), (16, 8, 1024, 256, 14, True, 0, 0),
    (16, 8, 2048, 512, 7, False, 0, 0), (24, 8, 2048, 512, 7, True, 0, 0), (8, 8, 1024, 256, 14, False, 0, 4), (32, 16, 256, 64, 8, False, 0, 0),
    (8, 4, 128, 128, 28, False, (64 << 16) | 64, 1), (8, 4, 64, 64, 13, False, 0, 0)])
def test_frame_pooled_means_from_the_epilogue(n, t, c, k, h, hooked, tile, ksplit):
    """VITTA_CONV_POOL: TAM's adaptive average pooling of relu(bn1(conv1(x))) (temporal_module.py:53) taken from conv1's
    accumulators -- per (frame, channel) means added as 64-bit fixed-point numbers into a zeroed frame-major [N, K] tensor -- against fp64; planes of 49 and 64
    pixels put two frames into one 32-pixel block, 13 x 13 planes leave ragged blocks, split K runs the epilogue once per tile;
    together with the hooked layer's moments and the raw output the launch also writes."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(n + c + k + h)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, 1, 1, generator=g) * c ** -0.5
    bn = _bn(k, g)
    raw = F.conv2d(x.double(), w.double())
    z = _bn_apply(raw, bn)
    ref = z.clamp_min(0).mean((2, 3))  # [frames, K]
    d = _dev()
    geom = CV.Geometry.forward(n, h, h, 1, 1, 0)
    y = torch.full((k, n * h * h), float("nan"), device=d)
    pooled = torch.zeros(n, k, dtype=torch.int64, device=d)  # fixed point, 32 fractional bits
    dec = lambda q: q.cpu().double() * 2.0 ** -32
    shift = z.mean((0, 2, 3)).float()
    stats = (shift.to(d), torch.zeros(k, device=d), torch.zeros(k, device=d)) if hooked else None
    CV.launch(geom, CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k, flags=CV.CONV_STATS if hooked else 0, epi_bn=[b.to(d) for b in bn],
              stats=stats, pool=pooled, tile=tile, ksplit=ksplit)
    _close(CV.from_cm(y, n, h, h), raw, what="raw output")
    _close(dec(pooled), ref, what="pooled means")
    if hooked:
        dd = z - shift.double().view(1, -1, 1, 1)
        # (the shift is the mean: s1 is a cancelling sum, bounded against the sum of magnitudes)
        assert (stats[1].cpu().double() - dd.sum((0, 2, 3))).abs().max().item() <= 1e-5 * dd.abs().sum((0, 2, 3)).max().item()
        _close(stats[2], (dd * dd).sum((0, 2, 3)), tol=1e-4, what="s2")
    # added, not stored: a second launch doubles the sums
    CV.launch(geom, CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k, epi_bn=[b.to(d) for b in bn], pool=pooled, tile=tile, ksplit=ksplit)
    _close(dec(pooled), 2 * ref, what="accumulation")
    # integer sums do not depend on the arrival order of the workgroups: a third launch into a fresh buffer repeats the first bit for bit
    again = torch.zeros_like(pooled)
    CV.launch(geom, CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k, epi_bn=[b.to(d) for b in bn], pool=again, tile=tile, ksplit=ksplit)
    assert torch.equal(again * 2, pooled)


@pytest.mark.parametrize("n,c,k,h,ksz,ksplit", [(16, 2048, 512, 7, 1, 0), (16, 512, 512, 7, 3, 0), (8, 1024, 256, 14, 1, 4),
                                                (8, 256, 256, 14, 3, 7), (4, 512, 128, 14, 1, 16), (16, 256, 256, 14, 3, 2)])
def test_split_k_matches_fp64_conv_and_leaves_the_workspace_clean(n, c, k, h, ksz, ksplit):
    """Few output tiles (the 14x14 / 7x7 stages): several workgroups share a tile, each walking a slice of K; the last
    arriver sums the slabs and runs the epilogue (here with BN + ReLU + moments, which must see the FULL sum exactly
    once).  Three launches back to back on the same workspace: the arrival counters must return to zero."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(c + k + h + ksplit)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, ksz, ksz, generator=g) * (c * ksz * ksz) ** -0.5
    bn = _bn(k, g)
    shift = torch.randn(k, generator=g) * 0.1
    pad = ksz // 2
    z = _bn_apply(F.conv2d(x.double(), w.double(), padding=pad), bn)
    ref = torch.relu(z)
    d = _dev()
    geom = CV.Geometry.forward(n, h, h, ksz, 1, pad)
    xc, wp = CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d))
    for rep in range(3):
        y = torch.full((k, n * h * h), float("nan"), device=d)
        s1, s2 = torch.zeros(k, device=d), torch.zeros(k, device=d)
        CV.launch(geom, xc, wp, y, c, k, flags=CV.CONV_EPI_APPLY | CV.CONV_EPI_RELU | CV.CONV_STATS, epi_bn=[t.to(d) for t in bn],
                  stats=(shift.to(d), s1, s2), ksplit=ksplit)
        _close(CV.from_cm(y, n, h, h), ref, what=f"launch {rep}")
        dz = z - shift.double().view(1, -1, 1, 1)
        assert (s1.cpu().double() - dz.sum((0, 2, 3))).abs().max().item() <= 1e-5 * dz.abs().sum((0, 2, 3)).max().item()
        _close(s2, (dz * dz).sum((0, 2, 3)), tol=1e-5, what="s2")
    ws = CV.workspace(d)
    assert int(ws[:65536].view(torch.int32).abs().sum().item()) == 0  # the counter prefix (slabs follow)


@pytest.mark.parametrize("n,c,k,h,ksz,stride", [(4, 64, 256, 28, 1, 1), (8, 1024, 256, 14, 1, 1), (4, 64, 64, 28, 3, 1),
                                                (8, 256, 256, 14, 3, 1), (8, 512, 512, 7, 3, 1), (4, 128, 128, 28, 3, 2),
                                                (8, 512, 512, 14, 3, 2), (4, 32, 64, 7, 3, 2), (4, 32, 64, 9, 3, 2),
                                                (4, 256, 512, 28, 1, 2), (4, 64, 128, 7, 1, 2)])
def test_data_gradient_matches_fp64_autograd(n, c, k, h, ksz, stride):
    """d input of conv(k x k, stride, pad k // 2): stride 1 = one launch with the flipped tap table; stride 2 (3x3) =
    one launch per parity class of the input pixel; strided pointwise = a half-resolution GEMM added at even
    positions by the consumer (RES_HALF on a following launch, here an identity 1x1)."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(n + c + k + h + 7 * ksz + stride)
    x = torch.randn(n, c, h, h, generator=g).double().requires_grad_(True)
    w = torch.randn(k, c, ksz, ksz, generator=g) * (c * ksz * ksz) ** -0.5
    pad = ksz // 2
    out = F.conv2d(x, w.double(), stride=stride, padding=pad)
    gy = torch.randn(out.shape, generator=g)
    out.backward(gy.double())
    d = _dev()
    wp = CV.pack_bwd(w.to(d))
    gy_cm = CV.to_cm(gy.to(d))
    geoms = CV.Geometry.dgrad(n, h, h, ksz, stride, pad)
    if stride == 2 and ksz == 1:
        ho = geoms[0].hy
        half = torch.full((c, n * ho * ho), float("nan"), device=d)
        CV.launch(geoms[0], gy_cm, wp, half, k, c)
        # consumer: an identity pointwise convolution on a zero tensor with the half-resolution addend
        eye = torch.eye(32, device=d).reshape(1, 32, 32).contiguous()
        zero = torch.zeros(32, n * h * h, device=d)
        for c0 in range(0, c, 32):
            got = torch.full((32, n * h * h), float("nan"), device=d)
            CV.launch(CV.Geometry.forward(n, h, h), zero, eye, got, 32, 32, flags=CV.CONV_RES_HALF,
                      res=half[c0:c0 + 32].contiguous())
            _close(CV.from_cm(got, n, h, h), x.grad[:, c0:c0 + 32])
        return
    gx = torch.full((c, n * h * h), float("nan"), device=d)
    for geom in geoms:
        CV.launch(geom, gy_cm, wp, gx, k, c)
    _close(CV.from_cm(gx, n, h, h), x.grad)


@pytest.mark.parametrize("n,c,k,h,hooked", [(8, 256, 1024, 14, True), (16, 512, 2048, 7, True), (4, 64, 256, 28, False)])
def test_bottleneck_tail_epilogue(n, c, k, h, hooked):
    """conv3 of a bottleneck in one launch: x2 raw -> relu(bn2(.)) on load -> 1x1 -> raw x3 (second output), moments of
    z3 = bn3(x3) (hooked layer), out = relu(z3 + identity)."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(h * c + k)
    x2 = torch.randn(n, c, h, h, generator=g)
    ident = torch.randn(n, k, h, h, generator=g)
    w = torch.randn(k, c, 1, 1, generator=g) * c ** -0.5
    bn2, bn3 = _bn(c, g), _bn(k, g)
    shift = torch.randn(k, generator=g) * 0.1
    a = torch.relu(_bn_apply(x2.double(), bn2))
    x3 = F.conv2d(a, w.double())
    z3 = _bn_apply(x3, bn3)
    out = torch.relu(z3 + ident.double())
    d = _dev()
    P = n * h * h
    y, yraw = torch.full((k, P), float("nan"), device=d), torch.full((k, P), float("nan"), device=d)
    s1, s2 = torch.zeros(k, device=d), torch.zeros(k, device=d)
    flags = CV.CONV_PRO_BN_RELU | CV.CONV_EPI_APPLY | CV.CONV_EPI_RELU | CV.CONV_RES | (CV.CONV_STATS if hooked else 0)
    CV.launch(CV.Geometry.forward(n, h, h), CV.to_cm(x2.to(d)), CV.pack_fwd(w.to(d)), y, c, k, flags=flags, y_raw=yraw,
              res=CV.to_cm(ident.to(d)), pro_bn=[t.to(d) for t in bn2], epi_bn=[t.to(d) for t in bn3],
              stats=(shift.to(d), s1, s2) if hooked else None)
    _close(CV.from_cm(yraw, n, h, h), x3, what="raw")
    _close(CV.from_cm(y, n, h, h), out, what="out")
    if hooked:
        dz = z3 - shift.double().view(1, -1, 1, 1)
        r1, r2 = dz.sum((0, 2, 3)), (dz * dz).sum((0, 2, 3))
        assert (s1.cpu().double() - r1).abs().max().item() <= 1e-5 * dz.abs().sum((0, 2, 3)).max().item()
        _close(s2, r2, tol=1e-5, what="s2")


@pytest.mark.parametrize("n,c,k,h,relu,inject,mask", [(8, 1024, 256, 14, True, True, False), (8, 256, 256, 14, True, False, False),
                                                       (4, 256, 64, 28, True, True, True), (8, 2048, 512, 7, False, True, False),
                                                       (8, 1024, 256, 14, True, "raw", False), (4, 256, 64, 28, True, "raw", True)])
def test_data_gradient_with_batchnorm_backward_epilogue(n, c, k, h, relu, inject, mask):
    """conv dgrad whose epilogue differentiates the BatchNorm (+ReLU) that FED the forward convolution:
        a = relu(bn(xr)), y = conv1x1(a);  given gy (+ a second gradient g2 arriving at a):
        g = W^T gy + g2 ; dz = g * [mask] + gscale (a_k + b_k (z - mu_k)) ; d gamma, d beta ; d xr = dz * s
    against fp64 autograd of the same composition.  inject == "raw" (VITTA_CONV_INJ_RAW, before_norm hooks): the statistics loss
    is a function of the RAW input xr -- its gradient joins d xr, d gamma / d beta do not see it."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(c + k + h)
    xr = torch.randn(n, k, h, h, generator=g)
    bn = _bn(k, g)
    # keep every pre-activation away from the ReLU kink (an fp32 / fp64 sign flip there is an O(1) gradient difference)
    z0 = _bn_apply(xr.double(), bn)
    xr = torch.where(z0.abs() < 1e-3, xr + 0.01, xr).double().requires_grad_(True)
    gam, bet = bn[0].double().requires_grad_(True), bn[1].double().requires_grad_(True)
    z = F.batch_norm(xr, bn[2].double(), bn[3].double(), gam, bet, False, 0.0, 1e-5)
    other = torch.randn(n, k, h, h, generator=g)  # a tensor whose sign decides the mask when `mask`
    if relu:
        a = z * (other.double() > 0) if mask else torch.relu(z)
    else:
        a = z
    w = torch.randn(c, k, 1, 1, generator=g) * k ** -0.5
    y = F.conv2d(a, w.double())
    gy = torch.randn(y.shape, generator=g)
    g2 = torch.randn(a.shape, generator=g)
    mu, ca, cb = torch.randn(k, generator=g) * 0.1, torch.randn(k, generator=g) * 1e-3, torch.randn(k, generator=g) * 1e-3
    gscale = 0.7
    loss = (y * gy.double()).sum() + (a * g2.double()).sum()
    if inject:
        v = lambda t: t.double().view(1, -1, 1, 1)
        f = xr if inject == "raw" else z
        loss = loss + gscale * ((v(ca) - v(cb) * v(mu)) * f + 0.5 * v(cb) * f * f).sum()
    loss.backward()
    d = _dev()
    P = n * h * h
    gx = torch.full((k, P), float("nan"), device=d)
    gm = torch.full((k, P), float("nan"), device=d)
    dgam, dbet = torch.full((k,), 0.25, device=d), torch.full((k,), -0.5, device=d)
    flags = CV.CONV_BWD_BN | CV.CONV_RES | (CV.CONV_BWD_RELU if relu else 0) | (CV.CONV_INJ_RAW if inject == "raw" else 0)
    CV.launch(CV.Geometry.dgrad(n, h, h)[0], CV.to_cm(gy.to(d)), CV.pack_bwd(w.to(d)), gx, c, k, flags=flags, y_raw=gm,
              res=CV.to_cm(g2.to(d)), bwd_bn=[t.to(d) for t in bn], bwd_x=CV.to_cm(xr.detach().float().to(d)),
              bwd_mask=CV.to_cm(other.to(d)) if mask else None,
              inj=(mu.to(d), ca.to(d), cb.to(d), torch.tensor([gscale], device=d)) if inject else None, dgamma=dgam, dbeta=dbet)
    _close(CV.from_cm(gx, n, h, h), xr.grad, what="dx")
    _close(dgam - 0.25, gam.grad, tol=1e-4, what="dgamma")
    _close(dbet + 0.5, bet.grad, tol=1e-4, what="dbeta")
    gref = F.conv_transpose2d(gy.double(), w.double()) + g2.double()
    if relu:
        gref = gref * ((other.double() > 0) if mask else (z.detach() > 0))
    _close(CV.from_cm(gm, n, h, h), gref, what="masked gradient")


# ---- where a three-term operand split could differ from fp32 arithmetic (and the exact-fp32 kernels are held to the same) ----
# Bound per OUTPUT ELEMENT, relative to that element's sum of |a b| (not to the tensor's maximum): an fp32 multiply-add chain
# of R terms is within R 2^-24 sum|a b| worst case, ~sqrt(R) 2^-24 typically; the split drops mid x lo, lo x mid, lo x lo
# (<= 2^-25 |a b| together) and rounds the third term (2^-27).  4 x 2^-23 x sqrt(R) holds both kernels with a margin of ~10
# and is 4 000 x below what a dropped `lo` term (2^-16) and 10^6 x below what a dropped `mid` term (2^-9) would produce.
def _elem_bound(mag, r):
    return 4.0 * 2.0 ** -23 * (r ** 0.5) * mag + 1e-30


def _log_uniform(nn_, lo, hi, g):
    return 10.0 ** (torch.rand(nn_, generator=g) * (hi - lo) + lo)


@pytest.mark.parametrize("n,c,k,h,ksz", [(8, 256, 64, 28, 1), (8, 128, 128, 28, 3), (16, 512, 512, 7, 3), (16, 1024, 256, 14, 1)])
def test_wide_dynamic_range_inside_one_reduction(n, c, k, h, ksz):
    """Per-channel scales log-uniform in 1e-4 .. 1e4 on the activations AND on the weights (un-normalised early features look
    like this): terms of one reduction span 16 decades.  Forward and data gradient, every element against fp64 relative to its
    own sum of magnitudes."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(77 * c + k + h + ksz)
    sa, sw_c, sw_k, sg = _log_uniform(c, -4, 4, g), _log_uniform(c, -2, 2, g), _log_uniform(k, -2, 2, g), _log_uniform(k, -4, 4, g)
    x = torch.randn(n, c, h, h, generator=g) * sa.view(1, -1, 1, 1)
    w = torch.randn(k, c, ksz, ksz, generator=g) * sw_c.view(1, -1, 1, 1) * sw_k.view(-1, 1, 1, 1) * (c * ksz * ksz) ** -0.5
    gy = torch.randn(n, k, h, h, generator=g) * sg.view(1, -1, 1, 1)
    pad = ksz // 2
    xd, wd, gd = x.double(), w.double(), gy.double()
    ref, mag = F.conv2d(xd, wd, padding=pad), F.conv2d(xd.abs(), wd.abs(), padding=pad)
    gref, gmag = F.conv_transpose2d(gd, wd, padding=pad), F.conv_transpose2d(gd.abs(), wd.abs(), padding=pad)
    d = _dev()
    y = torch.full((k, n * h * h), float("nan"), device=d)
    CV.launch(CV.Geometry.forward(n, h, h, ksz, 1, pad), CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k)
    err = (CV.from_cm(y, n, h, h).cpu().double() - ref).abs()
    bad = err > _elem_bound(mag, c * ksz * ksz)
    assert not bool(bad.any()), ("forward", int(bad.sum()), float((err / mag.clamp_min(1e-300)).max()))
    gx = torch.full((c, n * h * h), float("nan"), device=d)
    CV.launch(CV.Geometry.dgrad(n, h, h, ksz, 1, pad)[0], CV.to_cm(gy.to(d)), CV.pack_bwd(w.to(d)), gx, k, c)
    err = (CV.from_cm(gx, n, h, h).cpu().double() - gref).abs()
    bad = err > _elem_bound(gmag, k * ksz * ksz)
    assert not bool(bad.any()), ("data gradient", int(bad.sum()), float((err / gmag.clamp_min(1e-300)).max()))


@pytest.mark.parametrize("n,c,k,h,ksz", [(8, 256, 64, 28, 1), (8, 128, 128, 28, 3)])
def test_large_mean_small_variance_input_and_the_epilogue_moments(n, c, k, h, ksz):
    """x = 1000 + N(0, 1) (mean / sigma = 1e3): every product is ~1000 x the spread of the result, so a lost low-order term of
    the split shows up in the output's VARIANCE first.  BatchNorm constants as a trained net has them (running statistics =
    the convolution output's own), hooked moments about a shift near the mean (what the engine passes): output element-wise,
    sums within the bound that element-wise bound implies."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(5 * c + k + h + ksz)
    x = 1000.0 + torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, ksz, ksz, generator=g) * (c * ksz * ksz) ** -0.5
    pad = ksz // 2
    xd, wd = x.double(), w.double()
    raw, mag = F.conv2d(xd, wd, padding=pad), F.conv2d(xd.abs(), wd.abs(), padding=pad)
    rm, rv = raw.mean((0, 2, 3)).float(), raw.var((0, 2, 3), unbiased=False).float()
    gam, bet = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.3
    shift = bet + torch.randn(k, generator=g) * 0.05
    es = (gam.double() / torch.sqrt(rv.double() + 1e-5)).view(1, -1, 1, 1)
    z = (raw - rm.double().view(1, -1, 1, 1)) * es + bet.double().view(1, -1, 1, 1)
    d = _dev()
    y = torch.full((k, n * h * h), float("nan"), device=d)
    s1, s2 = torch.zeros(k, device=d), torch.zeros(k, device=d)
    CV.launch(CV.Geometry.forward(n, h, h, ksz, 1, pad), CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k,
              flags=CV.CONV_EPI_APPLY | CV.CONV_STATS, epi_bn=[t.to(d) for t in (gam, bet, rm, rv)], stats=(shift.to(d), s1, s2))
    # the fp32 epilogue itself rounds raw * s + t once more: (|raw s| + |t|) 2^-23
    eb = _elem_bound(mag, c * ksz * ksz) * es.abs() + 2.0 ** -22 * ((raw * es).abs() + (rm.double().view(1, -1, 1, 1) * es).abs())
    err = (CV.from_cm(y, n, h, h).cpu().double() - z).abs()
    assert not bool((err > eb).any()), float((err / eb).max())
    assert float((mag.mean((0, 2, 3)) / raw.std((0, 2, 3))).median()) > 50    # the case is what it says it is
    dz = z - shift.double().view(1, -1, 1, 1)
    r1, r2 = dz.sum((0, 2, 3)), (dz * dz).sum((0, 2, 3))
    b1 = eb.sum((0, 2, 3)) + 1e-5 * dz.abs().sum((0, 2, 3))
    b2 = (2 * dz.abs() * eb + eb * eb).sum((0, 2, 3)) + 1e-5 * r2
    assert bool(((s1.cpu().double() - r1).abs() <= b1).all()), float(((s1.cpu().double() - r1).abs() / b1).max())
    assert bool(((s2.cpu().double() - r2).abs() <= b2).all()), float(((s2.cpu().double() - r2).abs() / b2).max())
    # and the variance those sums give is the fp64 one to 1 %: a bf16-only product would be off by orders of magnitude
    cnt = n * h * h
    var = s2.cpu().double() / cnt - (s1.cpu().double() / cnt) ** 2
    vref = r2 / cnt - (r1 / cnt) ** 2
    assert float(((var - vref).abs() / vref).max()) <= 1e-2


@pytest.mark.parametrize("n,c,k,h,ksz", [(4, 64, 64, 28, 1), (4, 64, 64, 28, 3)])
def test_non_finite_inputs_stay_where_fp32_arithmetic_puts_them(n, c, k, h, ksz, arith):
    """One +Inf and one NaN among the activations.  Stated behaviour: an output element is non-finite exactly where fp32
    arithmetic makes it non-finite (the elements whose reduction contains the value), every other element is unaffected; the
    KIND is not preserved by the split form -- x - bf16(x) is NaN for x = Inf, so +-Inf outputs of the exact kernels come out
    as NaN.  Nothing downstream distinguishes them (a non-finite feature poisons the statistics either way)."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(c + k + h + ksz)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, ksz, ksz, generator=g) * (c * ksz * ksz) ** -0.5
    x[1, 3, 5, 7] = float("inf")
    x[2, 10, 9, 2] = float("nan")
    pad = ksz // 2
    ref = F.conv2d(x.double(), w.double(), padding=pad)
    d = _dev()
    y = torch.full((k, n * h * h), 12345.0, device=d)
    CV.launch(CV.Geometry.forward(n, h, h, ksz, 1, pad), CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k)
    got = CV.from_cm(y, n, h, h).cpu().double()
    fin = torch.isfinite(ref)
    assert int((~fin).sum()) == 2 * k * ksz * ksz                   # the two receptive fields, every output channel
    assert torch.equal(torch.isfinite(got), fin)
    assert (got[fin] - ref[fin]).abs().max().item() <= TOL * ref[fin].abs().max().item()
    if arith == "f32":                                              # the exact kernels also keep the kind
        assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.equal(got[torch.isinf(ref)], ref[torch.isinf(ref)])


def test_deferred_weight_gradient_reductions_of_a_group_equal_the_immediate_ones():
    """vitta_conv_wgrad_f32 with VITTA_WGRAD_DEFER_REDUCE + ONE vitta_conv_wgrad_reduce_f32 over the group (the convolutions of a
    bottleneck, trunk.py) == the same launches each followed by its own reduction, accumulated onto non-zero buffers."""
    from vitta_amd import conv as CV
    gen = torch.Generator().manual_seed(77)
    d = _dev()
    n, h = 4, 28
    group = [(64, 256, 1), (64, 64, 3), (256, 64, 1), (256, 512, 1)]
    geoms, xs, dys, bases = [], [], [], []
    for c, k, ksz in group:
        geoms.append(CV.Geometry.forward(n, h, h, ksz, 1, ksz // 2))
        xs.append(torch.randn(c, n * h * h, generator=gen).to(d))
        dys.append(torch.randn(k, n * h * h, generator=gen).to(d))
        bases.append((torch.randn(k, c, ksz, ksz, generator=gen) * 0.1).to(d))
    now = [b.clone() for b in bases]
    for g, x, dy, gw, (c, k, _) in zip(geoms, xs, dys, now, group):
        CV.wgrad(g, x, dy, gw, c, k)
    later = [b.clone() for b in bases]
    descs = [CV.wgrad(g, x, dy, gw, c, k, defer=i) for i, (g, x, dy, gw, (c, k, _)) in enumerate(zip(geoms, xs, dys, later, group))]
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(later, bases))   # nothing added yet
    CV.wgrad_reduce(descs)
    torch.cuda.synchronize()
    for a, b, base in zip(now, later, bases):
        assert (a - b).abs().max().item() <= 1e-5 * (a - base).abs().max().item()


def test_unsupported_shapes_are_refused_not_miscomputed():
    from vitta_amd import _lib, conv as CV
    d = _dev()
    x = torch.zeros(24, 4 * 36, device=d)
    with pytest.raises(_lib.VittaHipError):
        CV.launch(CV.Geometry.forward(4, 6, 6), x, torch.zeros(1, 24, 32, device=d), torch.zeros(32, 4 * 36, device=d), 24, 32)
    with pytest.raises(_lib.VittaHipError):
        CV.launch(CV.Geometry.forward(4, 6, 6), x.cpu(), torch.zeros(1, 24, 32), torch.zeros(32, 4 * 36), 24, 32)


@pytest.mark.parametrize("n,h,w", [(3, 224, 224), (2, 112, 112), (2, 64, 64), (1, 32, 48), (2, 30, 36), (1, 226, 220)])
def test_stem_conv7_matches_fp64_conv(n, h, w):
    """vitta_stem_conv7_f32 (stem_conv.hip): the 7x7 / stride 2 / pad 3 convolution of the ResNet stem, NCHW in and out,
    tiles of 256 consecutive output pixels (ragged last tile, odd output sizes)."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(n * 7 + h + w)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) * 147 ** -0.5
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=3)
    d = _dev()
    y = CV.stem_conv(x.to(d), CV.pack_stem(wt.to(d)))
    assert y.shape == ref.shape
    _close(y, ref)


@pytest.mark.parametrize("m,n,k", [(16, 101, 2048), (8, 101, 2048), (3, 7, 64), (24, 174, 2048), (1, 400, 2048)])
def test_head_linear_forward_and_gradients(m, n, k):
    """vitta_linear_{fwd,bwd}_f32 (head.hip) against fp64: y, dx, and ACCUMULATED dw / db."""
    from vitta_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g)
    wt = torch.randn(n, k, generator=g) * k ** -0.5
    b = torch.randn(n, generator=g)
    gy = torch.randn(m, n, generator=g)
    xr, wr, br = (t.double().requires_grad_() for t in (x, wt, b))
    ref = F.linear(xr, wr, br)
    ref.backward(gy.double())
    d = _dev()
    old = ops.DIRECT_PARAM_GRAD
    ops.DIRECT_PARAM_GRAD = False
    try:
        xg, wg, bg = (t.to(d).requires_grad_() for t in (x, wt, b))
        y = ops.HeadLinear.apply(xg, wg, bg)
        y.backward(gy.to(d))
    finally:
        ops.DIRECT_PARAM_GRAD = old
    _close(y, ref.detach())
    _close(xg.grad, xr.grad)
    _close(wg.grad, wr.grad)
    _close(bg.grad, br.grad)
    # frozen head (affine-only adaptation): only the data gradient
    xg2 = x.to(d).requires_grad_()
    ops.HeadLinear.apply(xg2, wt.to(d), b.to(d)).backward(gy.to(d))
    _close(xg2.grad, xr.grad)


@pytest.mark.parametrize("n,c,k,h,ksz,stride,pro", [
    (4, 64, 64, 56, 1, 1, False), (16, 256, 64, 14, 1, 1, False), (2, 64, 256, 28, 1, 1, True), (8, 512, 128, 7, 1, 1, True),
    (4, 64, 64, 28, 3, 1, False), (16, 256, 256, 14, 3, 1, False), (8, 128, 128, 28, 3, 2, False), (16, 512, 512, 7, 3, 1, False),
    (4, 256, 512, 28, 1, 2, False), (3, 64, 128, 10, 3, 1, False), (4, 64, 64, 9, 3, 2, False), (1, 128, 64, 6, 1, 1, False)])
def test_weight_gradient_matches_fp64_autograd(n, c, k, h, ksz, stride, pro):
    """vitta_conv_wgrad_f32 (conv_wgrad.hip) against the fp64 autograd weight gradient, ACCUMULATED onto a non-zero buffer;
    `pro`: the forward applied relu(bn(x)) on load, so does the gradient."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(n + c + k + h + ksz)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, ksz, ksz, generator=g) * (c * ksz * ksz) ** -0.5
    pad = ksz // 2
    bn = _bn(c, g) if pro else None
    wr = w.double().requires_grad_()
    a = torch.relu(_bn_apply(x.double(), bn)) if pro else x.double()
    y = F.conv2d(a, wr, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    d = _dev()
    geom = CV.Geometry.forward(n, h, h, ksz, stride, pad)
    base = torch.randn(k, c, ksz, ksz, generator=g) * 0.1
    gw = base.clone().to(d)
    CV.wgrad(geom, CV.to_cm(x.to(d)), CV.to_cm(dy.to(d)), gw, c, k, pro_bn=[t.to(d) for t in bn] if pro else None)
    _close(gw, wr.grad + base.double(), tol=5e-5)


@pytest.mark.parametrize("n,h,w", [(3, 224, 224), (2, 112, 112), (2, 64, 64), (1, 40, 56)])
def test_stem_backward_and_weight_gradient_match_fp64_autograd(n, h, w):
    """bn1 -> relu -> maxpool backward WITH the gradient w.r.t. the convolution output (vitta_stem_bn_relu_pool_bwd_f32)
    and the stem weight gradient (vitta_stem_conv7_wgrad_f32), accumulated onto a non-zero buffer, against fp64 autograd of
    BatchNorm(eval) -> ReLU -> MaxPool2d(3, 2, 1) on the kernel's own convolution output, resp. of the convolution.  An
    arg-max decided inside fp32 round-off of a tie moves one whole gradient entry: a handful of such windows are allowed."""
    from vitta_amd import _lib, conv as CV
    from vitta_amd.ops import _p, _ptr4, _stream
    g = torch.Generator().manual_seed(n + h + w)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) * 147 ** -0.5
    bn = _bn(64, g)
    d = _dev()
    yd = CV.stem_conv(x.to(d), CV.pack_stem(wt.to(d)))
    y = yd.cpu().double().requires_grad_()
    pooled = F.max_pool2d(torch.relu(_bn_apply(y, bn)), 3, 2, 1)
    gp = torch.randn(pooled.shape, generator=g)
    pooled.backward(gp.double())
    dy, dg, db = torch.zeros_like(yd), torch.zeros(64, device=d), torch.zeros(64, device=d)
    bnd = [t.to(d) for t in bn]
    _lib.check(_lib.lib().vitta_stem_bn_relu_pool_bwd_f32(_p(yd), _p(gp.to(d)), _ptr4(*bnd), 1e-5, n, 64, yd.shape[2], yd.shape[3],
                                                          _p(dg), _p(db), _p(dy), _stream()), "stem bwd")
    bad = ((dy.cpu().double() - y.grad).abs() > 1e-5 * y.grad.abs().max()).sum().item()
    assert bad <= 8, bad
    wr = wt.double().requires_grad_()
    F.conv2d(x.double(), wr, stride=2, padding=3).backward(dy.cpu().double())
    base = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    gw = base.clone().to(d)
    CV.stem_wgrad(x.to(d), dy, gw)
    _close(gw, wr.grad + base.double(), tol=5e-5, what="dw")


def test_repack_of_trainable_weights_matches_the_permutes():
    """vitta_conv_repack_f32 (one launch over every trainable convolution, LDS-tiled) == pack_fwd / pack_bwd per weight,
    incl. C * taps not a multiple of the 32-wide tile and 1x1 weights without a backward pack."""
    import ctypes as C
    import numpy as np
    from vitta_amd import conv as CV
    from vitta_amd._lib import check, lib
    d = _dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 64, 1, 1), (64, 48, 3, 3), (128, 64, 3, 3), (256, 64, 1, 1), (32, 16, 3, 3), (2048, 512, 1, 1), (96, 80, 3, 3)]
    ws = [torch.randn(s, generator=g).to(d) for s in shapes]
    dt = np.dtype([("src", "<u8"), ("fwd", "<u8"), ("bwd", "<u8"), ("first", "<i8"), ("K", "<i4"), ("C", "<i4"), ("taps", "<i4"),
                   ("pad", "<i4")])
    tab = np.zeros(len(ws), dtype=dt)
    outs, first = [], 0
    for i, w in enumerate(ws):
        k, c, kh, kw = w.shape
        taps = kh * kw
        pf = torch.full((taps, c, k), float("nan"), device=d)
        pb = torch.full((taps, k, c), float("nan"), device=d) if taps > 1 else None
        outs.append((pf, pb))
        tab[i] = (w.data_ptr(), pf.data_ptr(), pb.data_ptr() if pb is not None else 0, first, k, c, taps, 0)
        first += k * c * taps
    dtab = torch.from_numpy(tab.view(np.uint8).copy()).to(d)
    check(lib().vitta_conv_repack_f32(C.c_void_p(dtab.data_ptr()), len(ws), first, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "vitta_conv_repack_f32")
    for w, (pf, pb) in zip(ws, outs):
        assert torch.equal(pf, CV.pack_fwd(w).view_as(pf))
        if pb is not None:
            assert torch.equal(pb, CV.pack_bwd(w).view_as(pb))


def test_pack_b3_reconstructs_the_weights_and_the_split_kernel_is_the_path(arith):
    """vitta_conv_pack_b3: hi + mid + lo of the [tap][slab][plane][octet][O][8] image gives the fp32 weight back to 2^-23
    relative, hi alone is its bf16 rounding; and a qualifying launch runs on conv_b3.hip exactly when the arithmetic
    mode says so (vitta_conv_kernel)."""
    from vitta_amd import _lib, conv as CV
    d = _dev()
    g = torch.Generator().manual_seed(5)
    taps, r, o = 9, 64, 128
    wp = (torch.randn(taps, r, o, generator=g) * torch.rand(taps, r, o, generator=g).exp()).to(d)
    img = CV.pack_b3(wp).view(torch.bfloat16).view(taps, r // 32, 3, 4, o, 8).float()
    planes = img.permute(2, 0, 1, 3, 5, 4).reshape(3, taps, r, o)  # [plane][tap][slab * 32 + octet * 8 + j][o]
    assert torch.equal(planes[0], wp.to(torch.bfloat16).float())
    rec = planes[0].double() + planes[1].double() + planes[2].double()
    assert ((rec - wp.double()).abs() <= 2.0 ** -23 * wp.double().abs()).all()
    n, c, k, h = 4, 64, 128, 28
    x = torch.randn(c, n * h * h, device=d)
    w = torch.randn(k, c, 1, 1, device=d)
    y = torch.empty(k, n * h * h, device=d)
    CV.KERNEL_COUNTS = {}
    try:
        CV.launch(CV.Geometry.forward(n, h, h), x, CV.pack_fwd(w), y, c, k)
        counts = dict(CV.KERNEL_COUNTS)
    finally:
        CV.KERNEL_COUNTS = None
    assert (counts.get(_lib.CONV_KERNEL_B3, 0) == 1) == (arith == "b3"), counts


@pytest.mark.parametrize("n,c,k,h", [(4, 128, 128, 28), (8, 512, 512, 14), (4, 64, 64, 7), (4, 64, 128, 9), (16, 256, 256, 28),
                                     (32, 512, 512, 4), (32, 256, 256, 8), (32, 128, 128, 16), (16, 512, 512, 2)])
def test_parity_merged_stride2_data_gradient_is_one_launch(n, c, k, h, arith):
    """d input of conv(3x3, stride 2, pad 1) with the four parity classes of the input pixel in ONE launch
    (VITTA_CONV_PARITY4, conv_b3.hip): same result as fp64 autograd; the exact-fp32 arithmetic keeps one launch per class
    (the merged descriptor is refused there)."""
    from vitta_amd import conv as CV
    g = torch.Generator().manual_seed(n + c + k + h)
    x = torch.randn(n, c, h, h, generator=g).double().requires_grad_(True)
    w = torch.randn(k, c, 3, 3, generator=g) * (c * 9) ** -0.5
    out = F.conv2d(x, w.double(), stride=2, padding=1)
    gy = torch.randn(out.shape, generator=g)
    out.backward(gy.double())
    d = _dev()
    pack = CV.make_pack(CV.pack_bwd(w.to(d)))
    gm = CV.Geometry.dgrad_merged(n, h, h, 3, 2, 1)
    assert gm is not None and sum(gm.cls_ntaps) == 9
    ok = CV.merged_dgrad_supported(gm, pack, k, c)
    assert ok == (arith == "b3")
    if not ok:
        return
    gx = torch.full((c, n * h * h), float("nan"), device=d)
    CV.KERNEL_COUNTS = {}
    try:
        CV.launch(gm, CV.to_cm(gy.to(d)), pack, gx, k, c)
        counts = dict(CV.KERNEL_COUNTS)
    finally:
        CV.KERNEL_COUNTS = None
    assert sum(counts.values()) == 1
    _close(CV.from_cm(gx, n, h, h), x.grad)


@pytest.mark.parametrize("bad", ["nan", "inf", "1e12"])
def test_pooled_means_are_poisoned_by_non_finite_or_out_of_range_activations(bad, arith):
    """VITTA_CONV_POOL sums are 64-bit fixed point: a NaN / Inf activation has no integer image and |mean| >= 2^31 would wrap.  The
    reference's adaptive_avg_pool2d (temporal_module.py:53) hands NaN / Inf to the TAM; here such a contribution -- non-finite, or a
    32-pixel block's share of a frame mean reaching 2^22 -- exchanges the word for INT64_MIN (poisoned band |v| >= 2^61, which no
    later addition leaves), and the TAM branch kernels decode the band as NaN.  Stated deviation: Inf and finite means beyond ~5e8
    arrive as NaN.  Frames without such an activation keep their exact sums; the convolution output itself is untouched."""
    from vitta_amd import _lib, conv as CV
    from vitta_amd.ops import _p, _ptr4, _stream
    n, t, c, k, h = 16, 8, 64, 64, 28
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, 1, 1, generator=g) * c ** -0.5
    bn = _bn(k, g)
    f_bad, ch_bad = 3, 7
    x[f_bad, ch_bad, 5, 9] = {"nan": float("nan"), "inf": float("inf"), "1e12": 1e12}[bad]
    clean = x.clone()
    clean[f_bad, ch_bad, 5, 9] = 0.0
    ref = _bn_apply(F.conv2d(clean.double(), w.double()), bn).clamp_min(0).mean((2, 3))
    d = _dev()
    geom = CV.Geometry.forward(n, h, h, 1, 1, 0)
    y = torch.empty(k, n * h * h, device=d)
    pooled = torch.zeros(n, k, dtype=torch.int64, device=d)
    CV.launch(geom, CV.to_cm(x.to(d)), CV.pack_fwd(w.to(d)), y, c, k, epi_bn=[b.to(d) for b in bn], pool=pooled)
    q = pooled.cpu()
    poisoned = (q >= 2 ** 61) | (q < -2 ** 61)
    # which channels of the bad frame see a non-finite / huge activation: every one for NaN / Inf (the pixel feeds all of them through
    # relu(bn(.)) -- a negative infinity is rectified to 0 and stays exact), the positively weighted ones for 1e10
    z_full = _bn_apply(F.conv2d(x.double(), w.double()), bn)
    z_bad = z_full[f_bad, :, 5, 9]
    thr = 2.0 ** 22 * h * h  # the pixel's value at which its 32-pixel block's share of the frame mean reaches 2^22
    must = torch.isnan(z_bad) | (z_bad > 4 * thr)
    may = torch.isnan(z_bad) | (z_bad > 0.25 * thr)
    if bad == "inf" and arith == "b3":  # the split arithmetic turns an infinite operand into NaN (x - bf16(x) is NaN for x = Inf:
        must = may = torch.ones_like(must)  # conv_b3.hip's header, test_non_finite_inputs_...): every channel of the pixel is poisoned
    assert (poisoned[f_bad] | ~must).all() and (~poisoned[f_bad] | may).all(), (poisoned[f_bad].sum(), must.sum(), may.sum())
    assert int(poisoned.sum()) == int(poisoned[f_bad].sum()) and int(must.sum()) > 0  # no other frame
    ok = ~poisoned
    got = q.double() * 2.0 ** -32
    want = torch.where(torch.isfinite(z_full), z_full, torch.zeros_like(z_full)).clamp_min(0).mean((2, 3))  # (-inf is rectified to 0)
    assert ((got[ok] - want[ok]).abs() <= 1e-5 * want[ok].abs() + 1e-5 * ref.abs().max()).all()
    # the TAM branch kernels decode the band as NaN for the clip that holds the frame, and only for it
    L = _lib.lib()
    nb, o = n // t, k // 4
    r = lambda *s: torch.randn(*s, generator=g).to(d)
    wg1, wg3, w0, w3 = r(2 * t, t) * 0.3, r(3, 2 * t) * 0.3, r(o, k, 3) * (3 * k) ** -0.5, r(k, o) * o ** -0.5
    bng = [torch.rand(2 * t, generator=g).to(d) + 0.5, r(2 * t) * 0.1, r(2 * t) * 0.1, torch.rand(2 * t, generator=g).to(d) + 0.5]
    bnl = [torch.rand(o, generator=g).to(d) + 0.5, r(o) * 0.1, r(o) * 0.1, torch.rand(o, generator=g).to(d) + 0.5]
    sync = torch.zeros(256, dtype=torch.int32, device=d)
    for fused in (False, True):
        kern, gate, hpre = torch.empty(nb * k, 3, device=d), torch.empty(nb, k, t, device=d), torch.empty(2, nb, o, t, device=d)
        args = (_p(pooled.view(nb, t, k)), _p(wg1), _ptr4(*bng), 1e-5, _p(wg3), _p(w0), _ptr4(*bnl), 1e-5, _p(w3), nb, k, t)
        if fused:
            _lib.check(L.vitta_tam_branch_fwd_fused_f32(*args, _p(kern), _p(gate), _p(hpre), _p(sync), 1, _stream()), "fwd fused")
        else:
            _lib.check(L.vitta_tam_branch_fwd_f32(*args, _p(kern), _p(gate), _p(hpre), 1, _stream()), "fwd")
        torch.cuda.synchronize()
        clip = f_bad // t
        assert torch.isnan(gate[clip]).any() and torch.isnan(kern.view(nb, k, 3)[clip]).any()
        assert torch.isfinite(gate[1 - clip]).all() and torch.isfinite(kern.view(nb, k, 3)[1 - clip]).all()
