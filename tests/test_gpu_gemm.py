"""Dense layers of Video Swin-B on the hand-written GEMM (csrc/gemm.hip, through the C ABI) -- GPU box only.

Reference arithmetic: torch.nn.functional.linear / nn.GELU as models/videoswintransformer_models/swin_transformer.py:30-35,
144, 165, 304-311 call them, evaluated in fp64 on the CPU.  Tolerance (fp32 products, fp32 accumulation over K <= 4096):
|err| <= 2e-5 * max|y| + 1e-6."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _close(got, ref, rel=2e-5):
    ref = ref.to(torch.float64)
    err = (got.detach().cpu().to(torch.float64) - ref).abs().max().item()
    assert err <= rel * ref.abs().max().item() + 1e-6, (err, ref.abs().max().item())


# (M, N, K): Swin-B stage shapes (tokens x out x in) incl. ragged token counts (784 = 12.25 x 64, 392) and N = 3 C
SHAPES = [(3136, 1536, 512), (784, 1024, 4096), (392, 3072, 1024), (12544, 256, 256), (50176 // 4, 384, 128), (200, 128, 32),
          (1, 256, 64), (3136, 512, 2048)]


@pytest.mark.parametrize("tile", [1, 2, 3])
@pytest.mark.parametrize("m,n,k", SHAPES)
def test_gemm_modes_vs_fp64(m, n, k, tile):
    from vitta_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * k ** -0.5
    b = torch.randn(n, generator=g)
    aux = torch.randn(m, n, generator=g) * 1.5
    ad, wd, bd, auxd = (t.to(_dev()) for t in (a, w, b, aux))
    ops.GEMM_TILE = tile
    try:
        y0 = ops.gemm_nt(ad, wd, bd)
        y0n = ops.gemm_nt(ad, wd)
        pre = torch.empty(m, n, device=_dev())
        y1 = ops.gemm_nt(ad, wd, bd, mode=1, pre=pre)
        y2 = ops.gemm_nt(ad, wd, mode=2, aux=auxd)
    finally:
        ops.GEMM_TILE = 0
    a64, w64, b64, x64 = a.double(), w.double(), b.double(), aux.double()
    lin = a64 @ w64.t()
    _close(y0, lin + b64)
    _close(y0n, lin)
    _close(pre, lin + b64)
    _close(y1, F.gelu(lin + b64))
    x = x64.clone().requires_grad_(True)
    (dg,) = torch.autograd.grad(F.gelu(x).sum(), x)
    _close(y2, lin * dg, rel=4e-5)


@pytest.mark.parametrize("grid", [256, 512, 1000])
@pytest.mark.parametrize("m,n,k", [(3136, 512, 2048), (3136, 512, 1536), (784, 1024, 4096), (12544, 256, 1024), (200, 128, 32),
                                   (1, 256, 64), (392, 3072, 1024), (3100, 500, 512)])
def test_gemm_stream_k_vs_fp64_and_deterministic(m, n, k, grid):
    """vitta_gemm_nt_sk_f32 (the K slabs of all 64 x 64 tiles cut into `grid` equal ranges; partial tiles through the workspace, the
    last arriver of a tile's ticket adds them in range order): every epilogue against fp64, ragged M / N, ranges shorter and longer
    than a tile, fewer slabs than workgroups, a grid that is no multiple of eight; two launches give the same bits; the arrival
    counters are zero afterwards."""
    from vitta_amd import _lib
    from vitta_amd.ops import _p, _stream
    L, d = _lib.lib(), _dev()
    g = torch.Generator().manual_seed(m + n + k + grid)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * k ** -0.5
    b = torch.randn(n, generator=g)
    aux = torch.randn(m, n, generator=g) * 1.5
    ad, wd, bd, auxd = (t.to(d) for t in (a, w, b, aux))
    ws = torch.zeros(int(L.vitta_gemm_nt_sk_workspace_bytes(grid)), dtype=torch.uint8, device=d)

    def run(bias, mode, auxt=None, pre=None):
        y = torch.empty(m, n, device=d)
        _lib.check(L.vitta_gemm_nt_sk_f32(_p(ad), _p(wd), _p(bias), _p(auxt), _p(y), _p(pre), m, n, k, mode, grid, _p(ws), ws.numel(),
                                          _stream()), "vitta_gemm_nt_sk_f32")
        return y
    y0, y0b = run(bd, 0), run(bd, 0)
    y0n = run(None, 0)
    pre = torch.empty(m, n, device=d)
    y1 = run(bd, 1, pre=pre)
    y2 = run(None, 2, auxt=auxd)
    torch.cuda.synchronize()
    assert torch.equal(y0, y0b)
    assert int(ws[:65536 * 4].view(torch.int32).abs().max()) == 0
    a64, w64, b64, x64 = a.double(), w.double(), b.double(), aux.double()
    lin = a64 @ w64.t()
    _close(y0, lin + b64)
    _close(y0n, lin)
    _close(pre, lin + b64)
    _close(y1, F.gelu(lin + b64))
    x = x64.clone().requires_grad_(True)
    (dg,) = torch.autograd.grad(F.gelu(x).sum(), x)
    _close(y2, lin * dg, rel=4e-5)
    # what it refuses
    assert L.vitta_gemm_nt_sk_f32(_p(ad), _p(wd), None, None, _p(y0), None, m, n, k, 0, grid, _p(ws), 1024, _stream()) == -1  # workspace too small
    assert L.vitta_gemm_nt_sk_f32(_p(ad), _p(wd), None, None, _p(y0), None, m, n, k, 0, 0, _p(ws), ws.numel(), _stream()) == -1


def test_gemm_nt_takes_the_stream_k_form_where_the_tile_count_quantises_badly(abi_calls):
    """ops.gemm_nt: 392 tiles with K = 2048 -> vitta_gemm_nt_sk_f32 (same numbers as the plain launch to round-off); K = 512 or a
    forced tile -> vitta_gemm_nt_f32."""
    from vitta_amd import ops
    d = _dev()
    g = torch.Generator().manual_seed(1)
    a, w = torch.randn(3136, 2048, generator=g).to(d), (torch.randn(512, 2048, generator=g) * 2048 ** -0.5).to(d)
    assert ops.gemm_sk_pays(3136, 512, 2048) and not ops.gemm_sk_pays(3136, 512, 512) and not ops.gemm_sk_pays(3136, 1536, 2048)
    y = ops.gemm_nt(a, w)
    assert abi_calls.abi.get("vitta_gemm_nt_sk_f32", 0) == 1 and abi_calls.abi.get("vitta_gemm_nt_f32", 0) == 0
    ops.GEMM_TILE = 3
    try:
        y3 = ops.gemm_nt(a, w)
    finally:
        ops.GEMM_TILE = 0
    assert abi_calls.abi.get("vitta_gemm_nt_f32", 0) == 1
    assert (y - y3).abs().max().item() <= 1e-5 * y3.abs().max().item()
    ops.gemm_nt(a[:, :512].contiguous(), w[:, :512].contiguous())
    assert abi_calls.abi.get("vitta_gemm_nt_f32", 0) == 2


def test_gemm_rejects_unsupported_and_host_tensors():
    from vitta_amd import _lib, ops
    assert not ops.gemm_nt_supported(64, 64, 48)
    with pytest.raises(_lib.VittaHipError):
        ops.gemm_nt(torch.randn(8, 32), torch.randn(8, 32))
    with pytest.raises(_lib.VittaHipError):
        ops.gemm_nt(torch.randn(8, 48, device=_dev()), torch.randn(8, 48, device=_dev()))


@pytest.mark.parametrize("train_weights", [False, True])
def test_mlp_and_linear_autograd_match_module_path(train_weights):
    """Mlp / Linear modules on the hand-written path vs the same modules on torch's own kernels: outputs, input gradient,
    and (SGD over all parameters) weight / bias gradients."""
    from vitta_amd import ops, swin
    torch.manual_seed(3)
    mlp = swin.Mlp(256, 1024).to(_dev())
    lin = torch.nn.Linear(256, 768).to(_dev())
    for p in list(mlp.parameters()) + list(lin.parameters()):
        p.requires_grad_(train_weights)
    x = torch.randn(2, 196, 256, device=_dev())
    gy = torch.randn(2, 196, 256, device=_dev())
    gq = torch.randn(2, 196, 768, device=_dev())
    res = {}
    old = ops.DIRECT_PARAM_GRAD
    ops.DIRECT_PARAM_GRAD = False
    try:
        for fused in (True, False):
            swin.FUSED_DENSE = fused
            xi = x.clone().requires_grad_(True)
            params = [p for p in list(mlp.parameters()) + list(lin.parameters()) if p.requires_grad]
            y, q = mlp(xi), swin.linear(lin, xi)
            grads = torch.autograd.grad([y, q], [xi] + params, [gy, gq])
            res[fused] = [y, q] + list(grads)
    finally:
        swin.FUSED_DENSE = True
        ops.DIRECT_PARAM_GRAD = old
    assert len(res[True]) == len(res[False]) == (3 + (6 if train_weights else 0))
    for got, ref in zip(res[True], res[False]):
        assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("tile", [1, 2, 3])
@pytest.mark.parametrize("m,n,k", [(3136, 1536, 512), (784, 1024, 4096), (392, 3072, 1024), (12544, 256, 256), (200, 128, 64),
                                   (1, 256, 64)])
def test_gemm_bf16_operands_vs_fp64_on_rounded_operands(m, n, k, tile):
    """bf16-operand variant: the products of bf16-rounded operands are exact in fp32, so against fp64 on the ROUNDED
    operands only the fp32 accumulation order is left: same tolerance as the fp32 kernel (x2)."""
    from vitta_amd import ops
    g = torch.Generator().manual_seed(m + n + k + 1)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * k ** -0.5
    b = torch.randn(n, generator=g)
    aux = torch.randn(m, n, generator=g) * 1.5
    ad, bd, auxd = (t.to(_dev()) for t in (a, b, aux))
    wb = w.to(_dev()).to(torch.bfloat16)
    ops.GEMM_TILE = tile
    try:
        y0 = ops.gemm_nt(ad, wb, bd)
        pre = torch.empty(m, n, device=_dev())
        y1 = ops.gemm_nt(ad, wb, bd, mode=1, pre=pre)
        y2 = ops.gemm_nt(ad, wb, mode=2, aux=auxd)
    finally:
        ops.GEMM_TILE = 0
    a64, w64 = a.to(torch.bfloat16).double(), w.to(torch.bfloat16).double()
    lin = a64 @ w64.t()
    _close(y0, lin + b.double(), rel=4e-5)
    _close(pre, lin + b.double(), rel=4e-5)
    _close(y1, F.gelu(lin + b.double()), rel=4e-5)
    x = aux.double().clone().requires_grad_(True)
    (dg,) = torch.autograd.grad(F.gelu(x).sum(), x)
    _close(y2, lin * dg, rel=8e-5)


def test_dense_bf16_flag_switches_module_path_and_stays_close_to_fp32():
    from vitta_amd import ops, swin
    torch.manual_seed(5)
    mlp = swin.Mlp(256, 1024).to(_dev())
    for p in mlp.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 196, 256, device=_dev(), requires_grad=True)
    out = {}
    try:
        for flag in (False, True):
            ops.DENSE_BF16 = flag
            y = mlp(x)
            (gx,) = torch.autograd.grad(y, x, torch.ones_like(y))
            out[flag] = (y.detach(), gx)
    finally:
        ops.DENSE_BF16 = False
    for got, ref in zip(out[True], out[False]):
        err = (got - ref).abs().max().item()
        assert 0 < err <= 2e-2 * ref.abs().max().item()   # bf16 operand rounding: visible, and bounded


@pytest.mark.parametrize("train", [False, True])
def test_patch_embedding_as_dense_product_matches_conv3d(train):
    """PatchEmbed3D (swin_transformer.py:361-399: Conv3d with kernel == stride) on the dense kernel: same tokens as the
    library convolution, and (SGD over all parameters) the same weight / bias gradients."""
    from vitta_amd import ops, swin
    torch.manual_seed(11)
    pe = swin.PatchEmbed3D(patch_size=(2, 4, 4), in_chans=3, embed_dim=128, norm_layer=torch.nn.LayerNorm).to(_dev())
    for p in pe.parameters():
        p.requires_grad_(train)
    x = torch.randn(2, 3, 8, 32, 32, device=_dev())
    g = torch.randn(2, 4, 8, 8, 128, device=_dev())
    res = {}
    old = ops.DIRECT_PARAM_GRAD
    ops.DIRECT_PARAM_GRAD = False
    try:
        for fused in (True, False):
            swin.FUSED_DENSE = fused
            if train:
                y = pe(x)
                grads = torch.autograd.grad(y, [pe.proj.weight, pe.proj.bias], g)
                res[fused] = [y.detach()] + list(grads)
            else:
                with torch.no_grad():
                    res[fused] = [pe(x)]
    finally:
        swin.FUSED_DENSE = True
        ops.DIRECT_PARAM_GRAD = old
    for got, ref in zip(res[True], res[False]):
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-6




@pytest.mark.parametrize("m,n,k,acc", [(6272, 384, 128, True), (784, 1024, 4096, True), (3136, 512, 2048, False), (50176, 128, 96, True),
                                       (1568, 2048, 512, False)])
def test_dense_weight_gradient_on_the_convolution_kernel_vs_fp64(m, n, k, acc):
    """ops._weight_grad_conv: dW [N, K] (+)= g^T x of an nn.Linear (swin_transformer.py:30-35, 144, 165, 304-311 under SGD over
    all parameters, corpus/basics.py:547-560) as a pointwise `vitta_conv_f32` launch with the tokens as channel axis."""
    from vitta_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(m + n + k)
    g2, x2 = torch.randn(m, n, generator=gen).to(dev), torch.randn(m, k, generator=gen).to(dev)
    base = torch.randn(n, k, generator=gen).to(dev)
    out = base.clone()
    assert ops._weight_grad_conv(g2, x2, out, acc)
    ref = g2.double().t().cpu() @ x2.double().cpu() + (base.double().cpu() if acc else 0)
    _close(out, ref, rel=3e-5)
    assert not ops._weight_grad_conv(g2[:, :n - 1].contiguous(), x2, out[:n - 1].contiguous(), acc)  # N % 32: declined, not miscomputed


@pytest.mark.parametrize("m,n,k,bias", [(256, 128, 64, False), (3136, 1536, 512, True), (12544, 512, 2048, True), (1000, 384, 128, False),
                                         (50176, 384, 128, True), (129, 256, 192, True)])
def test_gemm_bf16x_vs_fp64_of_the_rounded_operands(m, n, k, bias):
    """gemm_bf16x.hip: y = a b^T (+ bias) on bfloat16 operands in memory (LDS-DMA of both, 128 x 128 tiles, fp32
    accumulation) against the fp64 product of the same bf16 values: only the accumulation order differs (ragged M included)."""
    from vitta_amd import _lib
    import ctypes as C
    dev = _dev()
    gen = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=gen).to(dev).to(torch.bfloat16)
    b = (torch.randn(n, k, generator=gen) * k ** -0.5).to(dev).to(torch.bfloat16)
    bv = torch.randn(n, generator=gen).to(dev) if bias else None
    y = torch.full((m, n), float("nan"), device=dev)
    L = _lib.lib()
    assert L.vitta_gemm_bf16x_supported(m, n, k)
    _lib.check(L.vitta_gemm_nt_bf16x_f32(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(bv.data_ptr() if bias else 0),
                                         C.c_void_p(y.data_ptr()), m, n, k, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "vitta_gemm_nt_bf16x_f32")
    ref = a.double().cpu() @ b.double().cpu().t() + (bv.double().cpu() if bias else 0)
    _close(y, ref, rel=2e-5)
    assert not L.vitta_gemm_bf16x_supported(m, n + 64, k) and not L.vitta_gemm_bf16x_supported(m, n, k + 16)


@pytest.mark.parametrize("m,n,k", [(3136, 2048, 512), (1000, 512, 128), (12544, 256, 1024), (129, 128, 64)])
@pytest.mark.parametrize("out_bf16", [False, True])
def test_gemm_bf16x_epilogues_and_bfloat16_outputs(m, n, k, out_bf16):
    """vitta_gemm_nt_bf16x: mode 0 (bias), mode 1 (bias + exact GELU, the pre-activation kept as bfloat16), mode 2 (times gelu' of
    a bfloat16 pre-activation) with the output as fp32 or bfloat16, against fp64 on the same bf16 operands; a bfloat16 output is
    the fp64 value rounded once."""
    from vitta_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=gen).to(dev).to(torch.bfloat16)
    b = (torch.randn(n, k, generator=gen) * k ** -0.5).to(dev).to(torch.bfloat16)
    bias = torch.randn(n, generator=gen).to(dev)
    hpre = torch.randn(m, n, generator=gen).to(dev).to(torch.bfloat16)
    acc = a.double().cpu() @ b.double().cpu().t()
    rel = 2.0 ** -8 if out_bf16 else 2e-5
    _close(ops.gemm_bf16x(a, b, bias, out_bf16=out_bf16).float(), acc + bias.double().cpu(), rel=rel)
    pre = torch.full((m, n), float("nan"), device=dev).to(torch.bfloat16)
    y1 = ops.gemm_bf16x(a, b, bias, mode=1, pre=pre, out_bf16=out_bf16)
    h = acc + bias.double().cpu()
    _close(pre.float(), h, rel=2.0 ** -8)
    _close(y1.float(), F.gelu(h), rel=rel)
    hd = hpre.double().cpu().requires_grad_(True)
    F.gelu(hd).sum().backward()
    _close(ops.gemm_bf16x(a, b, None, mode=2, aux=hpre, out_bf16=out_bf16).float(), acc * hd.grad, rel=rel)
    assert y1.dtype == (torch.bfloat16 if out_bf16 else torch.float32)


@pytest.mark.parametrize("train_weights", [False, True])
def test_bf16_data_flow_mlp_and_linear_match_the_fp32_path(train_weights):
    """ops.bf16_flow: LayerNorm -> bfloat16 -> FusedMlp / DenseLinear on gemm_bf16x.hip (h, gelu(h), the MLP output and the
    gradients between them bfloat16 in memory) against the exact-fp32 kernels on the same modules: outputs and every gradient
    within the bf16 recipe's error (operands and hand-overs rounded to 8 bits: a few 1e-2 of the tensor's maximum)."""
    import torch.nn as nn
    from vitta_amd import fused_ln, ops, swin
    dev = _dev()
    torch.manual_seed(5)
    c, rows = 256, 2 * 4 * 14 * 14
    norm, mlp, lin = nn.LayerNorm(c).to(dev), swin.Mlp(c, 4 * c).to(dev), nn.Linear(c, 3 * c).to(dev)
    for p_ in list(mlp.parameters()) + list(lin.parameters()):
        p_.requires_grad_(train_weights)
    x = torch.randn(2, 4, 14, 14, c, device=dev)
    gy1, gy2 = torch.randn(2, 4, 14, 14, c, device=dev), torch.randn(2, 4, 14, 14, 3 * c, device=dev)
    res = {}
    keep = (ops.DENSE_BF16,)
    try:
        for mode in (False, True):
            ops.DENSE_BF16 = mode
            for p_ in list(norm.parameters()) + list(mlp.parameters()) + list(lin.parameters()):
                p_.grad = None
            xx = x.clone().requires_grad_(True)
            y = fused_ln.ln(norm, xx, [mlp.fc1, mlp.fc2] if mode else None)
            assert y.dtype == (torch.bfloat16 if mode else torch.float32)
            m_ = mlp(y)
            y2 = fused_ln.ln(norm, xx, [lin] if mode else None)
            q = swin.linear(lin, y2)
            assert m_.dtype == (torch.bfloat16 if mode else torch.float32) and q.dtype == torch.float32
            torch.autograd.backward([m_, q], [gy1.to(m_.dtype), gy2])
            res[mode] = (m_.detach().float(), q.detach(), xx.grad.clone(), norm.weight.grad.clone(),
                         mlp.fc1.weight.grad.clone() if train_weights else None)
    finally:
        (ops.DENSE_BF16,) = keep
    for i, (a_, b_) in enumerate(zip(res[True], res[False])):
        if a_ is None:
            continue
        assert (a_ - b_).abs().max().item() <= 3e-2 * b_.abs().max().item(), (i, (a_ - b_).abs().max().item(), b_.abs().max().item())


@pytest.mark.parametrize("m,n,k,acc,bias", [(3136, 1024, 1024, False, True), (12544, 2048, 512, True, True), (50176, 256, 768, False, False),
                                            (200704, 384, 128, True, True), (128, 128, 128, False, True), (6272, 512, 2048, True, False)])
def test_gemm_tn_bf16_weight_gradient_against_fp64(m, n, k, acc, bias):
    """gemm_tn_bf16.hip (round 6): out [N, K] (+)= g^T x with g [M, N], x [M, K] bfloat16 in memory -- the dense weight gradients of the
    bf16 recipe under SGD over all parameters (autograd's d weight / d bias of swin_transformer.py:30-35, 144, 165) -- against the
    fp64 product of the SAME bf16 values: only the fp32 accumulation order is left (4e-5 of the maximum, the bound of the other
    bf16-operand products), with and without accumulation into a live gradient, with the bias gradient (column sums of g) riding in
    the launch; every shape of Swin-B's stages incl. the 256-way token split of stage 0."""
    from vitta_amd import _lib
    from vitta_amd.ops import _p, _stream, check
    L = _lib.lib()
    assert L.vitta_gemm_tn_bf16_supported(m, n, k) == 1
    d = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(m + n + k)
    g = (torch.randn(m, n, generator=gen) * 0.5).to(d).to(torch.bfloat16)
    x = torch.randn(m, k, generator=gen).to(d).to(torch.bfloat16)
    out0 = torch.randn(n, k, generator=gen).to(d)
    out = out0.clone()
    db0 = torch.randn(n, generator=gen).to(d)
    db = db0.clone()
    need = int(L.vitta_gemm_tn_bf16_workspace_bytes(m, n, k))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=d)
    check(L.vitta_gemm_tn_bf16(_p(g), _p(x), _p(out), m, n, k, 1 if acc else 0, _p(db) if bias else None, _p(ws), need, _stream()), "vitta_gemm_tn_bf16")
    torch.cuda.synchronize()
    ref = g.double().t() @ x.double()
    want = ref + (out0.double() if acc else 0)
    err = (out.double() - want).abs().max().item()
    assert err <= 4e-5 * ref.abs().max().item() + 1e-6, (err, ref.abs().max().item())
    if bias:
        bref = g.double().sum(0) + db0.double()
        berr = (db.double() - bref).abs().max().item()
        assert berr <= 1e-5 * max(1.0, bref.abs().max().item()) * (1 + m ** 0.5 / 64), (berr, bref.abs().max().item())
    else:
        assert torch.equal(db, db0)
    # shapes it declines say so
    assert L.vitta_gemm_tn_bf16_supported(m, 174, k) == 0 and L.vitta_gemm_tn_bf16_supported(m + 8, n, k) == 0
