"""Video Swin Transformer (Swin-B) + I3D head + Recognizer3D, MI355X-native restatement.

Interface / state_dict / named_modules mirror of
    models/videoswintransformer_models/swin_transformer.py:18-663
    models/videoswintransformer_models/recognizer3d.py:46-115
    models/videoswintransformer_models/i3d_head.py:11-77
so mmaction-style checkpoints (`backbone.layers.2.blocks.0.attn.relative_position_bias_table`,
`cls_head.fc_cls.weight`, ...) load unchanged and choose_layers(LayerNorm)[1:] yields the same 52
layers in the same order (42 of them under layers.2 / layers.3 / backbone.norm).

MI355X-first differences:
* activations stay channels-last (B, D, H, W, C) through the whole backbone; the reference converts
  to (B, C, D, H, W) and back around every stage (`rearrange` + `.contiguous()`, swin_transformer.py:
  402,412,654-661), two full copies per stage that buy nothing;
* the attention core (scale, QK^T, relative-position bias gather, shift mask, softmax, AV) is one
  call (`window_attention`): a fused HIP kernel on the GPU, plain torch on CPU tensors;
* LayerNorm outputs are never modified in place, so the statistics hooks can read them in backward.
"""
from functools import lru_cache

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class DropPath(nn.Module):
    """Stochastic depth per sample (timm 0.6.7 semantics: bernoulli(keep) / keep)."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def active(self):
        return self.drop_prob > 0.0 and self.training

    def sample(self, batch, device):
        """Per-sample scale bernoulli(keep)/keep, shape (batch,).  SwinTransformer3D.forward draws the masks of all
        its DropPath modules in ONE launch and parks each module's row in `_next`; stand-alone use draws here."""
        queue = getattr(self, "_next", None)
        if queue:
            mask = queue.pop(0)
            if mask.shape[0] == batch and mask.device == device:
                return mask
        keep = 1.0 - self.drop_prob
        mask = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return mask

    def forward(self, x):
        if not self.active():
            return x
        return x * self.sample(x.shape[0], x.device).view((x.shape[0],) + (1,) * (x.ndim - 1))


_DP_CACHE = {}


def draw_drop_path_masks(modules, batch, device, uses=2):
    """One bernoulli + one scale launch for all DropPath modules of a forward (instead of two per use); a block
    applies its module `uses` times (attention and MLP residual).  The keep probabilities live on the device
    (cached per configuration: no host-to-device copy inside a captured graph)."""
    mods = [m for m in modules if isinstance(m, DropPath) and m.active() for _ in range(uses)]
    for m in mods:
        m._next = []
    if not mods:
        return
    key = (tuple((m.drop_prob, m.scale_by_keep) for m in mods), batch, str(device))
    if key not in _DP_CACHE:
        keep = torch.tensor([1.0 - m.drop_prob for m in mods], dtype=torch.float32).clamp_(min=0.0)
        scale = torch.tensor([1.0 / k if (k > 0 and m.scale_by_keep) else 1.0 for k, m in zip(keep.tolist(), mods)],
                             dtype=torch.float32)
        _DP_CACHE[key] = (keep.unsqueeze(1).expand(len(mods), batch).contiguous().to(device), scale.unsqueeze(1).to(device))
    keep_dev, scale_dev = _DP_CACHE[key]
    masks = torch.bernoulli(keep_dev) * scale_dev
    for m, row in zip(mods, masks):
        m._next.append(row)


def residual(x, branch, drop_path):
    """x + drop_path(branch); on the GPU one pass (vitta_scale_add_f32) for this module's own DropPath / Identity."""
    if branch.dtype != x.dtype:  # a bfloat16 branch of the bf16 data flow meeting the fp32 residual stream outside a fused pass
        branch = branch.to(x.dtype)
    if x.is_cuda and FUSED_RESIDUAL and x.dtype == torch.float32 and (x.numel() // x.shape[0]) % 4 == 0:
        from . import ops
        if isinstance(drop_path, DropPath):
            scale = drop_path.sample(x.shape[0], x.device) if drop_path.active() else None
            return ops.ResidualDropPath.apply(x, branch, scale)
        if isinstance(drop_path, nn.Identity):
            return ops.ResidualDropPath.apply(x, branch, None)
    return x + drop_path(branch)


def linear(mod, x, out_bf16=False):
    """nn.Linear on the hand-written GEMM (csrc/gemm.hip; gemm_bf16x.hip for bfloat16 activations of the bf16 data flow) for device
    activations; the module itself elsewhere.  out_bf16: the output is handed on as bfloat16 (honoured only for a bfloat16 x)."""
    if FUSED_DENSE and x.is_cuda:
        from . import ops
        if ops.dense_supported(x, mod):
            return ops.DenseLinear.apply(x, mod.weight, mod.bias, bool(out_bf16 and x.dtype == torch.bfloat16))
    if x.dtype != mod.weight.dtype:
        x = x.to(mod.weight.dtype)
    return mod(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, out_f32=False):
        """out_f32: a bfloat16 x (bf16 data flow) still gets a float32 result -- for a reader other than a fused LayerNorm pass"""
        if (FUSED_DENSE and x.is_cuda and (self.drop.p == 0.0 or not self.training) and isinstance(self.act, nn.GELU)
                and getattr(self.act, "approximate", "none") == "none"):
            from . import ops
            if ops.dense_supported(x, self.fc1, self.fc2):
                # bias + GELU in fc1's epilogue, gelu' in the epilogue of fc2's data gradient
                return ops.FusedMlp.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, out_f32)
        if x.dtype != self.fc1.weight.dtype:  # a bfloat16 x of the bf16 data flow reaching the module chain (dropout > 0, another activation)
            x = x.to(self.fc1.weight.dtype)
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


def get_window_size(x_size, window_size, shift_size=None):
    """Clamp the window to the feature size; a clamped axis is not shifted (swin_transformer.py:71-84)."""
    ws = [min(x, w) for x, w in zip(x_size, window_size)]
    if shift_size is None:
        return tuple(ws)
    ss = [0 if x <= w else s for x, w, s in zip(x_size, window_size, shift_size)]
    return tuple(ws), tuple(ss)


def window_partition(x, ws):
    """(B, D, H, W, C) -> (B * nW, wd*wh*ww, C), windows ordered (d, h, w) like the reference."""
    B, D, H, W, C = x.shape
    x = x.view(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], C)


def window_reverse(windows, ws, B, D, H, W):
    x = windows.view(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def _axis_region(size, w, s):
    """Region id along one axis of the shifted-window mask: [0, size-w) -> 0, [size-w, size-s) -> 1,
    [size-s, size) -> 2; with s == 0 the whole axis is one region (swin_transformer.py:316-323)."""
    r = torch.full((size,), 2, dtype=torch.long)
    if s > 0:
        r[:size - w] = 0
        r[size - w:size - s] = 1
    return r


@lru_cache()
def compute_mask(D, H, W, window_size, shift_size, device):
    """(nW, N, N): 0 where both tokens of a window come from the same pre-shift region, -100 elsewhere."""
    rd = _axis_region(D, window_size[0], shift_size[0])
    rh = _axis_region(H, window_size[1], shift_size[1])
    rw = _axis_region(W, window_size[2], shift_size[2])
    region = (rd[:, None, None] * 3 + rh[None, :, None]) * 3 + rw[None, None, :]
    win = window_partition(region.view(1, D, H, W, 1).float(), window_size).squeeze(-1)  # nW, N
    diff = win.unsqueeze(1) - win.unsqueeze(2)
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).to(device)


@lru_cache()
def compute_region(D, H, W, window_size, shift_size, device):
    """(nW, N) int32 pre-shift region id of every token of every window: compute_mask == -100 exactly
    where two tokens of a window differ in it.  The fused kernel derives the mask from these."""
    rd = _axis_region(D, window_size[0], shift_size[0])
    rh = _axis_region(H, window_size[1], shift_size[1])
    rw = _axis_region(W, window_size[2], shift_size[2])
    region = (rd[:, None, None] * 3 + rh[None, :, None]) * 3 + rw[None, None, :]
    return window_partition(region.view(1, D, H, W, 1), window_size).squeeze(-1).to(torch.int32).contiguous().to(device)


@lru_cache()
def compute_rowmap(D, H, W, window_size, shift_size, device):
    """(nW, N) int32: natural row d*H*W + h*W + w of token n of window w, after the cyclic shift by -shift_size and
    the window partition (swin_transformer.py:222-233).  window_reverse + the inverse roll send window outputs back
    to exactly these rows, so the fused attention kernel gathers and scatters through this map and the activations
    never leave their natural order."""
    idx = torch.arange(D * H * W).view(1, D, H, W, 1)
    if any(s > 0 for s in shift_size):
        idx = torch.roll(idx, shifts=tuple(-s for s in shift_size), dims=(1, 2, 3))
    return window_partition(idx, window_size).squeeze(-1).to(torch.int32).contiguous().to(device)


def relative_position_code(window_size):
    """code[t] with relative_position_index[q, k] == code[q] - code[k] + offset (the index is linear in
    the token coordinates): lets a kernel index the bias table without any (N, N) operand."""
    wd, wh, ww = window_size
    t = torch.arange(wd * wh * ww)
    td, th, tw = t // (wh * ww), (t // ww) % wh, t % ww
    code = (td * (2 * wh - 1) + th) * (2 * ww - 1) + tw
    offset = ((wd - 1) * (2 * wh - 1) + (wh - 1)) * (2 * ww - 1) + (ww - 1)
    return code.to(torch.int32), int(offset)


def relative_position_index(window_size):
    """(N, N) index into the (2wd-1)(2wh-1)(2ww-1) bias table (swin_transformer.py:113-124)."""
    wd, wh, ww = window_size
    coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + torch.tensor([wd - 1, wh - 1, ww - 1])
    return (rel[..., 0] * (2 * wh - 1) + rel[..., 1]) * (2 * ww - 1) + rel[..., 2]


FUSED_DENSE = True      # qkv / proj / Mlp / PatchMerging.reduction on csrc/gemm.hip instead of rocBLAS + ATen GELU
FUSED_ATTENTION = True  # tests flip this to compare the fused kernel with the composed ops on the GPU
FUSED_RESIDUAL = True   # x + DropPath(branch) as one pass
FUSED_PARTITION = True  # ... and this to compare the row-mapped kernel with roll + window_partition copies


def window_attention(qkv, bias, mask, scale, num_heads):
    """softmax(scale * Q K^T + bias (+ mask)) V per (window, head).

    qkv (B_, N, 3*C) straight from the qkv Linear; bias (nH, N, N); mask (nW, N, N) or None
    (window b uses mask[b % nW]); returns (B_, N, C) with heads concatenated (swin_transformer.py:144-168)."""
    B_, N, C3 = qkv.shape
    C = C3 // 3
    if qkv.is_cuda and FUSED_ATTENTION:
        from . import ops
        if ops.wmsa_supported(N, C // num_heads):
            return ops.WindowAttention.apply(qkv, bias, mask, scale, num_heads)
        # window sizes the fused kernel does not cover (N > 400, e.g. (16,7,7)) take the composed form
        # below: still GPU library kernels, no host fallback
    if qkv.is_cuda:
        from ._lib import loud_once
        loud_once(("wmsa_composed", N), f"window attention with a dense bias on {N} tokens runs as composed library products (matmul / "
                  f"softmax), not on wmsa.hip" + ("" if FUSED_ATTENTION else ": swin.FUSED_ATTENTION is off (an A/B switch)"))
    q, k, v = qkv.view(B_, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, num_heads, N, N)
    attn = attn.softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B_, N, C)


class WindowAttention3D(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        if attn_drop != 0.0:
            raise NotImplementedError("attention dropout is 0 on the ViTTA path (recognizer3d.py:61)")
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        table = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) * (2 * window_size[2] - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(table, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        code, self.code_offset = relative_position_code(window_size)
        assert torch.equal(code[:, None].long() - code[None, :].long() + self.code_offset, self.relative_position_index)
        self.register_buffer("relative_position_code", code, persistent=False)  # not part of checkpoints
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x, mask=None, region=None):
        B_, N, C = x.shape
        if x.is_cuda and FUSED_ATTENTION and (mask is None or region is not None):
            from . import ops
            if ops.wmsa_rel_supported(N, C // self.num_heads, self.relative_position_bias_table.shape[0]):
                out = ops.WindowAttentionRel.apply(linear(self.qkv, x), self.relative_position_bias_table,
                                                   self.relative_position_code[:N], self.code_offset, region, self.scale,
                                                   self.num_heads)
                return self.proj_drop(linear(self.proj, out))
        if x.is_cuda:
            from ._lib import loud_once
            loud_once(("wmsa_rel_declined", N), f"relative-position window attention on {N} tokens takes the gathered dense-bias form"
                      + ("" if FUSED_ATTENTION else ": swin.FUSED_ATTENTION is off (an A/B switch)"))
        idx = self.relative_position_index[:N, :N].reshape(-1)
        bias = self.relative_position_bias_table[idx].view(N, N, self.num_heads).permute(2, 0, 1).contiguous()
        out = window_attention(linear(self.qkv, x), bias, mask, self.scale, self.num_heads)
        return self.proj_drop(linear(self.proj, out))


class SwinTransformerBlock3D(nn.Module):
    def __init__(self, dim, num_heads, window_size=(2, 7, 7), shift_size=(0, 0, 0), mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 use_checkpoint=False):
        super().__init__()
        assert all(0 <= s < w for s, w in zip(shift_size, window_size)), "shift_size must in 0-window_size"
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.mlp_ratio, self.use_checkpoint = mlp_ratio, use_checkpoint
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention3D(dim, window_size=window_size, num_heads=num_heads, qkv_bias=qkv_bias,
                                      qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def attention_branch(self, x, mask_matrix, region=None):
        """x: the NORMALISED block input (norm1 already applied)."""
        B, D, H, W, C = x.shape
        ws, ss = get_window_size((D, H, W), self.window_size, self.shift_size)
        pad = [(w - n % w) % w for n, w in zip((D, H, W), ws)]
        if any(pad):
            x = F.pad(x, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
        Dp, Hp, Wp = D + pad[0], H + pad[1], W + pad[2]
        shifted = any(s > 0 for s in ss)
        attn = self.attn
        n_tok = ws[0] * ws[1] * ws[2]
        if x.is_cuda and FUSED_ATTENTION and FUSED_PARTITION and not any(pad) and (not shifted or region is not None):
            from . import ops
            if ops.wmsa_rel_supported(n_tok, C // attn.num_heads, attn.relative_position_bias_table.shape[0]):
                # shift + partition + reverse + inverse shift as address arithmetic inside the attention kernel
                rowmap = compute_rowmap(D, H, W, ws, ss, x.device)
                # bf16 data flow: qkv, the context and the branch leave their producers as bfloat16
                io16 = x.dtype == torch.bfloat16 and ops.wmsa_io16_ok(n_tok, C // attn.num_heads, attn.relative_position_bias_table) \
                    and ops.bf16_dense_ok(B * D * H * W, attn.proj)
                out = ops.WindowAttentionRel.apply(linear(attn.qkv, x.view(B, D * H * W, C), io16), attn.relative_position_bias_table,
                                                   attn.relative_position_code[:n_tok], attn.code_offset,
                                                   region if shifted else None, attn.scale, attn.num_heads, rowmap)
                return attn.proj_drop(linear(attn.proj, out, io16)).view(B, D, H, W, C)
        if shifted:
            x = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        windows = self.attn(window_partition(x, ws), mask=mask_matrix if shifted else None,
                            region=region if shifted else None)
        x = window_reverse(windows, ws, B, Dp, Hp, Wp)
        if shifted:
            x = torch.roll(x, shifts=ss, dims=(1, 2, 3))
        if any(pad):
            x = x[:, :D, :H, :W, :].contiguous()
        return x

    def forward(self, x, mask_matrix, region=None, normed=None, next_norm=None, next_qkv=None):
        """`normed`: norm1(x) if the caller already has it; `next_norm`: the LayerNorm that consumes this block's
        output (the next block's norm1) -- then the closing residual update and that normalisation are one pass and
        the block returns (x_out, next_norm(x_out)) instead of x_out."""
        from .fused_ln import ln_pass, ln_residual
        if normed is None:  # (x comes back as the handle for the residual update below: one consumer of the block input)
            x, normed = ln_pass(self.norm1, x, [self.attn.qkv])
        a = self.attention_branch(normed, mask_matrix, region)
        x, y = ln_residual(self.norm2, x, a, self.drop_path, [self.mlp.fc1, self.mlp.fc2])  # x = x + drop_path(a); y = norm2(x): one pass
        m = self.mlp(y, out_f32=next_norm is None) if y.dtype == torch.bfloat16 else self.mlp(y)
        if next_norm is not None:
            return ln_residual(next_norm, x, m, self.drop_path, next_qkv)
        return residual(x, m, self.drop_path)


class PatchGather(torch.autograd.Function):
    """cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1) of PatchMerging
    (swin_transformer.py:281-286) for even H, W as ONE autograd node.  Composed of slices, autograd differentiates it as eight
    slice_backward nodes (a zero fill + a strided copy of half- and full-size tensors each) and three accumulation adds over the
    stage's whole output -- 2.1 GB of traffic behind the first PatchMerging of the 4 x 32-frame step, 0.8 ms per video over the three;
    the gradient is simply the four channel groups written back to their pixel parities: four strided copies, every byte once."""

    HIP = __import__("os").environ.get("VITTA_PATCH_GATHER", "1") != "0"  # 0: the strided copies of torch (A/B)

    @staticmethod
    def _hip(t):
        return PatchGather.HIP and t.is_cuda and t.dtype == torch.float32 and t.shape[-1] % 4 == 0

    @staticmethod
    def forward(ctx, x):
        if PatchGather._hip(x) and x.shape[-1] % 4 == 0:  # ONE launch (cat of four strided slices: four copy kernels)
            from . import ops
            x = x.contiguous()
            B, D, H, W, C = x.shape
            out = torch.empty(B, D, H // 2, W // 2, 4 * C, dtype=x.dtype, device=x.device)
            ops.check(ops.lib().vitta_patch_gather_f32(ops._p(x), ops._p(out), B * D, H // 2, W // 2, C, 0, ops._stream()),
                      "vitta_patch_gather_f32")
            return out
        return torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)

    @staticmethod
    def backward(ctx, g):
        B, D, H2, W2, C4 = g.shape
        C = C4 // 4
        gx = torch.empty(B, D, 2 * H2, 2 * W2, C, dtype=g.dtype, device=g.device)
        if PatchGather._hip(g) and C % 4 == 0:
            from . import ops
            g = g.contiguous()
            ops.check(ops.lib().vitta_patch_gather_f32(ops._p(g), ops._p(gx), B * D, H2, W2, C, 1, ops._stream()), "vitta_patch_gather_f32")
            return gx
        gx[:, :, 0::2, 0::2] = g[..., 0:C]
        gx[:, :, 1::2, 0::2] = g[..., C:2 * C]
        gx[:, :, 0::2, 1::2] = g[..., 2 * C:3 * C]
        gx[:, :, 1::2, 1::2] = g[..., 3 * C:]
        return gx


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        B, D, H, W, C = x.shape
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = PatchGather.apply(x)
        from .fused_ln import ln
        return linear(self.reduction, ln(self.norm, x, [self.reduction]))


class BasicLayer(nn.Module):
    """One stage; input and output are channels-last (B, D, H, W, C)."""

    def __init__(self, dim, depth, num_heads, window_size=(1, 7, 7), mlp_ratio=4.0, qkv_bias=False, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size = window_size
        self.shift_size = tuple(i // 2 for i in window_size)
        self.depth, self.use_checkpoint = depth, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim=dim, num_heads=num_heads, window_size=window_size,
                                   shift_size=(0, 0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio=mlp_ratio,
                                   qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                   drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                   norm_layer=norm_layer, use_checkpoint=use_checkpoint)
            for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        B, D, H, W, C = x.shape
        ws, ss = get_window_size((D, H, W), self.window_size, self.shift_size)
        Dp, Hp, Wp = (int(np.ceil(n / w)) * w for n, w in zip((D, H, W), ws))
        attn_mask = compute_mask(Dp, Hp, Wp, ws, ss, x.device)
        region = compute_region(Dp, Hp, Wp, ws, ss, x.device) if x.is_cuda else None
        normed = None
        for i, blk in enumerate(self.blocks):
            nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None
            out = blk(x, attn_mask, region, normed=normed, next_norm=nxt,
                      next_qkv=[self.blocks[i + 1].attn.qkv] if nxt is not None else None)
            x, normed = out if nxt is not None else (out, None)
        return self.downsample(x) if self.downsample is not None else x


class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size=(2, 4, 4), in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size, self.in_chans, self.embed_dim = patch_size, in_chans, embed_dim
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        """(B, 3, T, H, W) -> channels-last tokens (B, D, H', W', C)."""
        _, _, D, H, W = x.size()
        p = self.patch_size
        if W % p[2] or H % p[1] or D % p[0]:
            x = F.pad(x, (0, (-W) % p[2], 0, (-H) % p[1], 0, (-D) % p[0]))
        B, Cin, D, H, W = x.shape
        kin = Cin * p[0] * p[1] * p[2]
        from . import ops as _ops
        if (FUSED_DENSE and x.is_cuda and x.dtype == torch.float32 and kin % 32 == 0 and tuple(self.proj.stride) == tuple(p)
                and tuple(self.proj.kernel_size) == tuple(p) and tuple(self.proj.padding) == (0, 0, 0)
                and tuple(self.proj.dilation) == (1, 1, 1) and self.proj.groups == 1 and not self.proj._forward_hooks
                and not self.proj._forward_pre_hooks
                and _ops.gemm_nt_supported(B * (D // p[0]) * (H // p[1]) * (W // p[2]), self.proj.weight.shape[0], kin)):
            # kernel == stride: the convolution is a per-patch Linear(kin -> embed_dim).  One gather copy + the dense kernel
            # (csrc/gemm.hip) instead of the library's im2col + GEMM -- and instead of its NAIVE Conv3d weight-gradient
            # kernel under SGD over all parameters (7.9 ms per video, 18 % of that step)
            from . import ops
            Dd, Hh, Ww = D // p[0], H // p[1], W // p[2]
            patches = x.reshape(B, Cin, Dd, p[0], Hh, p[1], Ww, p[2]).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, Dd * Hh * Ww, kin)
            x = ops.DenseLinear.apply(patches, self.proj.weight, self.proj.bias)
            C = x.shape[-1]
        else:
            x = self.proj(x)
            B, C, Dd, Hh, Ww = x.shape
            x = x.flatten(2).transpose(1, 2)  # (B, L, C): the first LayerNorm sees a 3-D tensor, as in the reference
        if self.norm is not None:
            from .fused_ln import ln
            x = ln(self.norm, x)  # (the one-pass kernel; torch's layer norm was 0.6 ms per video at config 5's 200 704 tokens)
        return x.reshape(B, Dd, Hh, Ww, C)


class SwinTransformer3D(nn.Module):
    """forward returns channels-last (B, D, H, W, C) features (the reference returns (B, C, D, H, W))."""

    def __init__(self, pretrained=None, pretrained2d=True, patch_size=(4, 4, 4), in_chans=3, embed_dim=96,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=(2, 7, 7), mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2, norm_layer=nn.LayerNorm,
                 patch_norm=False, frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        self.num_layers, self.embed_dim, self.patch_norm = len(depths), embed_dim, patch_norm
        self.frozen_stages, self.window_size, self.patch_size = frozen_stages, window_size, patch_size
        self.patch_embed = PatchEmbed3D(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                        norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint))
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.norm = norm_layer(self.num_features)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x):
        x = self.pos_drop(self.patch_embed(x))
        if x.is_cuda and self.training:
            draw_drop_path_masks([blk.drop_path for layer in self.layers for blk in layer.blocks], x.shape[0], x.device)
        for layer in self.layers:
            x = layer(x)
        from .fused_ln import ln
        return ln(self.norm, x)


class I3DHead(nn.Module):
    """avg-pool over (D, H, W) -> dropout -> fc (i3d_head.py:57-77) on channels-last features."""

    def __init__(self, num_classes, in_channels, spatial_type="avg", dropout_ratio=0.5, init_std=0.01):
        super().__init__()
        self.num_classes, self.in_channels, self.spatial_type = num_classes, in_channels, spatial_type
        self.dropout_ratio, self.init_std = dropout_ratio, init_std
        self.dropout = nn.Dropout(p=dropout_ratio) if dropout_ratio != 0 else None
        self.fc_cls = nn.Linear(in_channels, num_classes)
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1)) if spatial_type == "avg" else None
        nn.init.normal_(self.fc_cls.weight, 0, init_std)
        nn.init.constant_(self.fc_cls.bias, 0)

    def forward(self, x):
        if self.avg_pool is not None:
            x = x.mean(dim=(1, 2, 3))  # channels-last: pool the three middle axes
        else:
            x = x.reshape(x.shape[0], -1)
        if self.dropout is not None:
            x = self.dropout(x)
        from . import fused_ln, ops
        if fused_ln.ENABLED and ops.head_linear_supported(x, self.fc_cls):  # head.hip: no library GEMM for [views, 1024] x [1024, K]
            return ops.HeadLinear.apply(x, self.fc_cls.weight, self.fc_cls.bias)
        return self.fc_cls(x)


class Recognizer3D(nn.Module):
    """Swin-B recognizer: forward(x[B, V, 3, T, H, W]) -> (video scores [B, K], per-view scores [B, V, K])."""

    def __init__(self, num_classes=None, patch_size=None, window_size=None, drop_path_rate=None,
                 embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)):
        super().__init__()
        self.patch_size, self.window_size, self.drop_path_rate = patch_size, window_size, drop_path_rate
        self.embed_dim, self.depths, self.num_heads = embed_dim, list(depths), list(num_heads)
        self.num_classes, self.in_channels = num_classes, int(embed_dim * 2 ** (len(depths) - 1))
        self.score_type = "score"
        self.backbone = SwinTransformer3D(patch_size=patch_size, in_chans=3, embed_dim=embed_dim, depths=self.depths,
                                          num_heads=self.num_heads, window_size=window_size, mlp_ratio=4.0,
                                          qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                                          drop_path_rate=drop_path_rate, patch_norm=True)
        self.cls_head = I3DHead(num_classes=num_classes, in_channels=self.in_channels, spatial_type="avg",
                                dropout_ratio=0.5)

    def forward(self, x):
        n_views = x.shape[1]
        feat = self.backbone(x.reshape((-1,) + x.shape[2:]))
        return self.average_clips(self.cls_head(feat), num_segs=n_views)

    def average_clips(self, cls_score, num_segs=1):
        cls_score = cls_score.view(cls_score.shape[0] // num_segs, num_segs, -1)
        if self.score_type == "prob":
            return F.softmax(cls_score, dim=2).mean(dim=2)
        return cls_score.mean(dim=1), cls_score
