"""`ln(norm, x)` / `ln_residual(norm, x, branch, drop_path)`: LayerNorm over channels-last activations as one HIP pass
(vitta_amd/csrc/layernorm.hip), optionally fused with the residual update + stochastic depth that precedes it, with
the ViTTA statistics of a hooked layer riding on the pass and the statistics-loss gradient added in its backward.

Semantics are exactly `norm(x)` (resp. `x = x + drop_path(branch); norm(x)`) of the module calls they replace,
INCLUDING forward hooks -- the same contract as fused_bn.bn_act:
* a LayerNorm carrying a CombineNormStatsRegHook_onereg bound to the batched engine takes the fused path once the
  engine has a plan for the current shapes and every hooked layer of the model is covered; its hook then does not
  fire as a Python callback;
* any other hook, unsupported width, CPU tensors -> the plain module calls.
"""
import torch
import torch.nn as nn

from .fused_bn import _engine_hook

ENABLED = True  # tests flip this to compare with the unfused module calls


def _site_for(norm, x):
    """(fusable, site)"""
    if not (ENABLED and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None
            and len(norm.normalized_shape) == 1 and x.is_cuda and x.dtype == torch.float32
            and x.shape[-1] == norm.normalized_shape[0] and not norm._forward_pre_hooks):
        return False, None
    from . import ops
    if not ops.ln_supported(x.shape[-1]):
        return False, None
    fusable, hook = _engine_hook(norm)
    if not fusable:
        return False, None
    if hook is None:
        return True, None
    if hook.kind != "ln" or hook.before_norm:
        return False, None
    site = hook.engine.fused_ln_site(hook.index, x)
    return site is not None, site


def _y16(x, consumers):
    """The fused pass may write its output as bfloat16: the bf16 data flow is on and every module of `consumers` (the dense layers
    the output feeds, in a chain) runs on gemm_bf16x.hip for this row count."""
    if not consumers:
        return False
    from . import ops
    return ops.bf16_dense_ok(x.numel() // x.shape[-1], *consumers)


def ln(norm, x, consumers=None):
    """consumers: the nn.Linear chain that is the ONLY reader of the output (qkv; fc1, fc2; a PatchMerging reduction)."""
    fusable, site = _site_for(norm, x)
    if fusable:
        from . import ops
        return ops.FusedLayerNorm.apply(x, None, None, norm.weight, norm.bias, norm.eps, site, _y16(x, consumers))
    return norm(x)


def ln_pass(norm, x, consumers=None):
    """(x, norm(x)) where the returned x is the handle the caller's OTHER reader of x must use (ops.FusedLayerNorm passthrough):
    the block's input then has one consumer in the autograd graph."""
    fusable, site = _site_for(norm, x)
    if fusable and x.requires_grad and torch.is_grad_enabled():
        from . import ops
        return ops.FusedLayerNorm.apply(x, None, None, norm.weight, norm.bias, norm.eps, site, _y16(x, consumers), True)
    return x, ln(norm, x, consumers)


def ln_residual(norm, x, branch, drop_path, consumers=None):
    """(x', norm(x')) with x' = x + drop_path(branch)."""
    from .swin import DropPath, residual
    fusable, site = _site_for(norm, x)
    if fusable:
        from . import ops
        y16 = _y16(x, consumers)
        if isinstance(drop_path, (DropPath, nn.Identity)):
            scale = drop_path.sample(x.shape[0], x.device) if isinstance(drop_path, DropPath) and drop_path.active() else None
            return ops.FusedLayerNorm.apply(x, branch, scale, norm.weight, norm.bias, norm.eps, site, y16)
        # a foreign stochastic-depth module (e.g. a test's mask replay): its own call, then the fused norm -- a step is
        # all-fused or all-recorded, so the LayerNorm must not drop back to the module call here
        x = x + drop_path(branch).to(x.dtype)
        return x, ops.FusedLayerNorm.apply(x, None, None, norm.weight, norm.bias, norm.eps, site, y16)
    x = residual(x, branch, drop_path)
    return x, norm(x)
