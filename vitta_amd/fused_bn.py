"""`bn_act(bn, x, residual=None, relu=True)`: eval-mode BatchNorm2d (+ residual add) (+ ReLU) as one HIP
pass, with the ViTTA statistics of a hooked layer riding on it (vitta_amd/csrc/bn_act.hip).

Semantics are exactly `relu(bn(x) + residual)` of the module calls it replaces, INCLUDING forward hooks:
* a layer carrying a CombineNormStatsRegHook_onereg bound to the batched engine takes the fused path once
  the engine has a launch plan for the current shapes (from the second step on); its hook does not fire
  as a Python callback, the kernel deposits the layer's partial moments straight into the plan workspace
  and the backward injects the statistics gradient;
* any other forward hook on the module (stand-alone statistics hooks, ComputeNormStatsHook, user hooks),
  training-mode BN, CPU tensors or shapes without a 16-byte path -> the plain module calls.
"""
import torch
import torch.nn as nn

ENABLED = True  # tests / bench flip this to compare with the unfused module calls


def _engine_hook(bn):
    """(fusable, hook): hook = the engine-bound statistics hook of this module or None; fusable False if
    the module carries any forward hook this path cannot honour."""
    hook = None
    for fn in bn._forward_hooks.values():
        owner = getattr(fn, "__self__", None)
        if owner is not None and getattr(owner, "engine", None) is not None and hasattr(owner, "index") and hook is None:
            hook = owner
        else:
            return False, None
    return True, hook


TWIN = "_vitta_twin"


def identity_source(x):
    """The handle of a block input the identity / downsample path should read: the twin left by `bn_act(...,
    fork=True)` of the producing block (same storage, separate gradient), or x itself."""
    return getattr(x, TWIN, x)


def bn_act(bn, x, residual=None, relu=True, act=None, fork=False):
    """`fork=True`: the result feeds two consumers (the next block's conv1 and its identity / downsample path).
    On the fused path it then carries a twin handle (see identity_source) so that the two gradients reach the
    backward kernel separately instead of through an autograd add."""
    if ENABLED and isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.affine and not bn._forward_pre_hooks:
        from . import ops
        if ops.bn_act_supported(x):
            fusable, hook = _engine_hook(bn)
            if fusable:
                site = None
                if hook is not None and hook.kind == "bn2d" and not hook.before_norm:
                    site = hook.engine.fused_site(hook.index, x)
                if hook is None or site is not None:
                    if fork and torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad):
                        z, twin = ops.FusedBNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                                       residual, relu, site, True)
                        setattr(z, TWIN, twin)
                        return z
                    return ops.FusedBNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                                residual, relu, site)
    out = bn(x)
    if residual is not None:
        out = out + residual
    if relu:
        out = act(out) if act is not None else torch.relu(out)
    return out
