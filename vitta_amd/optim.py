"""Optimizers of the TTA step on the flat parameter arena: one HIP launch per step.

`corpus/basics.py:547-560` builds `torch.optim.Adam(affine params, lr, betas=(0.9, 0.999), weight_decay=0)` under
`--update_only_bn_affine` and `torch.optim.SGD(all params, lr, momentum, weight_decay)` otherwise; `:671` steps it.
Both are element-wise, so over `tta.FlatArena` (every trainable tensor a view of one buffer) the step is
`vitta_adam_step_f32` / `vitta_sgd_step_f32` on that buffer: same arithmetic as torch's single-tensor
formulation, 1-2 launches instead of 25 (capturable foreach Adam) -- and capturable in a hipGraph by construction
(Adam's step counter is a device scalar the kernel call advances).

The objects expose the small part of the torch.optim surface the reference's loop and checkpointing touch:
`step()`, `zero_grad()`, `param_groups`, `state_dict()`.
"""
import torch

from . import _lib
from .ops import _p, _stream, check, lib


class _FlatOptimizer:
    def __init__(self, arena, defaults):
        flat = arena.flat_param
        if flat.device.type != "cuda" or flat.dtype != torch.float32:
            raise _lib.VittaHipError("the fused optimizers step CUDA(HIP) fp32 arenas; there is no CPU fallback")
        self.arena = arena
        self.defaults = dict(defaults)
        self.param_groups = [dict(defaults, params=[flat])]
        self.state = {}

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def state_dict(self):
        return {"state": {0: {k: v for k, v in self.state.items()}},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} | {"params": [0]} for g in self.param_groups]}


class FlatAdam(_FlatOptimizer):
    def __init__(self, arena, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(arena, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        flat = arena.flat_param
        self._ticket = torch.zeros(1, dtype=torch.int32, device=flat.device)  # arrival counter of the launch (zero at rest)
        self.state = {"step": torch.zeros(1, dtype=torch.float32, device=flat.device),
                      "exp_avg": torch.zeros_like(flat), "exp_avg_sq": torch.zeros_like(flat)}

    @torch.no_grad()
    def step(self):
        g, st, flat = self.param_groups[0], self.state, self.arena.flat_param
        step = st["step"]  # (may have been replaced by a loaded state: any one-element fp32 device tensor)
        if step.numel() != 1 or step.dtype != torch.float32 or step.device != flat.device:
            raise _lib.VittaHipError("FlatAdam.state['step'] must be a one-element float32 tensor on the arena's device")
        check(lib().vitta_adam_step_f32(_p(flat), _p(self.arena.grad), _p(st["exp_avg"]), _p(st["exp_avg_sq"]),
                                        _p(step), _p(self._ticket), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                        float(g["eps"]), float(g["weight_decay"]), flat.numel(), _stream()),
              "vitta_adam_step_f32")


class FlatSGD(_FlatOptimizer):
    def __init__(self, arena, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(arena, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False))
        flat = arena.flat_param
        self.state = {"momentum_buffer": torch.zeros_like(flat) if momentum != 0 else None}

    @torch.no_grad()
    def step(self):
        g, flat = self.param_groups[0], self.arena.flat_param
        check(lib().vitta_sgd_step_f32(_p(flat), _p(self.arena.grad), _p(self.state["momentum_buffer"]), float(g["lr"]),
                                       float(g["momentum"]), float(g["weight_decay"]), flat.numel(), _stream()),
              "vitta_sgd_step_f32")
