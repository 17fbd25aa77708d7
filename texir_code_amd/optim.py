"""torch.optim.Adam-compatible optimiser whose step is one fused HIP kernel per parameter, optionally fused with the
clamp the material trainer applies after every step (trainer/train_material.py:448-458,592-593)."""
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._clamps = {}

    def set_clamp(self, param, lo=-math.inf, hi=math.inf):
        """fuse `param.data.clamp_(lo, hi)` into every step of this parameter"""
        self._clamps[id(param)] = (float(lo), float(hi))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.TexirError("FusedAdam needs contiguous float32 CUDA parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                lo, hi = self._clamps.get(id(p), (-math.inf, math.inf))
                g = p.grad.contiguous()
                _lib.check(L.texir_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), p.numel(),
                                             float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]), lo, hi,
                                             _lib.stream_ptr()))
        return loss
