"""torch.optim.Adam-compatible optimiser whose step is one fused HIP kernel per parameter, optionally fused with the
clamp the material trainer applies after every step (trainer/train_material.py:448-458,592-593)."""
import math
import os

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """fuse_mip_fold=True: the [H,W,C] texture parameters ask texture.py's backward to leave the last mip fold (level 1 -> level 0) out
    and this optimiser adds 0.25 * level-1 gradient while it reads the gradient (bit-identical to fold + step; saves one
    read-modify-write of every texture per step).  Until step() has run, `p.grad` then lacks that term: code that reduces gradients
    across ranks in between must treat the pair (p.grad, p._texir_grad_l1), as dist_util.reduce_texture_grads does."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fuse_mip_fold=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._clamps = {}
        self.fuse_mip_fold = bool(fuse_mip_fold)
        for group in self.param_groups:
            for p in group["params"]:
                p._texir_defer_fold = self.fuse_mip_fold and p.dim() == 3 and p.shape[0] % 2 == 0 and p.shape[1] % 2 == 0
                p._texir_defer_levels = int(os.environ.get("TEXIR_DEFER_LEVELS", "2"))      # folds left to the step: level 1 -> 0, and (2) level 2 -> 1 as well
                p._texir_grad_l1 = p._texir_grad_l2 = None
                p._texir_l0_touched = False
        self._make_grad_arena()

    def _make_grad_arena(self):
        """ONE buffer for the mip-level gradient stacks of all deferring texture parameters (per device), so that a step clears them with
        a single fill (at the step's first trainable fetch, texture.texture) instead of one per parameter.  A slot is sized for the full pyramid of its texture (levels
        1 .. 1x1: one third of the texture); the backward uses the leading part its fetch's level count needs."""
        by_dev = {}
        for group in self.param_groups:
            for p in group["params"]:
                p._texir_arena = None
                if p._texir_defer_fold and p.is_cuda and p.dtype == torch.float32:
                    by_dev.setdefault(p.device, []).append(p)
        for dev, ps in by_dev.items():
            sizes = []
            for p in ps:
                H, W, C = p.shape
                n, h, w = 0, H // 2, W // 2
                while h >= 1 and w >= 1:
                    n += h * w * C
                    h, w = h // 2, w // 2
                sizes.append((n + 63) // 64 * 64)
            buf = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
            arena = {"buf": buf, "params": ps, "clean": set()}       # clean: ids of the parameters whose span is known to be zero
            off = 0
            for p, n in zip(ps, sizes):
                p._texir_arena, p._texir_arena_span = arena, (off, off + n)
                off += n

    def zero_grad(self, set_to_none=True):
        for group in self.param_groups:
            for p in group["params"]:
                p._texir_grad_l1 = p._texir_grad_l2 = None
                p._texir_l0_touched = False
                p._texir_l0_mask = None
                p._texir_l0_sparse = False
        super().zero_grad(set_to_none=set_to_none)

    def release(self):
        """stop deferring (call before handing the parameters to another optimiser that does not fuse the fold)"""
        for group in self.param_groups:
            for p in group["params"]:
                p._texir_defer_fold = False
                p._texir_grad_l1 = p._texir_grad_l2 = None
                p._texir_arena = None

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
                g1 = getattr(p, "_texir_grad_l1", None)
                if p.grad is None and g1 is None:
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
                g = None if p.grad is None else p.grad.contiguous()      # None: level-0 gradient identically zero (texture.py backward)
                mask = None
                if g1 is not None and getattr(p, "_texir_l0_sparse", False):
                    # texture.py's sparse level-0 gradient: valid only at the texels of the view's bit mask, in the parameter's own buffer
                    mask = getattr(p, "_texir_l0_mask", None)
                    if mask is not None:
                        if g is None:
                            g = p._texir_g0
                        else:
                            # (another fetch of the same parameter also produced a dense gradient in this backward pass: fold the sparse part in)
                            H, W, C = p.shape
                            bits = ((mask.view(-1, 1).to(torch.int64) >> torch.arange(32, device=p.device)) & 1).bool().reshape(-1)[:H * W].reshape(H, W)
                            g[bits] += p._texir_g0[bits]
                            mask = None
                if g1 is not None:
                    H, W, C = p.shape
                    # level 1 of the next forward's mip stack is written on the way (texture._mips_for then builds levels 2.. only)
                    mips = getattr(p, "_texir_mips", None)
                    mip1 = mips[1] if (mips is not None and mips[1].numel() >= (H // 2) * (W // 2) * C and mips[1].device == p.device) else None
                    _lib.check(L.texir_adam_step_tex(_lib.ptr(p), _lib.ptr(g), _lib.ptr(mask), _lib.ptr(g1), _lib.ptr(getattr(p, "_texir_grad_l2", None)),
                                                     _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]),
                                                     _lib.ptr(mip1), H, W, C, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                     int(st["step"]), lo, hi, _lib.stream_ptr()))
                    p._texir_mip1_version = (p.data_ptr(), p._version) if mip1 is not None else None
                    if not getattr(p, "_texir_l1_static", False):      # (hipGraph replay re-fills the same buffer: keep it)
                        p._texir_grad_l1 = p._texir_grad_l2 = None
                else:
                    p._texir_mip1_version = None                       # the texture changes behind the mip stack's back
                    _lib.check(L.texir_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), p.numel(),
                                                 float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]), lo, hi,
                                                 _lib.stream_ptr()))
        return loss
