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
        self._dev = {}                 # per device: {"state": float64 [n,4], "hyper": float32 [n,2], "lr": [(lr, beta1, beta2) last written]}; record index per param
        self._rec = {}

    def add_param_group(self, param_group):
        # (torch.optim.Optimizer.__init__ adds the constructor's groups through this method, before the attributes above exist)
        if hasattr(self, "_rec"):
            raise _lib.TexirError("FusedAdam: parameter groups are fixed at construction (the gradient arena and the device-resident step records "
                                  "are laid out over them); build a new optimiser instead")
        super().add_param_group(param_group)

    # ---- device-resident step count / learning rate (texir_adam_tick): the step needs no host argument that changes from step to step,
    # so a captured hipGraph can contain it (graph_step.GraphedMatStep) and eager steps run the very same kernels ----
    def _records(self, device):
        d = self._dev.get(device)
        if d is None:
            ps = [(g, p) for g in self.param_groups for p in g["params"] if p.device == device]
            if len(ps) > 64:
                raise _lib.TexirError("FusedAdam: at most 64 parameters per device")
            if torch.cuda.is_current_stream_capturing():
                raise _lib.TexirError("FusedAdam state must exist before hipGraph capture (call prepare() first)")
            host = torch.zeros((len(ps), 4), dtype=torch.float64)
            for i, (g, p) in enumerate(ps):
                self._rec[id(p)] = i
                host[i] = torch.tensor([float(self.state[p].get("step", 0)) if self.state[p] else 0.0, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1])], dtype=torch.float64)
            d = {"state": host.to(device), "hyper": torch.zeros((len(ps), 2), device=device, dtype=torch.float32),
                 "lr": [(float(g["lr"]), float(g["betas"][0]), float(g["betas"][1])) for g, _ in ps],
                 "params": [p for _, p in ps], "groups": [g for g, _ in ps]}
            self._dev[device] = d
        return d

    def prepare(self):
        """allocate everything a step touches (moments, device records) and push learning-rate / beta changes to the device: call before hipGraph
        capture and before every replay (cheap: compares floats, launches something only when a scheduler has changed a learning rate).
        (A recorded step keeps the betas of its capture in its kernel arguments: re-capture after changing betas.)"""
        for group in self.param_groups:
            for p in group["params"]:
                if p.is_cuda and p.requires_grad:            # (frozen members of the model -- the irradiance texture -- get no moments)
                    self._ensure_state(p)
        for device in {p.device for g in self.param_groups for p in g["params"] if p.is_cuda}:
            d = self._records(device)
            for i, g in enumerate(d["groups"]):
                now = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]))
                if now != d["lr"][i]:
                    d["state"][i, 1:4] = torch.tensor(now, dtype=torch.float64)
                    d["lr"][i] = now

    def _ensure_state(self, p):
        st = self.state[p]
        if not st:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.TexirError("FusedAdam moments must exist before hipGraph capture (call prepare() first)")
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        return st

    def load_state_dict(self, sd):
        """Captured hipGraphs have the ADDRESSES of the moments and of the device-resident step / learning-rate records baked into their kernel
        nodes, so loading must not replace those buffers: the loaded values are copied INTO the existing tensors (same shapes), and the device
        records are overwritten in place from the loaded step counts.  Graphs recorded before the load therefore stay valid and continue from
        the loaded state."""
        old = {p: dict(st) for p, st in self.state.items()}
        super().load_state_dict(sd)
        with torch.no_grad():
            for p, st in self.state.items():
                was = old.get(p)
                if not was:
                    continue
                for k in ("exp_avg", "exp_avg_sq"):
                    a, b = was.get(k), st.get(k)
                    if torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.device == b.device and a.dtype == b.dtype:
                        a.copy_(b)
                        st[k] = a
            group_of = {id(q): g for g in self.param_groups for q in g["params"]}      # (load_state_dict installs NEW group dicts)
            for device, d in self._dev.items():
                d["groups"] = [group_of[id(q)] for q in d["params"]]
                host = torch.zeros(tuple(d["state"].shape), dtype=torch.float64)
                for i, (g, p) in enumerate(zip(d["groups"], d["params"])):
                    st = self.state.get(p) or {}
                    step = st.get("step", 0)
                    host[i] = torch.tensor([float(step), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1])], dtype=torch.float64)
                    d["lr"][i] = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]))
                d["state"].copy_(host)

    def note_replayed_step(self, params):
        """a hipGraph replay has run the recorded step of `params`: advance the host mirrors of what the device did"""
        for p in params:
            self.state[p]["step"] += 1
            if getattr(p, "_texir_mips", None) is not None and getattr(p, "_texir_defer_fold", False):
                p._texir_mip1_fresh = (p.data_ptr(), p._version)

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
                p._texir_l1_zero = False
        super().zero_grad(set_to_none=set_to_none)

    def release(self):
        """stop deferring (call before handing the parameters to another optimiser that does not fuse the fold)"""
        for group in self.param_groups:
            for p in group["params"]:
                p._texir_defer_fold = False
                p._texir_grad_l1 = p._texir_grad_l2 = None
                p._texir_arena = None

    def has_pending(self, p):
        """True while a backward pass has left parts of p's gradient outside p.grad (parked mip-level stacks / the sparse level-0 buffer):
        until step() has consumed them, p.grad alone is NOT the gradient -- use dense_grad(p) for norms, clipping or logging, and do not
        step such a parameter with another optimiser (call release() first)."""
        return getattr(p, "_texir_grad_l1", None) is not None

    @torch.no_grad()
    def dense_grad(self, p):
        """the full gradient of p that step() is about to apply, materialised as one dense tensor (p.grad + the sparse level-0 part under
        its bit mask + the parked level-1 / level-2 stacks folded down: each coarser texel adds a quarter of its value to its four children)"""
        g = torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()
        g1 = getattr(p, "_texir_grad_l1", None)
        if g1 is None:
            return g
        H, W, C = p.shape
        if getattr(p, "_texir_l0_sparse", False) and getattr(p, "_texir_l0_mask", None) is not None:
            mask = p._texir_l0_mask
            bits = ((mask.view(-1, 1).to(torch.int64) >> torch.arange(32, device=p.device)) & 1).bool().reshape(-1)[:H * W].reshape(H, W)
            g[bits] += p._texir_g0[bits]
        up = lambda t: t.repeat_interleave(2, 0).repeat_interleave(2, 1)
        l1 = g1.view(H // 2, W // 2, C).clone()
        m1 = getattr(g1, "_texir_mask", None)
        if m1 is not None:
            # a never-cleared stack: only the texels of the view's tap mask hold this step's values (level 1 leads the stack: bits [0, H/2 * W/2))
            n1 = (H // 2) * (W // 2)
            bits1 = ((m1.view(-1, 1).to(torch.int64) >> torch.arange(32, device=p.device)) & 1).bool().reshape(-1)[:n1].reshape(H // 2, W // 2, 1)
            l1 = torch.where(bits1, l1, torch.zeros((), device=p.device))
        g2 = getattr(p, "_texir_grad_l2", None)
        if g2 is not None:
            l1 += 0.25 * up(g2.view(H // 4, W // 4, C))          # (level 2 is written densely by the fold kernels: only level 1 is read through the mask)
        return g + 0.25 * up(l1)

    def set_clamp(self, param, lo=-math.inf, hi=math.inf):
        """fuse `param.data.clamp_(lo, hi)` into every step of this parameter"""
        self._clamps[id(param)] = (float(lo), float(hi))

    @torch.no_grad()
    def step(self, closure=None, _count_on_host=True):
        """_count_on_host=False: the call is being RECORDED into a hipGraph (the host-side step counts advance per replay instead,
        note_replayed_step)"""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        self.prepare()
        todo = []
        for group in self.param_groups:
            for p in group["params"]:
                g1 = getattr(p, "_texir_grad_l1", None)
                if p.grad is None and g1 is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.TexirError("FusedAdam needs contiguous float32 CUDA parameters")
                todo.append((group, p, g1))
        # one tick per device advances the step counts of the parameters that step now and derives their step sizes / bias corrections
        for device in {p.device for _, p, _ in todo}:
            d = self._records(device)
            mask = 0
            for _, p, _ in todo:
                if p.device == device:
                    mask |= 1 << self._rec[id(p)]
            _lib.check(L.texir_adam_tick(_lib.ptr(d["state"]), _lib.ptr(d["hyper"]), d["state"].shape[0], mask, _lib.stream_ptr()))
        from . import texture as _tx
        pending, keep = [], []            # texture jobs waiting for their batched launch (texir_adam_step_tex_dev_batch); tensors they point to

        def flush():
            if pending:
                _lib.batch_call("texir_adam_step_tex_dev_batch", pending)
                del pending[:], keep[:]

        for group, p, g1 in todo:
            b1, b2 = group["betas"]
            st = self._ensure_state(p)
            if _count_on_host:
                st["step"] += 1
            hyper = self._dev[p.device]["hyper"][self._rec[id(p)]]
            lo, hi = self._clamps.get(id(p), (-math.inf, math.inf))
            g = None if p.grad is None else p.grad.contiguous()      # None: level-0 gradient identically zero (texture.py backward)
            mask = None
            if g1 is None and getattr(p, "_texir_l0_sparse", False):
                raise _lib.TexirError("FusedAdam.step: a sparse level-0 gradient without its parked level-1 stack (state of another backward pass?)")
            if g1 is not None and getattr(p, "_texir_l0_sparse", False):
                # texture.py's sparse level-0 gradient: valid only at the texels of the view's bit mask, in the parameter's own buffer
                mask = getattr(p, "_texir_l0_mask", None)
                if mask is not None:
                    if g is None:
                        g = p._texir_g0
                    else:
                        # (another fetch of the same parameter also produced a dense gradient in this backward pass: fold the sparse part in)
                        H, W, C = p.shape
                        # (one launch; the stage-1 step of a view that samples level 0: un-mipmapped fetch = dense gradient, trilinear fetch = sparse part.
                        # Until round 6 five elementwise torch launches over the whole texture -- no boolean indexing: index_put with a mask synchronises
                        # and cannot be recorded into a hipGraph -- 130 us of the 650 us stage-1 step at 4096^2)
                        rc = L.texir_grad_add_masked(_lib.ptr(g), _lib.ptr(p._texir_g0), _lib.ptr(mask), H * W, C, _lib.stream_ptr())
                        if rc:
                            raise _lib.TexirError(L.texir_batch_last_error().decode())
                        mask = None
            if g1 is not None:
                H, W, C = p.shape
                # level 1 of the next forward's mip stack is written on the way (texture._mips_for then builds levels 2.. only)
                mips = getattr(p, "_texir_mips", None)
                mip1 = mips[1] if (mips is not None and mips[1].numel() >= (H // 2) * (W // 2) * C and mips[1].device == p.device) else None
                g2 = getattr(p, "_texir_grad_l2", None)
                g1_read = None if (g2 is not None and getattr(p, "_texir_l1_zero", False)) else g1      # (all zeros: not read, texture.py backward)
                # a stack that is never cleared: valid -- and read -- only where the view's tap mask says so (texture._bwd_finish parks the mask on the stack)
                g1_mask = getattr(g1, "_texir_mask", None) if g2 is not None else None
                if g1_mask is not None and g1_read is None:
                    g1_mask = None
                if _tx._BATCH:
                    A = _lib.addr
                    pending.append(_lib.AdamTexJob(A(p), A(g), A(mask), A(g1_read), A(g1_mask), A(g2), A(st["exp_avg"]), A(st["exp_avg_sq"]), A(mip1), H, W, C,
                                                   A(hyper), float(b1), float(b2), float(group["eps"]), lo, hi))
                    keep.extend([g, mask, g1_read, g1_mask, g2, hyper])
                    if len(pending) == _lib.MAX_BATCH:
                        flush()
                else:
                    if getattr(g1, "_texir_mask", None) is not None:
                        raise _lib.TexirError("FusedAdam.step: a masked gradient stack needs the batched step (TEXIR_TEX_BATCH was switched off between backward and step)")
                    _lib.check(L.texir_adam_step_tex_dev(_lib.ptr(p), _lib.ptr(g), _lib.ptr(mask), _lib.ptr(g1_read), _lib.ptr(g2),
                                                         _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), _lib.ptr(mip1), H, W, C, _lib.ptr(hyper),
                                                         float(b1), float(b2), float(group["eps"]), lo, hi, _lib.stream_ptr()))
                # one-shot: the next mip build of this parameter may start from level 1 (texture._mips_for consumes the flag)
                p._texir_mip1_fresh = (p.data_ptr(), p._version) if mip1 is not None else None
                p._texir_grad_l1 = p._texir_grad_l2 = None         # consumed (a hipGraph replay re-attaches its own, graph_step.step)
            else:
                flush()                                            # (launch order = parameter order)
                p._texir_mip1_fresh = None                         # the texture changes behind the mip stack's back
                _lib.check(L.texir_adam_step_dev(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), p.numel(), _lib.ptr(hyper),
                                                 float(b1), float(b2), float(group["eps"]), lo, hi, _lib.stream_ptr()))
        flush()
        return loss
