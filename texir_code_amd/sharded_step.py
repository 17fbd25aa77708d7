"""Multi-GPU material step, parity mode (the pixels of the ONE view of a step are split across the ranks; the reference is
single-GPU, trainer/train_material.py:408-458 is the step that is reproduced).

What is sharded is the per-pixel work that costs: the GGX specular trace (forward) and its backward -- rank r handles the pixel slice
dist_util.pixel_range(P, r, world).  Everything on the TEXTURE side is replicated on every rank and therefore identical to the single-GPU step:
mip builds, the four full-view fetches, the loss on the full view, the fetch backward (gathers over the view's tap lists, folds, the sparse level-0
path) and the fused Adam.  Two small all_gathers per step carry per-pixel data only:

    forward   rgb of the rank's pixels                [P_r, 3]  ->  [P, 3]      (98 304 x 12 B at c = 128)
    backward  d rgb / d albedo, d rgb / d roughness    [P_r, 4]  ->  [P, 4]      (98 304 x 16 B)

instead of an all-reduce of the texture-gradient stacks (84 MB at 4k^2 textures, longer than the whole single-GPU step).  A pixel's specular term does
not depend on which other pixels share its wave, and every rank applies the identical texture-side kernels to identical inputs: the trajectory is the
single-GPU trajectory BIT FOR BIT, for any world size (tests/test_gpu_scale.py).  Stage 0 has no specular term: it needs no communication at all and
runs as the plain replicated GraphedMatStep.

The three phases between the collectives are hipGraphs (use_graph=True): P1 = zero_grad, mip builds, fetches, specular forward of the slice; P2 = loss
on the gathered view, its gradient, specular backward of the slice; P3 = the view's fetches again under autograd (mip stacks reused) with the total
gradients -> gathers, folds -> optimiser step.  use_graph=False runs the
same three phases eagerly."""
import numpy as np
import torch

from . import dist_util
from .scene import spec_backward_raw, spec_forward_raw, spec_shift_arg


class ShardedMatStep:
    def __init__(self, model, loss_fn, optimizer, params, use_graph=True):
        import torch.distributed as dist
        self.model, self.loss_fn, self.opt, self.params = model, loss_fn, optimizer, list(params)
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
        self.use_graph = bool(use_graph)
        self.views, self.pool = {}, None
        self.side = torch.cuda.Stream()
        for p in self.params:
            p._texir_replicated_grads = True        # texture.py: nothing is reduced across ranks, the sparse level-0 gradient stays valid
            p._texir_mip1_graph = True
        self._seed = None

    # ---- the three phases -------------------------------------------------------------------------------------------------------------------
    # Every phase is self-contained for autograd: no graph made in one phase is walked in another (each phase is its own hipGraph recording, and an
    # autograd graph carries the stream and the allocations of the recording that made it).  P1 and P2 call the specular kernels directly; P3
    # fetches once more under autograd (the mip stacks P1 built are reused: 2-3 short fetch launches per step) and walks that graph itself.
    def _p1(self, st):
        m, stage = self.model, st["stage"]
        self.opt.zero_grad(set_to_none=True)
        gb = m._view_consts(m._gbuffer(st["mvp"], st["key"]))
        with torch.no_grad():
            albedo, rw, rough, irr = m._fetch_materials(gb, womipmap=(stage == 1))
        P, (p0, p1) = st["P"], st["range"]
        flat = lambda t, k: t.reshape(P, k)
        st["vals"] = {"albedo": albedo, "roughness": rough, "roughness_womipmap": rw}
        r_full = rw if stage == 1 else rough
        sl = {"normal": flat(gb["normal"], 3)[p0:p1].contiguous(), "albedo": flat(albedo, 3)[p0:p1].contiguous(),
              "rough": flat(r_full, 1)[p0:p1].reshape(-1).contiguous(), "points": flat(gb["_points"], 3)[p0:p1].contiguous(),
              "irr": flat(irr, 3)[p0:p1].contiguous(), "cam": st["cam"].to(torch.float32).reshape(3).contiguous(),
              "shift": spec_shift_arg(st["shift"][p0:p1], p1 - p0, st["gt"].device)}
        rgb_slice, sl["Ls"], sl["dw"] = spec_forward_raw(m.scene, sl["normal"], sl["albedo"], sl["rough"], sl["points"], sl["irr"], sl["cam"], sl["shift"], int(m.sample_l[1]),
                                                         want_dw=bool(m.materials_r.requires_grad))
        st["slice"] = sl
        st["send1"][: p1 - p0].copy_(rgb_slice)

    def _p2(self, st):
        m, stage = self.model, st["stage"]
        P, c, (p0, p1) = st["P"], st["c"], st["range"]
        gb = m._gbuffer(st["mvp"], st["key"])
        rgb = torch.cat([st["recv1"][r][: b - a] for r, (a, b) in enumerate(st["ranges"])], 0).requires_grad_(True)
        train = {"albedo": m.materials_a.requires_grad, "roughness": m.materials_r.requires_grad, "roughness_womipmap": m.materials_r.requires_grad}
        leaves = {k: (None if t is None else t.detach().requires_grad_(bool(train[k]))) for k, t in st["vals"].items()}
        sh = lambda t, k: None if t is None else t.reshape(6, c, c, k)
        preds = {"rgb": sh(rgb, 3), "albedo": sh(leaves["albedo"], 3), "roughness": sh(leaves["roughness"], 1),
                 "roughness_womipmap": sh(leaves["roughness_womipmap"], 1), "empty_mask": gb["mask"]}
        out = self.loss_fn(st["gt"], preds, st["gmask"], st["fm"], st["seg"], stage=stage, room_seg_mask=st["room"])
        st["out"] = (out[0].detach(),) + tuple(o.detach() if torch.is_tensor(o) else o for o in out[1:])
        if self._seed is None or self._seed.device != out[0].device:
            self._seed = torch.ones((), device=out[0].device)
        names = [k for k, t in leaves.items() if t is not None and t.requires_grad]
        got = torch.autograd.grad(out[0], [rgb] + [leaves[k] for k in names], self._seed, allow_unused=True)
        st["d_leaf"] = dict(zip(names, got[1:]))
        # backward of the slice's specular term (stage 1 renders on the detached albedo: no albedo gradient)
        sl = st["slice"]
        d_a, d_r = spec_backward_raw(sl["normal"], sl["rough"], sl["points"], sl["irr"], sl["cam"], sl["shift"], sl["Ls"], got[0][p0:p1], int(m.sample_l[1]),
                                     need_albedo=(stage == 2 and m.materials_a.requires_grad), need_rough=m.materials_r.requires_grad, dw=sl["dw"])
        s2 = st["send2"]
        s2.zero_()
        if d_a is not None:
            s2[: p1 - p0, 0:3].copy_(d_a)
        if d_r is not None:
            s2[: p1 - p0, 3].copy_(d_r)

    def _p3(self, st, step):
        """step: "none" (warm-up: fetch backward only), "eager", or "record" (the call is being recorded: host step counts advance per replay)"""
        m, stage, P = self.model, st["stage"], st["P"]
        full = torch.cat([st["recv2"][r][: b - a] for r, (a, b) in enumerate(st["ranges"])], 0)       # [P, 4]: d albedo (3), d roughness (1) of the specular term
        d_spec_a, d_spec_r = full[:, 0:3], full[:, 3:4]
        # the view's fetches under autograd (same kernels, same values as in P1; the mip stacks of this step are reused)
        gb = m._gbuffer(st["mvp"], st["key"])
        for p in self.params:
            p._texir_reuse_mips = True
        try:
            albedo, rw, rough, _ = m._fetch_materials(gb, womipmap=(stage == 1))
        finally:
            for p in self.params:
                p._texir_reuse_mips = False
        f, dl = {"albedo": albedo, "roughness": rough, "roughness_womipmap": rw}, st["d_leaf"]
        outs, grads = [], []

        def add(t, d_direct, extra):
            """total gradient of a fetch output = what the loss sent straight into it (+) what came through the specular term"""
            if t is None or not t.requires_grad:
                return
            g = None if d_direct is None else d_direct.reshape(t.shape)
            if extra is not None:
                e = extra.reshape(t.shape)
                g = e if g is None else g + e
            if g is not None:
                outs.append(t)
                grads.append(g.contiguous())

        add(f["albedo"], dl.get("albedo"), d_spec_a if stage == 2 else None)
        add(f["roughness"], dl.get("roughness"), d_spec_r if stage == 2 else None)
        n_tri = len(outs)                        # the trilinear fetches: outputs of ONE node (texture.texture_batch) -- walked together
        add(f["roughness_womipmap"], dl.get("roughness_womipmap"), d_spec_r if stage == 1 else None)
        # the fetch backward of the full view: gathers over the view's tap lists, folds, parked stacks -- the gradient lands on the parameters exactly as in
        # the single-process step.  The trilinear fetches of the textures first, then the un-mipmapped one (the order the single-process
        # backward runs them; texture.py asks `owner.grad is None` to decide between the sparse and the dense level-0 form)
        train = [p for p in self.params if p.requires_grad]
        total = {id(p): None for p in train}
        walks = ([(outs[:n_tri], grads[:n_tri])] if n_tri else []) + [([t], [g]) for t, g in zip(outs[n_tri:], grads[n_tri:])]
        for ts, gs in walks:
            part = torch.autograd.grad(ts, train, gs, allow_unused=True)
            for p, d in zip(train, part):
                if d is not None:
                    total[id(p)] = d if total[id(p)] is None else total[id(p)] + d
            for p in train:
                p.grad = total[id(p)]
        if step == "record":
            everyone = [q for grp in self.opt.param_groups for q in grp["params"]]
            st["stepping"] = [p for p in everyone if p.grad is not None or getattr(p, "_texir_grad_l1", None) is not None]
            self.opt.step(_count_on_host=False)
        elif step == "eager":
            self.opt.step()

    # ---- collectives --------------------------------------------------------------------------------------------------------------------------
    def _gather(self, recv, send):
        import torch.distributed as dist
        if self.world > 1:
            dist_util.account(send, self.world)
            dist.all_gather(recv, send)
        else:
            recv[0].copy_(send)

    # ---- set-up -----------------------------------------------------------------------------------------------------------------------------
    def capture(self, key, mvp, cam, gt, gmask, seg, fm, room, stage):
        """inputs must be device tensors that stay alive; one state (and, with use_graph, three graphs) per (view key, stage)"""
        from .plot_writer import capture_gate           # no worker thread may touch the HIP runtime while a stream is being captured
        if stage == 0:
            raise ValueError("stage 0 has no specular term: run it as the replicated GraphedMatStep (no communication needed)")
        dev = gt.device
        c = gt.shape[1]
        P = gt.shape[0] * c * c
        ranges = [dist_util.pixel_range(P, r, self.world) for r in range(self.world)]
        mx = max(b - a for a, b in ranges)
        z = lambda k: torch.zeros((mx, k), device=dev)
        st = {"key": key, "stage": stage, "mvp": mvp, "cam": cam, "gt": gt, "gmask": gmask, "seg": seg, "fm": fm, "room": room if stage == 2 else None,
              "P": P, "c": c, "ranges": ranges, "range": ranges[self.rank],
              "shift": torch.zeros((P, 2), dtype=torch.float32).pin_memory(), "shift_ev": torch.cuda.Event(), "shift_pending": False,
              "send1": z(3), "recv1": [z(3) for _ in range(self.world)], "send2": z(4), "recv2": [z(4) for _ in range(self.world)]}
        self.views[(key, stage)] = st
        # eager warm-up (G-buffer cache, tap lists, mip / gradient buffers): forward + both exchanges + backward WITHOUT the optimiser step; it must
        # not consume the CPU-generator stream of the training run
        rng = torch.get_rng_state()
        st["shift"].copy_(torch.rand(P, 2))
        torch.set_rng_state(rng)
        self._p1(st)
        self._gather(st["recv1"], st["send1"])
        self._p2(st)
        self._gather(st["recv2"], st["send2"])
        self._p3(st, "none")
        self.opt.zero_grad(set_to_none=True)
        # nothing of the warm-up's autograd graph may outlive it: the parameters' AccumulateGrad nodes would be kept alive with the warm-up's stream and
        # the recorded backward would synchronise with a stream that is not being captured
        if self.use_graph:
            for k in ("vals", "slice", "d_leaf", "out"):
                st.pop(k, None)
        if hasattr(self.opt, "prepare"):
            self.opt.prepare()                         # moments + device-resident step records exist before a capture
        if not self.use_graph:
            return
        import gc
        from .scene import defer_destroy
        gc.collect()
        was = gc.isenabled()
        gc.disable()
        try:
            with defer_destroy(), capture_gate():
                graphs = []
                for phase in (lambda: self._p1(st), lambda: self._p2(st), lambda: self._p3(st, "record")):
                    self.side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self.side):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=self.pool, stream=self.side):
                            phase()
                    torch.cuda.current_stream().wait_stream(self.side)
                    self.pool = g.pool()
                    graphs.append(g)
                    if len(graphs) < 3:
                        # what the next phase reads from the collective must exist: run the exchange once on the recorded (not executed) buffers
                        self._gather(st["recv1"] if len(graphs) == 1 else st["recv2"], st["send1"] if len(graphs) == 1 else st["send2"])
                st["graphs"] = graphs
        finally:
            if was:
                gc.enable()
        self.opt.zero_grad(set_to_none=True)

    def _run_eager(self, st):
        self._p1(st)
        self._gather(st["recv1"], st["send1"])
        self._p2(st)
        self._gather(st["recv2"], st["send2"])
        self._p3(st, "eager")
        # the pinned shift buffer is read zero-copy by the specular forward (and by the backward when it has no saved workspace): the next step() may
        # not overwrite it before those kernels have run -- a host that does not .item() every step (train.log_lag > 0) would otherwise run ahead
        st["shift_ev"].record()
        st["shift_pending"] = True

    def step(self, key, stage, shift=None):
        """one optimiser step on a captured view; returns the loss tensor.  The step's GGX shifts are the reference's full-view draw from the CPU generator
        (utils/sample_util.py:102; same seed on every rank => the same stream as the single-GPU run): every rank draws all P and reads its slice."""
        st = self.views[(key, stage)]
        if shift is None:
            shift = torch.rand(st["P"], 1, 2).reshape(st["P"], 2)
        if st["shift_pending"]:
            st["shift_ev"].synchronize()
            st["shift_pending"] = False
        st["shift"].copy_(shift)
        if "graphs" not in st:
            self._run_eager(st)
            return st["out"][0]
        from .texture import refresh_mips
        for p in self.params:
            mips = getattr(p, "_texir_mips", None)
            if mips is None or getattr(p, "_texir_mip1_fresh", None) == (p.data_ptr(), p._version):
                continue
            if not p.requires_grad and mips[0][:2] == (p.data_ptr(), p._version):
                continue
            refresh_mips(p)
        if hasattr(self.opt, "prepare"):
            self.opt.prepare()
        g1, g2, g3 = st["graphs"]
        g1.replay()
        self._gather(st["recv1"], st["send1"])
        g2.replay()
        st["shift_ev"].record()
        st["shift_pending"] = True
        self._gather(st["recv2"], st["send2"])
        g3.replay()
        self.opt.note_replayed_step(st["stepping"])
        return st["out"][0]
