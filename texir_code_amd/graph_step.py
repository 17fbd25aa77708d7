"""hipGraph capture of a whole material-estimation step (trainer/train_material.py:408-458): forward + loss + backward AND the optimiser
step -- 13 launches in stage 2: mip pyramids + their tails + the fetches of both textures (one batched launch each, texture.texture_batch),
specular trace, four loss passes, specular backward, gather + folds of both textures (batched; the gradient stacks are read through the view's
tap mask and never cleared; the roughness fetch's two gradients -- specular term and loss -- are added inside the gather), one tick of the
device-resident step counts and ONE fused Adam launch over both textures
(optim.FusedAdam keeps step count and learning rate in device memory, so no kernel argument changes from replay to replay).  The per-step GGX shifts still come from the CPU generator exactly as the reference draws them
(utils/sample_util.py:102) -- they are copied into a static device buffer the captured kernels read.
Multi-GPU runs (a gradient all-reduce between backward and step) keep the optimiser step outside the graph (step_in_graph=False)."""
import queue
import threading

import torch


class _DrawWorker:
    """ONE helper thread that runs the host-side draw of the NEXT step's GGX shifts (the CPU generator's 196 608 floats: 0.4-0.5 ms, as long as a whole material
    step on the GPU) while the main thread reports the current step.  Requests are executed one at a time in submission order and every request is awaited
    before anything else may touch the global generator, so the generator's stream is consumed in exactly the order of the reference's loop
    (utils/sample_util.py:102); torch.rand releases the GIL while it runs.  The thread makes no HIP call (a foreign thread inside the HIP runtime would
    invalidate a stream capture)."""

    def __init__(self):
        self.q = queue.SimpleQueue()
        self.thread = threading.Thread(target=self._run, name="texir-shift-draw", daemon=True)
        self.thread.start()

    def _run(self):
        torch.set_num_threads(1)
        while True:
            job = self.q.get()
            if job is None:
                return
            fn, done, err = job
            try:
                fn()
            except BaseException as e:          # surfaces in wait()
                err.append(e)
            done.set()

    def submit(self, fn):
        done, err = threading.Event(), []
        self.q.put((fn, done, err))
        return done, err

    def close(self):
        self.q.put(None)


class GraphedMatStep:
    def __init__(self, model, loss_fn, optimizer, params, step_in_graph=None):
        self.model, self.loss_fn, self.opt, self.params = model, loss_fn, optimizer, list(params)
        if step_in_graph is None:
            import torch.distributed as dist
            step_in_graph = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        import os
        if os.environ.get("TEXIR_GRAPH_STEP") == "0":        # A/B switch: optimiser step launched eagerly after the replay (round 2)
            step_in_graph = False
        self.step_in_graph = bool(step_in_graph) and hasattr(optimizer, "note_replayed_step")
        self.stepped = {}
        # The GGX shifts of a step are drawn on the host (CPU generator, as the reference) into a PINNED buffer of the view, and the recorded
        # kernels read that host memory directly (pinned allocations are mapped into the device's address space): no staging copy sits between
        # the draw and the replay (0.8 MB per step: 98 304 x 2 floats, read once by the forward and once by the backward over the host
        # link, hidden inside the 220 us specular kernel).  One buffer per captured view, so the next step's shifts can be written while the
        # current step still runs.  TEXIR_SHIFT_ZEROCOPY=0 keeps round 2's device buffer + async copy through a pinned ring.
        self.zero_copy = os.environ.get("TEXIR_SHIFT_ZEROCOPY", "1") != "0"
        self.shift_bufs = {}
        self.graphs, self.losses, self.outs, self.pool = {}, {}, {}, None
        self.inputs = {}                   # (key, stage) -> the captured step's input tensors: the recorded kernels hold their ADDRESSES, so they must outlive the graph
        self.side = torch.cuda.Stream()
        self.grads_to_none = os.environ.get("TEXIR_GRAPH_GRADS_TO_NONE", "1") == "1"
        if not self.grads_to_none:
            for p in self.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        self.static_shift = None
        self.grads = {}
        for p in self.params:
            p._texir_mip1_graph = True         # recorded mip builds start from level 1; step() keeps that level valid (texture._mips_for)

    def _fwd_bwd(self, inp, stage, record_step=False):
        mvp, cam, gt, gmask, seg, fm, room, key = inp
        # grads are re-created by the backward itself (autograd assigns instead of accumulating): inside the captured graph their
        # addresses are static, and the zero-fill + accumulate passes over the full textures disappear.  Before the forward: a gradient
        # still parked on a parameter (a warm-up pass without optimiser step) would make the forward keep the gradient arena as it is.
        self.opt.zero_grad(set_to_none=self.grads_to_none)
        preds = self.model(mvp, key, cam, stage)
        out = self.loss_fn(gt, preds, gmask, fm, seg, stage=stage, room_seg_mask=room)
        loss = out[0]
        # only detached views are kept: an autograd graph rooted in capture-time tensors must not outlive the capture
        self._last_out = (out[0].detach(),) + tuple(o.detach() if torch.is_tensor(o) else o for o in out[1:])
        if getattr(self, "_seed", None) is None or self._seed.device != loss.device:
            self._seed = torch.ones((), device=loss.device)          # (a persistent unit seed: loss.backward() would fill a fresh one per step)
        torch.autograd.backward(loss, self._seed)
        if record_step:
            # which parameters this graph steps (a stage-1 step leaves the albedo texture without gradient): their host-side step counts
            # advance per replay (FusedAdam.note_replayed_step)
            # (every parameter of the optimiser that holds a gradient is stepped by the recorded opt.step, not only self.params)
            everyone = [q for grp in self.opt.param_groups for q in grp["params"]]
            self._stepping = [p for p in everyone if p.grad is not None or getattr(p, "_texir_grad_l1", None) is not None]
            self.opt.step(_count_on_host=False)
        return loss

    def capture(self, key, mvp, cam, gt, gmask, seg, fm, room, stage):
        """inputs must be device tensors (they are kept alive with the graph, self.inputs: the recorded kernels read them by address); one graph per
        (view key, stage)"""
        from .plot_writer import capture_gate           # no worker thread may touch the HIP runtime while a stream is being captured
        P = gt.shape[0] * gt.shape[1] * gt.shape[2]
        if self.static_shift is None:
            self.static_shift = torch.zeros((P, 2), device=gt.device)
        if self.zero_copy and key not in self.shift_bufs:
            self.shift_bufs[key] = [torch.zeros((P, 2), dtype=torch.float32).pin_memory(), torch.cuda.Event(), False]
        inp = (mvp, cam, gt, gmask, seg, fm, room, key)
        self.model._static_shift = None
        rng_state = torch.get_rng_state()               # warm-up must not consume the CPU-generator stream of the training run
        self._fwd_bwd(inp, stage)                       # eager warm-up: G-buffer cache, mask compaction, mip-stack buffers (no optimiser step)
        torch.set_rng_state(rng_state)
        if hasattr(self.opt, "prepare"):
            self.opt.prepare()                          # moments + device-resident step records exist before the capture
        self.model._static_shift = self.shift_bufs[key][0] if self.zero_copy else self.static_shift
        import gc
        from .scene import defer_destroy
        # finalisers (hipFree of dead scenes / tensors) must not run inside a capture: the collector is switched off for its duration; a full collection
        # beforehand is only worth its 20-50 ms once per step object (a stage captures one graph per view: 16-500 of them back to back)
        # (round 6: a FULL collection cost 127 ms per step object on the c4 run -- the interpreter holds millions of long-lived objects by then; the young
        # generation is where a dead scene or tensor cycle of the preceding stage sits, and collecting it costs microseconds)
        if not getattr(self, "_collected_once", False):
            gc.collect(0)
            self._collected_once = True
        gc_was = gc.isenabled()
        gc.disable()
        try:
            with defer_destroy(), capture_gate():
                self.side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.side):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=self.pool, stream=self.side):
                        loss = self._fwd_bwd(inp, stage, record_step=self.step_in_graph)
                torch.cuda.current_stream().wait_stream(self.side)
        except BaseException:
            # a capture that aborts has RECORDED the arena fill without running it: the arena still holds the warm-up pass's gradients
            for p in self.params:
                if getattr(p, "_texir_arena", None) is not None:
                    p._texir_arena["clean"] = set()
            self.opt.zero_grad(set_to_none=True)
            raise
        finally:
            self.model._static_shift = None
            if gc_was:
                gc.enable()
        self.pool = g.pool()
        # The gradient stacks live in per-parameter buffers shared by all graphs (texture.py backward); whatever else autograd assigned
        # during this capture (a level-0 gradient, when the view samples level 0) is remembered per graph and re-attached to the
        # parameters before every optimiser step.
        self.grads[(key, stage)] = [(p.grad, getattr(p, "_texir_grad_l1", None), getattr(p, "_texir_l0_touched", True),
                                     getattr(p, "_texir_l0_mask", None), getattr(p, "_texir_l0_sparse", False), getattr(p, "_texir_grad_l2", None),
                                     getattr(p, "_texir_l1_zero", False))
                                    for p in self.params]
        self.graphs[(key, stage)] = g
        # the graph reads its inputs (camera position, ground truth, masks) by address on every replay: a caller that passed a temporary -- `cam.cuda()` --
        # would otherwise leave the recorded kernels reading whatever the allocator puts there next
        self.inputs[(key, stage)] = inp
        self.stepped[(key, stage)] = list(getattr(self, "_stepping", [])) if self.step_in_graph else None
        self.losses[(key, stage)] = loss.detach()      # keep no autograd graph of the captured region alive
        self.outs[(key, stage)] = self._last_out
        self.opt.zero_grad(set_to_none=self.grads_to_none)   # (the recorded backward parked its gradients on the parameters: they belong to the graph)

    def draw_shift(self):
        """the step's GGX shifts from the CPU generator, exactly the reference's draw (sample_util.py:102).  Callers may draw the
        NEXT step's shifts right after launching a step so that the host RNG overlaps the GPU work (same stream order)."""
        P = self.static_shift.shape[0]
        return torch.rand(P, 1, 2).reshape(P, 2)

    def request_shift(self, key):
        """draw the NEXT step's shifts for view `key` on the helper thread, straight into the pinned buffer that view's recorded kernels read (zero-copy mode);
        wait_shift() must be called before the step is launched -- and before anything else draws from the global CPU generator.  Returns False (nothing
        requested) where the direct form does not apply."""
        if not self.zero_copy or key not in self.shift_bufs:
            return False
        self.wait_shift()
        buf, ev, pending = self.shift_bufs[key]
        if pending:
            ev.synchronize()                 # (main thread: the previous replay of this view -- an epoch ago -- has finished reading the buffer)
            self.shift_bufs[key][2] = False
        if getattr(self, "_worker", None) is None:
            self._worker = _DrawWorker()
        P = buf.shape[0]
        self._pending_draw = self._worker.submit(lambda: torch.rand(P, 1, 2, out=buf.view(P, 1, 2)))
        return True

    def wait_shift(self):
        pd = getattr(self, "_pending_draw", None)
        if pd is not None:
            self._pending_draw = None
            pd[0].wait()
            if pd[1]:
                raise pd[1][0]

    def close(self):
        self.wait_shift()
        if getattr(self, "_worker", None) is not None:
            self._worker.close()
            self._worker = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stage_shift(self, key, shift):
        """write the shifts of the NEXT step on view `key` where its recorded kernels will read them.  Callers that know the next view may call
        this right after launching a step (then pass staged=True to step()): the host-side copy then overlaps the GPU work."""
        if not self.zero_copy:
            return self._stage_shift(shift)
        buf, ev, pending = self.shift_bufs[key]
        if pending:
            ev.synchronize()                 # the previous replay of this view has finished reading the buffer
            self.shift_bufs[key][2] = False
        buf.copy_(shift)

    def _stage_shift(self, shift):
        """host -> device copy of the step's shifts through a ring of pinned buffers.  The copy is asynchronous, so a staging buffer
        may only be rewritten once ITS previous copy has run: each slot carries an event recorded right after its copy, and a slot is
        reused only after that event (callers may queue several steps without a host sync in between)."""
        if getattr(self, "_ring", None) is None:
            self._ring = [(torch.empty(tuple(self.static_shift.shape), dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(3)]
            self._ring_used = [False] * 3
            self._ring_next = 0
        i = self._ring_next
        buf, ev = self._ring[i]
        if self._ring_used[i]:
            ev.synchronize()
        buf.copy_(shift)
        self.static_shift.copy_(buf, non_blocking=True)
        ev.record()
        self._ring_used[i] = True
        self._ring_next = (i + 1) % len(self._ring)

    def step(self, key, stage, reduce_grads=None, shift=None, staged=False):
        """one optimiser step on a captured view; returns the (static) loss tensor of that graph.  staged=True: the caller has already put
        this step's shifts in place (stage_shift)"""
        self.wait_shift()                               # (a draw requested for this step has landed; nothing else may be in flight on the generator)
        if stage != 0 and not staged:                   # stage 0 is Lambertian only: the reference draws no shifts there
            if shift is None:
                shift = self.draw_shift()
            self.stage_shift(key, shift)
        # the captured forward may build only mip levels 2.. (level 1 comes from the previous optimiser step, optim.FusedAdam): if the
        # texture was changed any other way since, rebuild the stack eagerly before replaying
        from .texture import refresh_mips
        for p in self.params:
            mips = getattr(p, "_texir_mips", None)
            if mips is None or getattr(p, "_texir_mip1_fresh", None) == (p.data_ptr(), p._version):
                continue
            # A texture this stage does not train (roughness in stage 0, albedo in stage 1) is never stepped, so the one-shot flag never
            # matches -- but nothing changes it either: its stack, built in full by the eager warm-up, stays valid as long as the cache key
            # (data pointer, version, shape) does.  The same rule texture._mips_for applies to any frozen texture; no build per replay.
            if not p.requires_grad and mips[0][:2] == (p.data_ptr(), p._version):
                continue
            refresh_mips(p)
        if hasattr(self.opt, "prepare"):
            self.opt.prepare()                         # a learning-rate scheduler's change reaches the device record here
        self.graphs[(key, stage)].replay()
        if self.zero_copy and key in self.shift_bufs:
            self.shift_bufs[key][1].record()
            self.shift_bufs[key][2] = True
        if self.stepped[(key, stage)] is not None:
            self.opt.note_replayed_step(self.stepped[(key, stage)])      # the replay contained the optimiser step
            return self.losses[(key, stage)]
        for p, (g, g1, l0, mask, sparse, g2, l1z) in zip(self.params, self.grads[(key, stage)]):
            p.grad = g
            p._texir_grad_l1 = g1
            p._texir_grad_l2 = g2
            p._texir_l1_zero = l1z
            p._texir_l0_touched = l0
            p._texir_l0_mask = mask
            p._texir_l0_sparse = sparse
        if reduce_grads is not None:
            reduce_grads(self.params)              # multi-GPU: dist_util.reduce_texture_grads
        self.opt.step()
        return self.losses[(key, stage)]
