"""hipGraph capture of the launch-bound part of a material-estimation step (forward + loss + backward is ~40 short kernels:
mip builds, texture fetches, specular trace, three loss passes, scatter + folds).  The per-step GGX shifts still come from
the CPU generator exactly as the reference draws them (utils/sample_util.py:102) -- they are copied into a static device
buffer the captured kernels read.  Optimiser step and gradient all-reduce stay outside the graph."""
import torch


class GraphedMatStep:
    def __init__(self, model, loss_fn, optimizer, params):
        self.model, self.loss_fn, self.opt, self.params = model, loss_fn, optimizer, list(params)
        self.graphs, self.losses, self.pool = {}, {}, None
        self.side = torch.cuda.Stream()
        self.static_shift = None
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)

    def _fwd_bwd(self, inp, stage):
        mvp, cam, gt, gmask, seg, fm, room, key = inp
        preds = self.model(mvp, key, cam, stage)
        loss = self.loss_fn(gt, preds, gmask, fm, seg, stage=stage, room_seg_mask=room)[0]
        self.opt.zero_grad(set_to_none=False)
        loss.backward()
        return loss

    def capture(self, key, mvp, cam, gt, gmask, seg, fm, room, stage):
        """inputs must be device tensors that stay alive; one graph per (view key, stage)"""
        P = gt.shape[0] * gt.shape[1] * gt.shape[2]
        if self.static_shift is None:
            self.static_shift = torch.zeros((P, 2), device=gt.device)
        inp = (mvp, cam, gt, gmask, seg, fm, room, key)
        self.model._static_shift = None
        self._fwd_bwd(inp, stage)                       # eager warm-up: G-buffer cache, mask compaction, mip-stack buffers
        self.model._static_shift = self.static_shift
        try:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, stream=self.side):
                    loss = self._fwd_bwd(inp, stage)
            torch.cuda.current_stream().wait_stream(self.side)
        finally:
            self.model._static_shift = None
        self.pool = g.pool()
        self.graphs[(key, stage)] = g
        self.losses[(key, stage)] = loss

    def step(self, key, stage, all_reduce=None):
        """one optimiser step on a captured view; returns the (static) loss tensor of that graph"""
        P = self.static_shift.shape[0]
        self.static_shift.copy_(torch.rand(P, 1, 2).reshape(P, 2), non_blocking=True)
        self.graphs[(key, stage)].replay()
        if all_reduce is not None:
            for p in self.params:
                all_reduce(p.grad)
        self.opt.step()
        return self.losses[(key, stage)]
