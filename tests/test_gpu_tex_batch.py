"""GPU: the batched forms of the texture-side launches (texir_*_batch, include/texir_hip.h "batched forms") and the never-cleared, mask-read gradient
stacks against the one-launch-per-texture forms they replace: same fetch values, same parameters / moments / mip level 1 after several optimiser steps,
BIT FOR BIT -- for textures of equal and of different sizes and channel counts, with and without level-0 taps, eagerly and through the C-ABI alone."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _coords(P, seed, lo=-3.0, hi=-0.7):
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand(P, 2, generator=g)
    da = (torch.randn(P, 4, generator=g) * torch.logspace(lo, hi, P).unsqueeze(-1)).float()
    return uv.cuda(), da.cuda()


def _run(sizes, batch, mask, steps=4, lo=-3.0, hi=-0.7, P=6000):
    """`steps` optimiser steps over two views (alternating) on textures of the given (H, W, C); returns everything a step leaves behind"""
    from texir_code_amd import texture as T
    from texir_code_amd.optim import FusedAdam
    was = (T._BATCH, T._GRAD_MASK)
    T._BATCH, T._GRAD_MASK = batch, mask
    try:
        torch.manual_seed(11)
        ps = [torch.nn.Parameter(torch.rand(H, W, Cc).cuda()) for H, W, Cc in sizes]
        opt = FusedAdam(ps, lr=2e-2, fuse_mip_fold=True)
        opt.set_clamp(ps[-1], 0.01, 0.8)
        views = [(_coords(P, 1, lo, hi), {}), (_coords(P // 2, 2, lo, hi), {})]
        ws = [[torch.randn(v[0][0].shape[0], Cc, generator=torch.Generator().manual_seed(7 + i)).cuda() for i, (_, _, Cc) in enumerate(sizes)] for v in views]
        outs, dense = [], []
        for it in range(steps):
            (uv, da), cache = views[it % 2]
            opt.zero_grad()
            fetched = T.texture_batch(ps, uv, da, "linear-mipmap-linear", 13, cache=cache)
            outs.append([f.detach().clone() for f in fetched])
            sum((f * w).sum() for f, w in zip(fetched, ws[it % 2])).backward()
            dense.append([opt.dense_grad(p).clone() for p in ps])
            opt.step()
        torch.cuda.synchronize()
        mips = [p._texir_mips[1][: (p.shape[0] // 2) * (p.shape[1] // 2) * p.shape[2]].clone() for p in ps]
        return {"outs": outs, "dense": dense, "p": [p.detach().clone() for p in ps], "m": [opt.state[p]["exp_avg"].clone() for p in ps],
                "v": [opt.state[p]["exp_avg_sq"].clone() for p in ps], "mip1": mips}
    finally:
        T._BATCH, T._GRAD_MASK = was


def _same(a, b, what):
    for k in ("p", "m", "v", "mip1"):
        for i, (x, y) in enumerate(zip(a[k], b[k])):
            assert torch.equal(x, y), (what, k, i, float((x - y).abs().max()))
    for it, (xs, ys) in enumerate(zip(a["outs"], b["outs"])):
        for i, (x, y) in enumerate(zip(xs, ys)):
            assert torch.equal(x, y), (what, "fetch", it, i)
    for it, (xs, ys) in enumerate(zip(a["dense"], b["dense"])):
        for i, (x, y) in enumerate(zip(xs, ys)):
            assert torch.equal(x, y), (what, "dense_grad", it, i)


@pytest.mark.parametrize("sizes", [[(256, 512, 3), (256, 512, 1)], [(128, 128, 3), (256, 256, 1)], [(64, 64, 4), (64, 64, 2), (128, 64, 1)]])
@pytest.mark.parametrize("fine", [False, True])
def test_batched_and_masked_steps_are_the_single_launch_steps_bit_for_bit(tx, sizes, fine):
    # fine: footprints small enough that many taps land on mip levels 0 and 1 (the sparse level-0 path and the level-1 mask carry values)
    lo, hi = (-4.5, -2.0) if fine else (-3.0, -0.7)
    base = _run(sizes, batch=False, mask=False, lo=lo, hi=hi)
    _same(base, _run(sizes, batch=True, mask=False, lo=lo, hi=hi), "batched")
    _same(base, _run(sizes, batch=True, mask=True, lo=lo, hi=hi), "batched + masked")


def test_masked_stack_is_never_cleared_and_still_right(tx):
    """the gradient arena is poisoned before every backward: with the mask nothing may depend on what it held (without the mask the fill would wipe the poison
    anyway, so the poisoning itself is harmless there)"""
    from texir_code_amd import texture as T
    from texir_code_amd.optim import FusedAdam
    sizes = [(256, 256, 3), (256, 256, 1)]

    def run(poison):
        torch.manual_seed(3)
        ps = [torch.nn.Parameter(torch.rand(H, W, Cc).cuda()) for H, W, Cc in sizes]
        opt = FusedAdam(ps, lr=1e-2, fuse_mip_fold=True)
        (uv, da), cache = _coords(5000, 4), {}
        for it in range(3):
            opt.zero_grad()
            a, r = T.texture_batch(ps, uv, da, "linear-mipmap-linear", 13, cache=cache)
            if poison:
                ps[0]._texir_arena["buf"].fill_(float("nan"))
            ((a * a).sum() + (r * 3).sum()).backward()
            assert getattr(ps[0]._texir_grad_l1, "_texir_mask", None) is not None
            opt.step()
        return [p.detach().clone() for p in ps]

    clean, dirty = run(False), run(True)
    for x, y in zip(clean, dirty):
        assert torch.isfinite(y).all() and torch.equal(x, y)


def test_batch_entry_points_validate_their_jobs(tx):
    from texir_code_amd import _lib
    L = _lib.lib()
    t = torch.rand(64, 64, 3, device="cuda")
    uv, da = _coords(100, 1)
    out = torch.empty(100, 3, device="cuda")
    A = _lib.addr
    good = _lib.TexFetchJob(A(t), None, 64, 64, 3, 1, -1, 0, A(uv), A(da), 100, A(out))
    _lib.batch_call("texir_tex_fetch_forward_batch", [good])
    arr = (_lib.TexFetchJob * 5)(*([good] * 5))
    assert L.texir_tex_fetch_forward_batch(arr, 5, None) != 0 and b"jobs" in L.texir_batch_last_error()
    bad = _lib.TexFetchJob(A(t), None, 64, 64, 7, 1, -1, 0, A(uv), A(da), 100, A(out))
    with pytest.raises(_lib.TexirError, match="job 0: bad texture"):
        _lib.batch_call("texir_tex_fetch_forward_batch", [bad])
    # a mask without the two deferred folds is refused
    g = _lib.TexGatherJob(A(t), A(t), 64, 64, 3, 5, None, None, None, 0, None, None, A(out), 1, 1, A(t))
    with pytest.raises(_lib.TexirError, match="rest_mask needs defer_last_fold = 2"):
        _lib.batch_call("texir_tex_gather_backward_batch", [g])


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_recorded_step_keeps_its_trajectory_under_every_switch(golden, stage):
    """the whole material step as a hipGraph (graph_step.GraphedMatStep on the 20 k room, different texture sizes for albedo and roughness): five replays with
    one launch per texture and cleared stacks, with batched launches, and with batched launches + mask-read stacks leave the same textures, bit for bit --
    in stage 1 the roughness texture also takes a dense gradient from its un-mipmapped fetch, in stages 0 / 1 one member of the batch is frozen"""
    from test_gpu_optim_regressions import _small_graphed_world
    from texir_code_amd import texture as T
    was = (T._BATCH, T._GRAD_MASK)
    got = []
    try:
        for batch, mask in ((False, False), (True, False), (True, True)):
            T._BATCH, T._GRAD_MASK = batch, mask
            m, opt, gs, c = _small_graphed_world(golden, stage)
            gen = torch.Generator().manual_seed(1)
            losses = []
            for _ in range(5):
                # (the clone is a launch of its own between two replays -- and a small allocation: it once landed on the freed camera vector a recorded
                # kernel was still reading by address; GraphedMatStep keeps its captured inputs alive since)
                losses.append(gs.step("v", stage, shift=torch.rand(6 * c * c, 2, generator=gen)).clone())
            torch.cuda.synchronize()
            got.append((m.materials_a.detach().clone(), m.materials_r.detach().clone(), torch.stack(losses)))
    finally:
        T._BATCH, T._GRAD_MASK = was
    for k in (1, 2):
        for i, name in enumerate(("albedo", "roughness", "losses")):
            assert torch.equal(got[0][i], got[k][i]), (stage, k, name, float((got[0][i] - got[k][i]).abs().max()))


@pytest.mark.parametrize("mask", [False, True])
def test_two_consumers_of_one_fetch_add_their_gradients_in_the_gather(tx, mask):
    """fanout = 2: the roughness fetch is handed out as two tensors (specular term, loss); their gradients are added inside the gather launch -- the same
    parameters, bit for bit, as with ONE output whose two gradients autograd adds in a launch of its own"""
    from texir_code_amd import texture as T
    from texir_code_amd.optim import FusedAdam
    was = T._GRAD_MASK
    T._GRAD_MASK = mask
    try:
        def run(fan):
            torch.manual_seed(21)
            ps = [torch.nn.Parameter(torch.rand(128, 256, 3).cuda()), torch.nn.Parameter(torch.rand(256, 256, 1).cuda())]
            opt = FusedAdam(ps, lr=1e-2, fuse_mip_fold=True)
            (uv, da), cache = _coords(7000, 9, -4.0, -1.0), {}
            w1 = torch.randn(7000, 1, generator=torch.Generator().manual_seed(1)).cuda()
            w2 = torch.randn(7000, 1, generator=torch.Generator().manual_seed(2)).cuda()
            for it in range(3):
                opt.zero_grad()
                a, r = T.texture_batch(ps, uv, da, "linear-mipmap-linear", 13, cache=cache, fanout=[1, 2] if fan else None)
                r1, r2 = r if fan else (r, r)
                assert not fan or (r1 is not r2 and r1.data_ptr() == r2.data_ptr())
                ((a * a).sum() + (r1 * w1).sum() + (r2 * r2 * w2).sum()).backward()
                opt.step()
            torch.cuda.synchronize()
            return [p.detach().clone() for p in ps] + [opt.state[p]["exp_avg"].clone() for p in ps]
        for x, y in zip(run(False), run(True)):
            assert torch.equal(x, y), float((x - y).abs().max())
    finally:
        T._GRAD_MASK = was
