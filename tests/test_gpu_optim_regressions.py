"""GPU: regressions of the fused texture optimiser's bookkeeping (round-2 advisor findings).
(1) gradient accumulation -- fwd, bwd, fwd, bwd, step without zero_grad in between -- against torch.optim.Adam on dense
autograd gradients of the torch restatement of the fetch; (2) edits of the texture between optimiser step and forward
(in-place on the parameter, or the reference's `.data` clamp followed by refresh_mips) never meet a stale mip level 1."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _fetch_args(P, seed, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand(P, 2, generator=g)
    da = (torch.randn(P, 4, generator=g) * torch.logspace(-3.0, -0.7, P).unsqueeze(-1)).float()
    w = torch.randn(P, 3, generator=g)
    return uv.to(dev), da.to(dev), w.to(dev)


@pytest.mark.parametrize("cached", [False, True])
def test_gradient_accumulation_over_two_backward_passes(tx, cached):
    """the second forward used to clear the gradient arena that still held the first backward's parked level-1/2 gradients"""
    from oracle import ref_torch as RT
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.texture import texture
    torch.manual_seed(5)
    t0 = torch.rand(64, 64, 3)
    a = torch.nn.Parameter(t0.clone())                     # torch reference: dense autograd gradients, torch.optim.Adam
    b = torch.nn.Parameter(t0.clone().cuda())
    oa = torch.optim.Adam([a], lr=3e-2)
    ob = FusedAdam([b], lr=3e-2, fuse_mip_fold=True)
    caches = [{} if cached else None, {} if cached else None]
    fetches = [_fetch_args(700, 1), _fetch_args(900, 2)]
    for it in range(3):
        oa.zero_grad()
        ob.zero_grad()
        for k, (uv, da, w) in enumerate(fetches):
            (RT.texture(a, uv.cpu(), da.cpu(), "linear-mipmap-linear", 13) * w.cpu()).sum().backward()
            (texture(b, uv, da, "linear-mipmap-linear", 13, cache=caches[k]) * w).sum().backward()
        # the full gradient FusedAdam is about to apply (its parts are spread over p.grad, the sparse level-0 buffer and the parked stacks)
        dense = ob.dense_grad(b)
        assert rel_l2(dense.cpu().numpy(), a.grad.numpy()) < 1e-5, it
        oa.step()
        ob.step()
        assert rel_l2(b.detach().cpu().numpy(), a.detach().numpy()) < 1e-5, it


def test_texture_edits_between_step_and_forward_rebuild_the_pyramid(tx):
    """level 1 written by the fused step is trusted for exactly one build, and only when the parameter is unchanged"""
    from oracle import ref_torch as RT
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.texture import refresh_mips, texture
    torch.manual_seed(6)
    p = torch.nn.Parameter(torch.rand(64, 64, 3, device="cuda"))
    opt = FusedAdam([p], lr=0.2, fuse_mip_fold=True)
    uv, da, w = _fetch_args(600, 3)
    da = da.abs() * 4 + 0.05                                # coarse footprints: the result depends on levels >= 1

    def fwd():
        return texture(p, uv, da, "linear-mipmap-linear", 13)

    def ref():
        return RT.texture(p.detach().cpu(), uv.cpu(), da.cpu(), "linear-mipmap-linear", 13).numpy()

    def one_step():
        opt.zero_grad()
        (fwd() * w).sum().backward()
        opt.step()

    one_step()
    one_step()
    assert rel_l2(fwd().detach().cpu().numpy(), ref()) < 2e-6           # plain: step wrote level 1, the build starts there
    one_step()
    with torch.no_grad():
        p.clamp_(0.3, 0.6)                                                # in-place edit of the parameter: version bump => full build
    assert rel_l2(fwd().detach().cpu().numpy(), ref()) < 2e-6
    one_step()
    assert rel_l2(fwd().detach().cpu().numpy(), ref()) < 2e-6           # first forward after the step consumes the flag ...
    p.data.mul_(0.5)                                                      # ... so a later `.data` edit between two forwards is safe too
    assert rel_l2(fwd().detach().cpu().numpy(), ref()) < 2e-6
    one_step()
    p.data.clamp_(0.35, 0.5)                                              # the reference's idiom right after the step (train_material.py:458):
    refresh_mips(p)                                                       # invisible to torch; the documented contract is refresh_mips
    assert rel_l2(fwd().detach().cpu().numpy(), ref()) < 2e-6


def test_dense_grad_of_sparse_level0_state(tx):
    """with cached tap lists the level-0 gradient is sparse (p.grad is None); dense_grad materialises what step() applies"""
    from oracle import ref_torch as RT
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.texture import texture
    torch.manual_seed(7)
    t0 = torch.rand(128, 128, 3)
    a = t0.clone().requires_grad_(True)
    b = torch.nn.Parameter(t0.clone().cuda())
    ob = FusedAdam([b], lr=3e-2, fuse_mip_fold=True)
    uv, da, w = _fetch_args(800, 4)
    da[:6] = 0                                             # a handful of pixels sample level 0
    (RT.texture(a, uv.cpu(), da.cpu(), "linear-mipmap-linear", 13) * w.cpu()).sum().backward()
    (texture(b, uv, da, "linear-mipmap-linear", 13, cache={}) * w).sum().backward()
    assert b.grad is None and b._texir_l0_sparse
    assert rel_l2(ob.dense_grad(b).cpu().numpy(), a.grad.numpy()) < 1e-5
    assert ob.has_pending(b)                              # (what a caller checks before handing the parameter to anything that reads p.grad)
    ob.step()
    assert not ob.has_pending(b)


def test_tap_list_budget_counts_live_lists_only(tx):
    """the global tap-list budget (TEXIR_TAP_CACHE_GB) gets a dropped view cache's share back, so a long-lived process that walks through many
    scenes does not end up on the float-atomic fallback for good"""
    import gc
    from texir_code_amd import texture as T
    p = torch.nn.Parameter(torch.rand(64, 64, 3, device="cuda"))
    uv, da, w = _fetch_args(500, 9)
    before = T._tap_bytes
    cache = {}
    (T.texture(p, uv, da, "linear-mipmap-linear", 13, cache=cache) * w).sum().backward()
    assert T._tap_bytes > before
    del cache
    p.grad = None
    gc.collect()
    assert T._tap_bytes == before


def _small_graphed_world(golden, stage):
    from texir_code_amd import cameras, conf as C
    from texir_code_amd.graph_step import GraphedMatStep
    from texir_code_amd.loss import RenderLoss
    from texir_code_amd.models import MaterialModel
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.scene import Scene
    from texir_code_amd.trainer.train_material import build_masks
    g = golden("irt_room.npz")
    cf = C.parse_string("train{ pano_img_res = [32,64]\n sample_light = [64,16]\n hdr_exposure = 0 }\nmodels{ render{ sample_type = [uniform, importance] } }")
    mvp, cam = cameras.cube_mvps(cameras.grid_cameras(1)[0])
    torch.manual_seed(5)
    sc = Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"], device=0)
    m = MaterialModel.from_arrays(sc, g["hdr"], torch.rand(64, 64, 3) * 2, cf, albedo_res=64, roughness_res=128)
    m.lean_outputs = True
    c = m.cube_res
    gt = torch.rand(6, c, c, 3, device="cuda")
    gmask = torch.ones(6, c, c, 1, device="cuda")
    seg, fm, _ = build_masks(torch.randint(40, 49, (6, c, c, 1)).float().cuda(), torch.rand(6, c, c, 3, device="cuda") - 0.5)
    room = torch.ones((1, 6, c, c, 1), device="cuda")
    m.materials_a.requires_grad = stage in (0, 2)
    m.materials_r.requires_grad = stage in (1, 2)
    opt = FusedAdam(m.parameters(), lr=3e-2, fuse_mip_fold=True)
    opt.set_clamp(m.materials_r, 1e-2, 0.8)
    gs = GraphedMatStep(m, RenderLoss("L1", 1, lazy_item=True, unit_upstream=True), opt, [m.materials_a, m.materials_r])
    gs.capture("v", mvp, cam.cuda(), gt, gmask, seg, fm, room if stage == 2 else None, stage)
    return m, opt, gs, c


@pytest.mark.parametrize("stage", [0, 1])
def test_untrained_texture_is_not_rebuilt_before_every_replay(golden, monkeypatch, stage):
    """the texture a stage does not train (roughness in stage 0, albedo in stage 1) never raises the optimiser's one-shot
    level-1 flag; its eagerly built stack stays valid and GraphedMatStep.step must not run a full mip build for it per replay"""
    from texir_code_amd import texture as T
    m, opt, gs, c = _small_graphed_world(golden, stage)
    calls = []
    real = T.refresh_mips
    monkeypatch.setattr(T, "refresh_mips", lambda p: (calls.append(p), real(p))[1])
    gen = torch.Generator().manual_seed(1)
    frozen = m.materials_r if stage == 0 else m.materials_a
    before = frozen.detach().clone()
    for _ in range(4):
        gs.step("v", stage, shift=torch.rand(6 * c * c, 2, generator=gen))
    torch.cuda.synchronize()
    assert calls == [], [tuple(p.shape) for p in calls]
    assert torch.equal(frozen.detach(), before)
    # ... while a change the version counter can see does trigger exactly one rebuild of that texture
    with torch.no_grad():
        frozen.mul_(0.9)
    gs.step("v", stage, shift=torch.rand(6 * c * c, 2, generator=gen))
    gs.step("v", stage, shift=torch.rand(6 * c * c, 2, generator=gen))
    assert len(calls) == 1 and calls[0] is frozen


def test_load_state_dict_keeps_the_buffers_captured_graphs_point_to(golden):
    """moments and device-resident step records keep their addresses across load_state_dict (captured kernels have them baked in)
    and take the loaded values: a replay after the load continues the loaded trajectory bit for bit"""
    import copy
    from texir_code_amd.texture import refresh_mips
    m, opt, gs, c = _small_graphed_world(golden, 2)
    gen = torch.Generator().manual_seed(2)
    shifts = [torch.rand(6 * c * c, 2, generator=gen) for _ in range(4)]
    for s in shifts[:2]:
        gs.step("v", 2, shift=s)
    torch.cuda.synchronize()
    sd = copy.deepcopy(opt.state_dict())
    snap = [p.detach().clone() for p in (m.materials_a, m.materials_r)]
    for s in shifts[2:]:
        gs.step("v", 2, shift=s)
    torch.cuda.synchronize()
    want = [p.detach().clone() for p in (m.materials_a, m.materials_r)]
    dev = m.materials_a.device
    ptrs = [opt.state[p][k].data_ptr() for p in (m.materials_a, m.materials_r) for k in ("exp_avg", "exp_avg_sq")] + \
           [opt._dev[dev]["state"].data_ptr(), opt._dev[dev]["hyper"].data_ptr()]
    with torch.no_grad():
        for p, s0 in zip((m.materials_a, m.materials_r), snap):
            p.copy_(s0)
    opt.load_state_dict(sd)
    assert ptrs == [opt.state[p][k].data_ptr() for p in (m.materials_a, m.materials_r) for k in ("exp_avg", "exp_avg_sq")] + \
                   [opt._dev[dev]["state"].data_ptr(), opt._dev[dev]["hyper"].data_ptr()]
    assert int(opt.state[m.materials_a]["step"]) == 2 and float(opt._dev[dev]["state"][opt._rec[id(m.materials_a)], 0]) == 2.0
    for s in shifts[2:]:
        gs.step("v", 2, shift=s)
    torch.cuda.synchronize()
    for got, ref in zip((m.materials_a, m.materials_r), want):
        assert torch.equal(got.detach(), ref)
    assert int(opt.state[m.materials_a]["step"]) == 4


def test_level1_stack_is_not_read_when_no_tap_touches_level1(tx):
    """round 4: a view whose taps start at mip level >= 2 leaves the parked level-1 gradient stack all zeros; FusedAdam then passes NULL for it
    (the kernel's read of a quarter of the texture disappears) -- same bits as reading the zeros, and a view that DOES touch level 1 keeps the read"""
    from oracle import ref_torch as RT
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.texture import texture
    torch.manual_seed(8)
    t0 = torch.rand(128, 128, 3)
    for coarse in (True, False):
        uv, da, w = _fetch_args(700, 5)
        da = da.abs() * (40.0 if coarse else 1.0) + (0.08 if coarse else 0.0)         # footprints of >= 4 texels: levels >= 2 only
        res = []
        for skip in (True, False):
            p = torch.nn.Parameter(t0.clone().cuda())
            opt = FusedAdam([p], lr=3e-2, fuse_mip_fold=True)
            cache = {}
            for _ in range(3):
                opt.zero_grad()
                (texture(p, uv, da, "linear-mipmap-linear", 13, cache=cache) * w).sum().backward()
                flagged = bool(getattr(p, "_texir_l1_zero", False))
                if not skip:
                    p._texir_l1_zero = False                                          # force the read of the (all-zero) stack
                opt.step()
            res.append((p.detach().clone(), flagged))
        assert res[0][1] == coarse, (coarse, res[0][1])
        assert torch.equal(res[0][0], res[1][0])
        a = torch.nn.Parameter(t0.clone())
        oa = torch.optim.Adam([a], lr=3e-2)
        for _ in range(3):
            oa.zero_grad()
            (RT.texture(a, uv.cpu(), da.cpu(), "linear-mipmap-linear", 13) * w.cpu()).sum().backward()
            oa.step()
        assert rel_l2(res[0][0].cpu().numpy(), a.detach().numpy()) < 1e-5


def test_masked_gradient_add_equals_the_torch_form(tx):
    """texir_grad_add_masked (round 6: the sparse level-0 gradient folded into a dense one in ONE launch, FusedAdam.step's stage-1 case) against the five elementwise
    torch launches it replaced: identical bits, for 1 and 3 channels, a texel count that is not a multiple of 32, an all-zero and an all-ones mask"""
    from texir_code_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device="cuda").manual_seed(7)
    for H, W, C in ((37, 29, 1), (64, 64, 3), (5, 7, 4)):
        n = H * W
        for kind in ("random", "zeros", "ones"):
            words = (n + 31) // 32
            mask = {"random": torch.randint(-2 ** 31, 2 ** 31 - 1, (words,), device="cuda", dtype=torch.int64, generator=gen).to(torch.int32),
                    "zeros": torch.zeros(words, device="cuda", dtype=torch.int32), "ones": torch.full((words,), -1, device="cuda", dtype=torch.int32)}[kind]
            g = torch.randn(H, W, C, device="cuda", generator=gen)
            g0 = torch.randn(H, W, C, device="cuda", generator=gen)
            g0[0, 0, 0] = float("nan")                        # a never-cleared buffer may hold anything outside the mask ...
            if kind != "ones":
                mask[0] &= ~1                                 # ... so texel 0 is outside it here
            bits = ((mask.view(-1, 1) >> torch.arange(32, device="cuda", dtype=torch.int32)) & 1).bool().reshape(-1)[:n].reshape(H, W, 1)
            want = g + torch.where(bits, g0, torch.zeros((), device="cuda"))
            got = g.clone()
            assert L.texir_grad_add_masked(_lib.ptr(got), _lib.ptr(g0), _lib.ptr(mask), n, C, _lib.stream_ptr()) == 0
            torch.cuda.synchronize()
            same = (got == want) | (torch.isnan(got) & torch.isnan(want))
            assert bool(same.all()), (H, W, C, kind)
            if kind != "ones":
                assert not bool(torch.isnan(got).any())
    assert L.texir_grad_add_masked(None, None, None, 4, 9, _lib.stream_ptr()) < 0 and b"bad size" in L.texir_batch_last_error()
