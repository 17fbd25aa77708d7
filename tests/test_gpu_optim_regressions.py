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
