"""GPU parity on the `house` scene family (texir_code_amd/synth.py::_house: 3 x 3 rooms joined by doors, large untessellated shell triangles next to
millimetre-scale clutter, windows to the outside) -- the shape of the reference's data (README.md:21-34, configs/mat_hdrhouse.conf).
Small sibling against the C oracle in every kernel form + hit / miss agreement; full-size properties of bench workload `house`."""
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def house20k(tx):
    from oracle import oracle as O
    from texir_code_amd import synth
    sc0 = synth.make_scene(20000, seed=666, tex_res=256, style="house")
    pos, nrm, valid = synth.make_texel_gbuffer(sc0, 256)
    shift = synth.make_shifts(256 * 256)
    sc = tx.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    osc = O.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    return sc0, sc, osc, pos.reshape(-1, 3), nrm.reshape(-1, 3), valid.reshape(-1), shift


@pytest.mark.parametrize("form", ["1", "64", "binary"])
@pytest.mark.parametrize("N", [64, 2048])
def test_house_scene_irt_vs_oracle(house20k, tx, form, N, monkeypatch):
    sc0, sc, osc, pos, nrm, valid, shift = house20k
    if form == "binary":
        monkeypatch.setenv("TEXIR_BVH_WIDTH", "2")
        sc = tx.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    else:
        monkeypatch.setenv("TEXIR_IRT_TEXELS_PER_WAVE", form)
    n_tex = 1500 if N == 64 else 200
    v = np.argwhere(valid > 0)[:, 0]
    v = v[:: max(1, v.size // n_tex)][:n_tex]
    v = np.unique(np.concatenate([v, np.argwhere(valid > 0)[:, 0][5000:5000 + 130]]))
    ids = torch.from_numpy(v.astype(np.int32)).cuda()
    irr, st = sc.irt_generate(torch.from_numpy(pos), torch.from_numpy(nrm), torch.from_numpy(shift), N, "uniform", texel_ids=ids, stats=True)
    irr = irr.cpu().numpy()
    vm = np.zeros(valid.size, np.uint8)
    vm[v] = 1
    ref = osc.irt_generate(pos, nrm, vm, shift, N, "uniform", tracer="bvh")
    assert rel_l2(irr[v], ref[v]) < 1e-4, rel_l2(irr[v], ref[v])          # (north-star bar: 1e-3)
    rays, _, _, hits = [int(x) for x in st[:4].tolist()]
    assert rays == v.size * N and hits < rays                              # some rays leave through the windows / the entrance


def test_house_scene_hits_agree_with_oracle_and_bruteforce(house20k):
    """closest hits of random interior rays: the same triangle as the oracle's BVH and as its brute-force loop wherever the hit is unambiguous;
    the large shell triangles (areas of square metres next to mm^2 clutter) and the door openings are what this scene adds"""
    sc0, sc, osc, pos, nrm, valid, shift = house20k
    rng = np.random.default_rng(4)
    R = 20000
    org = np.stack([rng.uniform(0.3, 15.7, R), rng.uniform(0.2, 2.8, R), rng.uniform(0.3, 11.7, R)], -1).astype(np.float32)
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rad, t, pid, uv = sc.trace_shade(torch.from_numpy(org), torch.from_numpy(d), return_hits=True)
    t, pid = t.cpu().numpy(), pid.cpu().numpy()
    to, po, _ = osc.cast_rays(org, d, tracer="bvh")
    tb, pb, _ = osc.cast_rays(org[:2000], d[:2000], tracer="brute")
    hit, hit_o = np.isfinite(t), np.isfinite(to)
    assert (hit != hit_o).mean() < 1e-3
    both = hit & hit_o
    assert np.abs(t[both] - to[both]).max() < 1e-3 * max(1.0, to[both].max())
    assert (pid[both] != po[both].astype(pid.dtype)).mean() < 5e-3           # ties on shared edges may name the neighbour
    bb = np.isfinite(tb) & hit[:2000]
    assert np.abs(t[:2000][bb] - tb[bb]).max() < 1e-3 * max(1.0, tb[bb].max())
    assert 0.0 < (~hit).mean() < 0.2                                          # rays do escape through the openings, most do not
    ref = osc.trace_shade(org, d, tracer="bvh")
    assert rel_l2(rad.cpu().numpy(), ref) < 1e-3


def test_full_size_house_properties():
    """bench workload `house` at full size (4096^2 texels, 1 M triangles): every ray traced, p_hit < 1, seams zero, exact linearity, determinism and the
    8-way shard union bit for bit at 2048 spp; a random sample of texels against the C oracle at 256 spp"""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle as O
    from texir_code_amd import scene as S, dist_util
    sc0, pos, nrm, valid, shift, res, spp = bench.make_workload("house")
    sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    ids = dist_util.morton_order(torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32), res).cuda()
    d = lambda a: torch.from_numpy(a).cuda()
    dpos, dnrm, dshift = d(pos).reshape(-1, 3), d(nrm).reshape(-1, 3), d(shift)
    v = torch.from_numpy(valid.reshape(-1) > 0).cuda()
    N = 256
    base, st = sc.irt_generate(dpos, dnrm, dshift, N, "uniform", texel_ids=ids, stats=True)
    rays, _, _, hits = [int(x) for x in st[:4].tolist()]
    assert rays == ids.numel() * N and 0.9 < hits / rays < 0.9999
    assert torch.isfinite(base).all() and bool((base[~v] == 0).all()) and float(base[v].min()) >= 0
    hdr = torch.from_numpy(sc0["hdr"]).cuda()
    sc.set_texture(hdr * 4.0)
    assert torch.equal(sc.irt_generate(dpos, dnrm, dshift, N, "uniform", texel_ids=ids), base * 4.0)
    sc.set_texture(hdr)
    rng = np.random.default_rng(2)
    vi = np.argwhere(valid.reshape(-1) > 0)[:, 0]
    pick = np.sort(rng.choice(vi, 600, replace=False))
    osc = O.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    ref = osc.irt_generate(pos.reshape(-1, 3)[pick], nrm.reshape(-1, 3)[pick], None, shift[pick], N, "uniform", tracer="bvh")
    got = base[torch.from_numpy(pick).cuda()].cpu().numpy()
    assert rel_l2(got, ref) < 1e-4, rel_l2(got, ref)
    del base
    N = 2048
    full = sc.irt_generate(dpos, dnrm, dshift, N, "uniform", texel_ids=ids)
    assert torch.equal(full, sc.irt_generate(dpos, dnrm, dshift, N, "uniform", texel_ids=ids))
    # 300 texels of the timed 2048-spp output against the oracle (the bench line's workload, the bench line's sample count)
    pick = np.sort(rng.choice(vi, 300, replace=False))
    ref = osc.irt_generate(pos.reshape(-1, 3)[pick], nrm.reshape(-1, 3)[pick], None, shift[pick], N, "uniform", tracer="bvh")
    got = full[torch.from_numpy(pick).cuda()].cpu().numpy()
    assert rel_l2(got, ref) < 1e-4, rel_l2(got, ref)
    acc = torch.zeros_like(full)
    for r in range(8):
        sc.irt_generate(dpos, dnrm, dshift, N, "uniform", texel_ids=dist_util.shard_block_cyclic(ids, r, 8, 4096), out=acc)
    assert torch.equal(acc, full)
