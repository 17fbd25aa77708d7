"""Parity of the HIP path (through the C-ABI of libtexir_hip.so) against the CPU oracle and the golden
vectors captured from the reference.  Needs an MI355X: run with `pytest -m gpu`."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tx():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from texir_code_amd import scene as S
    return S


@pytest.fixture(scope="module")
def room(golden, tx):
    g = golden("irt_room.npz")
    from oracle import oracle as O
    return g, tx.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"]), O.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])


def test_generate_dir_golden(golden, tx):
    g = golden("gen_dir.npz")
    nrm = torch.from_numpy(g["normals"]).cuda()
    rough = torch.from_numpy(g["roughness"]).cuda()
    for c in range(int(g["n_cases"])):
        mode, N = str(g["c%d_mode" % c]), int(g["c%d_N" % c])
        L = tx.generate_dir(nrm, N, torch.from_numpy(g["c%d_shift" % c]).cuda(), mode, rough if mode == "importance" else None)
        ref = g["c%d_L" % c]
        err = np.abs(L.cpu().numpy() - ref).max()
        assert err < 5e-6, (mode, N, err)


def test_trace_shade_vs_bruteforce(room):
    g, sc, osc = room
    rng = np.random.default_rng(3)
    valid = np.argwhere(g["valid"].reshape(-1) > 0)[:, 0]
    pick = rng.choice(valid, 4096)
    org = g["pos"].reshape(-1, 3)[pick]
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d *= rng.uniform(0.5, 2.0, (4096, 1)).astype(np.float32)        # query_irf takes un-normalised dirs
    rad, t, pid, uv = sc.trace_shade(torch.from_numpy(org), torch.from_numpy(d), return_hits=True)
    t_ref, pid_ref, uv_ref = osc.cast_rays(org, d, tracer="brute")
    rad_ref = osc.shade_hits(t_ref, pid_ref, uv_ref)
    same = (pid.cpu().numpy().astype(np.uint32) == pid_ref)
    assert same.mean() > 0.998
    tt = t.cpu().numpy()
    m = same & np.isfinite(t_ref)
    assert np.abs(tt[m] - t_ref[m]).max() < 1e-4 * max(1.0, t_ref[m].max())
    # the barycentrics that leave the library are the CALLER's (weights of its corners 1 and 2, Open3D's primitive_uvs), whatever rotation the leaf
    # record stores the triangle in (csrc/bvh_build.h: quad leaves)
    assert np.abs(uv.cpu().numpy()[m] - uv_ref[m]).max() < 2e-4
    assert rel_l2(rad.cpu().numpy(), rad_ref) < 1e-3


def test_trace_shade_query_irf_edge_cases(golden, tx):
    """t<=1e-4 is a miss, zero-length directions miss, far rays miss -> radiance 0"""
    g = golden("irt_box.npz")
    sc = tx.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    org = torch.tensor([[4.0, 1.5, 3.0], [4.0, 1.5, 3.0], [4.0, 5e-5, 3.0], [40.0, 1.5, 3.0]])
    d = torch.tensor([[0.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, -1.0, 0.0], [1.0, 0.0, 0.0]])
    rad, t, pid, uv = sc.trace_shade(org, d, return_hits=True)
    rad = rad.cpu().numpy()
    assert np.all(rad[0] == 0) and np.any(rad[1] > 0) and np.all(rad[2] == 0) and np.all(rad[3] == 0)
    assert abs(float(t[1]) - 1.5) < 1e-5 and float(t[2]) <= 1e-4
    assert int(pid[3]) == -1 and math.isinf(float(t[3]))


@pytest.mark.parametrize("per_wave", ["0", "1", "64"])
@pytest.mark.parametrize("name", ["irt_box.npz", "irt_room.npz"])
def test_irt_matches_reference_forward(golden, tx, name, per_wave, monkeypatch):
    """whole TracerO3d.forward loop (reference code, stub-imported) vs texir_irt_generate on identical shifts, for the automatic
    choice and for both kernel forms (1 / 64 texels per wave)"""
    monkeypatch.setenv("TEXIR_IRT_TEXELS_PER_WAVE", per_wave)
    g = golden(name)
    sc = tx.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    ids = torch.nonzero(torch.from_numpy(g["valid"].reshape(-1)) > 0)[:, 0].cuda()
    irr, st = sc.irt_generate(torch.from_numpy(g["pos"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["shift"]), int(g["N"]),
                              str(g["mode"]), texel_ids=ids, stats=True)
    irr = irr.cpu().numpy()
    ref = g["irr"].reshape(-1, 3)
    assert rel_l2(irr, ref) < 1e-3            # north_star tolerance
    assert rel_l2(irr, ref) < 2e-5            # what we actually expect from float32 re-ordering
    assert np.all(irr[g["valid"].reshape(-1) == 0] == 0)
    st = st.cpu().numpy()
    assert st[0] == ids.numel() * int(g["N"])
    # the counting build and the production build of the kernel agree bit-for-bit on the listed texels
    irr2 = sc.irt_generate(torch.from_numpy(g["pos"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["shift"]), int(g["N"]), str(g["mode"]),
                           texel_ids=ids).cpu().numpy()
    assert np.array_equal(irr, irr2)


def test_irt_all_texels_without_id_list(golden, tx):
    g = golden("irt_box.npz")
    sc = tx.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    irr = sc.irt_generate(torch.from_numpy(g["pos"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["shift"]), int(g["N"])).cpu().numpy()
    v = g["valid"].reshape(-1) > 0
    assert rel_l2(irr[v], g["irr"].reshape(-1, 3)[v]) < 2e-5
    assert np.all(irr[~v] == 0)            # zero normal -> zero direction -> miss


@pytest.mark.parametrize("per_wave", ["1", "64"])
@pytest.mark.parametrize("N,mode", [(100, "uniform"), (128, "cosine"), (2048, "uniform"), (1, "uniform"), (2, "cosine"), (512, "uniform"), (1024, "cosine")])
def test_irt_vs_oracle_various_N(room, N, mode, per_wave, monkeypatch):
    """power-of-two and other sample counts (natural sample order; 1, 2, 4 and 8 pass ranges per texel in the 64-texel form), a
    ragged texel list (70 = one full + one partial 64-texel wave), every kernel form"""
    monkeypatch.setenv("TEXIR_IRT_TEXELS_PER_WAVE", per_wave)
    g, sc, osc = room
    v = np.argwhere(g["valid"].reshape(-1) > 0)[:, 0][::37][:70]
    ids = torch.from_numpy(v.astype(np.int32)).cuda()
    irr = sc.irt_generate(torch.from_numpy(g["pos"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["shift"]), N, mode, texel_ids=ids).cpu().numpy()
    valid = np.zeros(g["valid"].size, np.uint8)
    valid[v] = 1
    ref = osc.irt_generate(g["pos"], g["nrm"], valid, g["shift"], N, mode, tracer="bvh")
    assert rel_l2(irr[v], ref[v]) < 1e-4


def test_scalar_float_node_path_changes_nothing(room, tx, monkeypatch):
    """wave-uniform node steps read the FLOAT form of the node through the scalar cache (TEXIR_UNIFORM_SLOAD = 2); a scene created with
    TEXIR_UNIFORM_FLOAT=0 has no float nodes and every step takes the per-lane quantised path.  Both prune with conservative boxes, so the
    closest hits -- and with them every irradiance bit -- must be the same."""
    g, sc, _ = room
    monkeypatch.setenv("TEXIR_UNIFORM_FLOAT", "0")
    sc_q = tx.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    monkeypatch.delenv("TEXIR_UNIFORM_FLOAT")
    assert sc_q.info()["node_bytes"] < sc.info()["node_bytes"]          # (the default scene carries both forms)
    v = np.argwhere(g["valid"].reshape(-1) > 0)[:, 0]
    ids = torch.from_numpy(v.astype(np.int32)).cuda()
    args = (torch.from_numpy(g["pos"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["shift"]), 256, "uniform")
    a = sc.irt_generate(*args, texel_ids=ids).cpu().numpy()
    b = sc_q.irt_generate(*args, texel_ids=ids).cpu().numpy()
    assert np.array_equal(a[v], b[v])
    # single rays as well (incoherent: the scalar path is taken only by chance)
    rng = np.random.default_rng(5)
    o = np.repeat(g["pos"].reshape(-1, 3)[v[:4096]] + 1e-3 * g["nrm"].reshape(-1, 3)[v[:4096]], 4, 0).astype(np.float32)
    d = rng.normal(size=o.shape).astype(np.float32)
    ra = sc.trace_shade(torch.from_numpy(o), torch.from_numpy(d)).cpu().numpy()
    rb = sc_q.trace_shade(torch.from_numpy(o), torch.from_numpy(d)).cpu().numpy()
    assert np.array_equal(ra, rb)


def test_empty_inputs_are_noops(room, tx):
    """zero texels / rays / pixels / taps: every entry point returns an empty (or untouched) result instead of launching a 0-block grid or
    raising (the reference's tensors of length 0 flow through torch the same way)"""
    from texir_code_amd.scene import generate_dir, spec_render
    from texir_code_amd.texture import texture
    g, sc, _ = room
    pos, nrm, shift = torch.from_numpy(g["pos"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["shift"])
    none = torch.zeros(0, dtype=torch.int32, device="cuda")
    out = torch.full((pos.reshape(-1, 3).shape[0], 3), 7.0, device="cuda")
    irr = sc.irt_generate(pos, nrm, shift, 64, "uniform", texel_ids=none, out=out)
    assert irr.shape == out.shape and bool((irr == 7.0).all())                   # nothing listed, nothing written
    z3 = torch.zeros(0, 3, device="cuda")
    assert sc.trace_shade(z3, z3).shape == (0, 3)
    rad, t, pid, uv = sc.trace_shade(z3, z3, return_hits=True)
    assert rad.shape == (0, 3) and t.numel() == 0 and pid.numel() == 0 and uv.numel() == 0
    assert generate_dir(z3, 16, torch.zeros(0, 2), "cosine").shape == (0, 16, 3)
    rgb = spec_render(sc, z3, z3, torch.zeros(0, device="cuda"), z3, z3, torch.zeros(3, device="cuda"), torch.zeros(0, 2, device="cuda"), 16)
    assert rgb.shape == (0, 3)
    tex = torch.rand(16, 16, 3, device="cuda", requires_grad=True)
    o = texture(tex, torch.zeros(0, 2, device="cuda"), torch.zeros(0, 4, device="cuda"), "linear-mipmap-linear", 4)
    assert o.shape == (0, 3)
    o.sum().backward()
    assert tex.grad is not None and float(tex.grad.abs().sum()) == 0.0
    # ragged: a texel list that is not a multiple of the 64-texel wave, shorter than one wave, and one texel
    v = np.argwhere(g["valid"].reshape(-1) > 0)[:, 0]
    ref = sc.irt_generate(pos, nrm, shift, 64, "uniform", texel_ids=torch.from_numpy(v.astype(np.int32)).cuda()).cpu().numpy()
    for n in (1, 63, 65, 130):
        ids = torch.from_numpy(v[:n].astype(np.int32)).cuda()
        part = sc.irt_generate(pos, nrm, shift, 64, "uniform", texel_ids=ids).cpu().numpy()
        assert np.array_equal(part[v[:n]], ref[v[:n]]), n                        # the estimator is per texel: list length and wave packing do not matter


def test_irt_constant_radiance_closed_room(tx):
    """analytic KAT: closed room, constant radiance L  =>  E -> pi*L  (sum ndl*2pi/N -> pi)"""
    from texir_code_amd import synth
    sc0 = synth.make_scene(12, tex_res=16)
    hdr = np.full((16, 16, 3), 0.5, np.float32)
    sc = tx.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], hdr)
    pos, nrm, valid = synth.make_texel_gbuffer(sc0, 32)
    ids = torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].cuda()
    shift = synth.make_shifts(32 * 32)
    irr = sc.irt_generate(torch.from_numpy(pos), torch.from_numpy(nrm), torch.from_numpy(shift), 4096, "uniform", texel_ids=ids).cpu().numpy()
    v = valid.reshape(-1) > 0
    assert abs(irr[v].mean() - math.pi * 0.5) < 2e-3
    assert np.abs(irr[v] - math.pi * 0.5).max() < 2e-2


def test_spec_forward_vs_oracle(room):
    g, sc, osc = room
    from texir_code_amd import scene as S
    rng = np.random.default_rng(5)
    v = np.argwhere(g["valid"].reshape(-1) > 0)[:, 0][::13][:300]
    P = v.size
    n = g["nrm"].reshape(-1, 3)[v]
    pts = g["pos"].reshape(-1, 3)[v]
    alb = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    r = rng.uniform(0.01, 0.8, P).astype(np.float32)
    irr = rng.uniform(0, 3, (P, 3)).astype(np.float32)
    cam = np.array([4.0, 1.5, 3.0], np.float32)
    shift = rng.uniform(0, 1, (P, 2)).astype(np.float32)
    for Sn in (16, 24, 256):
        ref, ls_ref = osc.spec_forward(n, alb, r, pts, irr, cam, shift, Sn, tracer="bvh", return_ls=True)
        t = lambda a: torch.from_numpy(a).cuda()
        rgb = S.spec_render(sc, t(n), t(alb), t(r), t(pts), t(irr), t(cam), t(shift), Sn)
        assert rel_l2(rgb.cpu().numpy(), ref) < 1e-3, Sn
        assert rel_l2(rgb.cpu().numpy(), ref) < 5e-5, Sn


def test_spec_forward_on_reference_lighting(golden, tx):
    """direct pin of the HIP render + specular_reflectance: the reference's own traced radiance `Ls` from the
    fixture goes in as the lighting (texir_spec_forward, ls_given = 1), the reference's `rgb` must come out -- no tracer in between"""
    g = golden("spec_render.npz")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    rgb = tx.spec_render(None, t(g["normal"]), t(g["albedo"]), t(g["roughness"].reshape(-1)), t(g["points"]), t(g["irr"]), t(g["cam"]), t(g["shift"]),
                         int(g["S"]), lighting=t(g["Ls"]))
    assert rel_l2(rgb.cpu().numpy(), g["rgb"]) < 1e-3
    assert rel_l2(rgb.cpu().numpy(), g["rgb"]) < 1e-5


def test_spec_backward_matches_reference_autograd(golden, tx):
    """d rgb / d albedo, d roughness from the reference's autograd graph (golden) vs texir_spec_backward"""
    from texir_code_amd import _lib
    g = golden("spec_render.npz")
    P, S = g["normal"].shape[0], int(g["S"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    normal, rough, pts, irr, cam, shift, Ls, d_rgb = (t(g["normal"]), t(g["roughness"].reshape(-1)), t(g["points"]), t(g["irr"]), t(g["cam"]),
                                                     t(g["shift"]), t(g["Ls"]), t(g["d_rgb"]))
    d_a = torch.empty((P, 3), device="cuda")
    d_r = torch.empty((P,), device="cuda")
    _lib.check(_lib.lib().texir_spec_backward(_lib.ptr(normal), _lib.ptr(rough), _lib.ptr(pts), _lib.ptr(irr), _lib.ptr(cam), _lib.ptr(shift),
                                              _lib.ptr(Ls), _lib.ptr(d_rgb), P, S, 1e-14, _lib.ptr(d_a), _lib.ptr(d_r), _lib.stream_ptr()))
    assert rel_l2(d_a.cpu().numpy(), g["d_albedo"]) < 1e-6
    ref = g["d_roughness"].reshape(-1)
    got = d_r.cpu().numpy()
    assert rel_l2(got, ref) < 1e-3
    assert rel_l2(got, ref) < 2e-4


def test_errors_are_loud(golden, tx):
    from texir_code_amd import _lib
    g = golden("irt_box.npz")
    sc = tx.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    with pytest.raises(_lib.TexirError):
        sc.irt_generate(torch.from_numpy(g["pos"]), torch.from_numpy(g["nrm"]), torch.from_numpy(g["shift"]), 0)
    with pytest.raises(_lib.TexirError):
        tx.Scene(g["verts"], g["tris"] + 1000, g["tri_uvs"], g["hdr"])


@pytest.mark.parametrize("loss_type", ["L1", "L2"])
@pytest.mark.parametrize("stage", [0, 1, 2])
def test_render_loss_matches_reference(golden, tx, loss_type, stage):
    """RenderLoss/SegLoss value + gradients (reference autograd, golden) vs the fused HIP loss"""
    from texir_code_amd.loss import RenderLoss
    g = golden("render_loss.npz")
    t = lambda k: torch.from_numpy(g[k]).cuda()
    rgb = t("rgb").requires_grad_(True)
    alb = t("albedo").requires_grad_(True)
    r = t("roughness").requires_grad_(True)
    rw = t("roughness_womipmap").requires_grad_(True)
    preds = {"rgb": rgb, "albedo": alb, "roughness": r, "roughness_womipmap": rw, "empty_mask": t("empty_mask")}
    L = RenderLoss(loss_type=loss_type, w_gradient=1)
    res = L(t("gt"), preds, t("gt_mask"), t("floor_max_mask"), t("seg_mask"), stage, t("room_seg_mask"))
    assert len(res) == (2 if stage == 0 else 3)
    k = "%s_s%d_" % (loss_type, stage)
    assert abs(float(res[0]) - float(g[k + "loss"])) < 1e-5 * max(1.0, abs(float(g[k + "loss"])))
    assert abs(res[1] - float(g[k + "seg"])) < 1e-5 * max(1.0, abs(float(g[k + "seg"])))
    res[0].backward()
    z = lambda x, ref: x.grad.cpu().numpy() if x.grad is not None else np.zeros_like(ref)
    for name, x in (("d_rgb", rgb), ("d_albedo", alb), ("d_roughness", r), ("d_roughness_womipmap", rw)):
        ref = g[k + name]
        got = z(x, ref)
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, name
        else:
            assert rel_l2(got, ref) < 1e-3, (name, rel_l2(got, ref))
            assert rel_l2(got, ref) < 1e-5, (name, rel_l2(got, ref))


def test_render_loss_rejects_non_onehot_masks(golden, tx):
    from texir_code_amd.loss import RenderLoss, compact_masks
    g = golden("render_loss.npz")
    seg = torch.from_numpy(g["seg_mask"]).cuda().clone()
    seg[0] = 1
    seg[1] = 1
    with pytest.raises(ValueError):
        compact_masks(seg)
    with pytest.raises(Exception):
        RenderLoss(loss_type="huber")            # models/loss.py:76: only L1 / L2 / psnr / ssim / msssim exist


def test_texture_fetch_fwd_bwd_vs_torch_restatement(tx):
    """nvdiffrast-style bilinear / trilinear fetch (restated; parity unpinned) vs the torch-CPU restatement + its autograd"""
    from texir_code_amd.texture import texture
    from oracle import ref_torch as RT
    torch.manual_seed(3)
    for (H, W, C) in [(64, 64, 3), (128, 32, 1)]:
        tex = torch.rand(H, W, C)
        P = 500
        uv = torch.rand(P, 2) * 1.4 - 0.2            # exercises wrap
        da = (torch.randn(P, 4) * torch.logspace(-3.5, -0.5, P).unsqueeze(-1)).float()
        da[:5] = 0                                    # zero footprint -> level 0
        G = torch.randn(P, C)
        for mode in ("linear", "linear-mipmap-linear"):
            t_ref = tex.clone().requires_grad_(True)
            o_ref = RT.texture(t_ref, uv, da, mode, 13)
            (o_ref * G).sum().backward()
            t_gpu = tex.cuda().requires_grad_(True)
            o = texture(t_gpu, uv.cuda(), da.cuda(), mode, 13)
            (o * G.cuda()).sum().backward()
            assert rel_l2(o.detach().cpu().numpy(), o_ref.detach().numpy()) < 2e-6, (mode, H, W, C)
            assert rel_l2(t_gpu.grad.cpu().numpy(), t_ref.grad.numpy()) < 2e-6, (mode, H, W, C)


def test_gbuffer_cast_vs_bruteforce(golden, tx):
    from texir_code_amd import gbuffer as GB, cameras
    from oracle import oracle as O, ref_torch as RT
    g = golden("irt_room.npz")
    sc = tx.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    osc = O.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    rng = np.random.default_rng(1)
    cn = rng.normal(size=(3 * g["tris"].shape[0], 3)).astype(np.float32)
    GB.set_corner_normals(sc, cn)
    E = np.eye(4, dtype=np.float32)
    E[:3, 3] = [3.1, 1.4, 2.2]
    mvp, cam = cameras.cube_mvps(E)
    c = 24
    out = GB.cast_gbuffer(sc, mvp, c, flip_v=True)
    ref = RT.gbuffer(osc, g["verts"], g["tris"], g["tri_uvs"], mvp.numpy(), c, corner_normals=cn, flip_v=True)
    tri = out["tri_id"].reshape(-1).cpu().numpy()
    same = tri == ref["tri_id"]
    assert same.mean() > 0.995
    assert (tri > 0).mean() > 0.99            # closed room: every pixel sees a surface
    for k, tol in (("position", 1e-5), ("normal", 1e-4), ("uv", 1e-5), ("uv_da", 2e-3)):
        a = out[k].reshape(tri.size, -1).cpu().numpy()[same]
        b = ref[k].reshape(tri.size, -1)[same]
        assert rel_l2(a, b) < tol, (k, rel_l2(a, b))
    # the six faces tile the sphere: view rays of face centres point along +-x, +-y, +-z
    pos = out["position"].cpu().numpy()
    ctr = np.stack([pos[f, c // 2, c // 2] - cam.numpy() for f in range(6)])
    ctr /= np.linalg.norm(ctr, axis=-1, keepdims=True)
    assert np.allclose(np.abs(ctr).max(-1), 1.0, atol=0.1) and len({tuple(np.round(v).astype(int)) for v in ctr}) == 6


def test_fused_adam_matches_torch_adam(tx):
    from texir_code_amd.optim import FusedAdam
    torch.manual_seed(0)
    p0 = torch.rand(1000, 7, device="cuda")
    a = torch.nn.Parameter(p0.clone())
    b = torch.nn.Parameter(p0.clone())
    oa = torch.optim.Adam([a], lr=3e-2)
    ob = FusedAdam([b], lr=3e-2)
    ob.set_clamp(b, 1e-2, 0.8)
    sa = torch.optim.lr_scheduler.StepLR(oa, 2, 0.8)
    sb = torch.optim.lr_scheduler.StepLR(ob, 2, 0.8)
    for it in range(6):
        g = torch.randn_like(p0)
        a.grad = g.clone()
        b.grad = g.clone()
        oa.step()
        a.data.clamp_(1e-2, 0.8)           # trainer/train_material.py:458
        ob.step()
        sa.step()
        sb.step()
        assert (a - b).abs().max().item() < 2e-6, it          # (step size / bias correction are derived on the device, texir_adam_tick: an ulp of float32 apart at most)


@pytest.fixture(scope="module")
def c2_workload(tx):
    """BASELINE.json configs[1] at full size: 200k triangles, 2048^2 texels, 2048^2 radiance texture"""
    from texir_code_amd import synth, dist_util
    sc0 = synth.make_scene(200000, seed=666, tex_res=2048)
    pos, nrm, valid = synth.make_texel_gbuffer(sc0, 2048)
    shift = synth.make_shifts(2048 * 2048)
    sc = tx.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    ids = torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32)
    ids = dist_util.morton_order(ids, 2048).cuda()
    d = lambda a: torch.from_numpy(a).cuda()
    return sc0, sc, d(pos).reshape(-1, 3), d(nrm).reshape(-1, 3), d(shift), ids, valid


def test_gather_backward_matches_atomic_scatter_and_is_deterministic(tx):
    """texture(..., cache=dict): tap lists sorted once, backward = gather.  Same gradient as the float-atomic scatter (to summation
    order), identical bits from run to run, for both filter modes"""
    from texir_code_amd.texture import texture
    torch.manual_seed(8)
    uv = torch.rand(5000, 2, device="cuda")
    da = (torch.rand(5000, 4, device="cuda") - 0.5) * 0.3
    g = torch.randn(5000, 3, device="cuda")
    for mode in ("linear", "linear-mipmap-linear"):
        grads = []
        for cache in (None, {}, {}):
            t = torch.rand(128, 64, 3, device="cuda", requires_grad=True)
            torch.manual_seed(1)
            with torch.no_grad():
                t.copy_(torch.rand(128, 64, 3, device="cuda"))
            out = texture(t, uv, da, mode, 7, cache=cache)
            out.backward(g)
            grads.append(t.grad.clone())
            if cache is not None:
                out2 = texture(t, uv, da, mode, 7, cache=cache)         # second use of the cached lists
                t.grad = None
                out2.backward(g)
                assert torch.equal(t.grad, grads[-1])
        assert rel_l2(grads[1].cpu().numpy(), grads[0].cpu().numpy()) < 1e-5
        assert torch.equal(grads[1], grads[2])
    # over the cache budget a view silently keeps the atomic scatter
    import texir_code_amd.texture as TX
    budget, TX._TAP_BUDGET = TX._TAP_BUDGET, 0
    try:
        c = {}
        t = torch.rand(128, 64, 3, device="cuda", requires_grad=True)
        texture(t, uv, da, "linear-mipmap-linear", 7, cache=c).backward(g)
        assert list(c.values()) == [None] and rel_l2(t.grad.cpu().numpy() * 0 + 1, np.ones((128, 64, 3))) == 0
    finally:
        TX._TAP_BUDGET = budget


def test_fused_mip_fold_adam_is_bit_identical(tx):
    """FusedAdam(fuse_mip_fold=True): the texture backward leaves level 1 un-folded and the optimiser adds 0.25 * level 1 while it reads
    the gradient.  (a) the kernel pair equals fold + plain step bit for bit on the same gradient buffers; (b) end to end the two
    optimisers agree to the float-atomics tolerance of the scatter."""
    from texir_code_amd import _lib
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.texture import texture
    L = _lib.lib()
    torch.manual_seed(4)
    H = W = 64
    C, levels = 3, int(L.texir_mip_levels(H, W, 6))
    uv = torch.rand(3000, 2, device="cuda")
    da = (torch.rand(3000, 4, device="cuda") - 0.5) * 0.2
    tgt = torch.rand(3000, 3, device="cuda")
    # (a) deferred backward -> (g0, g1 | coarser levels already folded into g1)
    d_out = torch.randn(3000, 3, device="cuda")
    n_rest = int(L.texir_mip_elems(H, W, C, levels))
    g0 = torch.zeros(H, W, C, device="cuda")
    rest = torch.zeros(n_rest, device="cuda")
    _lib.check(L.texir_tex_fetch_backward_deferred(_lib.ptr(g0), _lib.ptr(rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(da), 3000, _lib.ptr(d_out),
                                                   _lib.stream_ptr()))
    g1 = rest[:(H // 2) * (W // 2) * C].clone()
    # the library's own last fold on copies of the same buffers (P = 0: no scatter, folds only; levels = 2 so that only level 1 -> 0 runs)
    folded, r2 = g0.clone(), g1.clone()
    _lib.check(L.texir_tex_fetch_backward(_lib.ptr(folded), _lib.ptr(r2), H, W, C, 2, _lib.ptr(uv), _lib.ptr(da), 1, 0, _lib.ptr(d_out), _lib.stream_ptr()))
    assert not torch.equal(folded, g0)
    outs = []
    for fused in (False, True):
        torch.manual_seed(9)
        p = torch.rand(H, W, C, device="cuda")
        m, v = torch.rand_like(p) * 0.1, torch.rand_like(p) * 0.01
        if fused:
            _lib.check(L.texir_adam_step_tex(_lib.ptr(p), _lib.ptr(g0), None, _lib.ptr(g1), None, _lib.ptr(m), _lib.ptr(v), None, H, W, C, 3e-2, 0.9, 0.999, 1e-8, 3, 0.0, 0.8,
                                             _lib.stream_ptr()))
        else:
            _lib.check(L.texir_adam_step(_lib.ptr(p), _lib.ptr(folded), _lib.ptr(m), _lib.ptr(v), p.numel(), 3e-2, 0.9, 0.999, 1e-8, 3, 0.0, 0.8,
                                         _lib.stream_ptr()))
        outs.append((p, m, v))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # (b) end to end
    res = []
    for fuse in (False, True):
        torch.manual_seed(5)
        t = torch.nn.Parameter(torch.rand(H, W, 3, device="cuda"))
        opt = FusedAdam([t], lr=3e-2, fuse_mip_fold=fuse)
        opt.set_clamp(t, 0.0, float("inf"))
        for _ in range(3):
            loss = (texture(t, uv, da, "linear-mipmap-linear", 6) - tgt).abs().mean()
            opt.zero_grad()
            loss.backward()
            assert (getattr(t, "_texir_grad_l1", None) is not None) == fuse
            opt.step()
            assert getattr(t, "_texir_grad_l1", None) is None
        res.append(t.detach().cpu().numpy())
    assert rel_l2(res[1], res[0]) < 1e-5


@pytest.mark.parametrize("shape", [(64, 64, 3), (128, 64, 1), (32, 96, 2)])
def test_two_level_deferred_fold_is_bit_identical(tx, monkeypatch, shape):
    """TEXIR_DEFER_LEVELS=2 (default): the gather backward stops folding at level 2 and the optimiser step performs level 2 -> 1 -> 0 while it
    reads the gradient.  Same fused multiply-adds in the same order as the fold kernels, so the trajectory equals the one-level deferral
    (TEXIR_DEFER_LEVELS=1) bit for bit -- with the vector and the scalar Adam kernel, with the pyramid and the per-level fold kernels."""
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.texture import texture
    H, W, C = shape
    torch.manual_seed(4)
    uv = torch.rand(5000, 2, device="cuda")
    da = (torch.rand(5000, 4, device="cuda") - 0.5) * 0.3
    tgt = torch.rand(5000, C, device="cuda")
    res = {}
    for name, env in (("one", {"TEXIR_DEFER_LEVELS": "1"}), ("two", {"TEXIR_DEFER_LEVELS": "2"}), ("two_scalar", {"TEXIR_DEFER_LEVELS": "2", "TEXIR_ADAM_SCALAR": "1"}),
                      ("two_per_level", {"TEXIR_DEFER_LEVELS": "2", "TEXIR_MIP_PER_LEVEL": "1"})):
        for k in ("TEXIR_DEFER_LEVELS", "TEXIR_ADAM_SCALAR", "TEXIR_MIP_PER_LEVEL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(5)
        t = torch.nn.Parameter(torch.rand(H, W, C, device="cuda"))
        opt = FusedAdam([t], lr=3e-2, fuse_mip_fold=True)
        opt.set_clamp(t, 0.0, float("inf"))
        cache = {}
        for _ in range(4):
            loss = (texture(t, uv, da, "linear-mipmap-linear", 6, cache=cache) - tgt).abs().mean()
            opt.zero_grad()
            loss.backward()
            assert (getattr(t, "_texir_grad_l2", None) is not None) == (name != "one")
            opt.step()
        res[name] = t.detach().cpu().numpy().copy()
    for name in ("two", "two_scalar", "two_per_level"):
        assert np.array_equal(res[name], res["one"]), name


def test_adam_tex_null_level0_gradient_and_fused_mip_level1(tx, monkeypatch):
    """texir_adam_step_tex: (a) grad = NULL equals an all-zero level-0 gradient bit for bit; (b) the level-1 texels it writes on the
    way equal texir_mip_build's level 1 of the updated texture bit for bit, and a build continued from them (from_level = 1) equals
    the full build; for C = 1 and 3 and a non-square texture"""
    from texir_code_amd import _lib
    L = _lib.lib()
    for (H, W, C) in ((64, 64, 3), (128, 32, 1), (256, 256, 3), (10, 6, 3), (2048, 1360, 3), (64, 64, 4), (32, 64, 2)):
        levels = int(L.texir_mip_levels(H, W, 13))
        n_rest = int(L.texir_mip_elems(H, W, C, levels))
        torch.manual_seed(H + C)
        p0 = torch.rand(H, W, C, device="cuda")
        m0, v0 = torch.rand_like(p0) * 0.1, torch.rand_like(p0) * 0.01
        g1 = torch.randn((H // 2) * (W // 2) * C, device="cuda")
        res = []
        for null_g in (False, True):
            p, m, v = p0.clone(), m0.clone(), v0.clone()
            rest = torch.full((n_rest,), -7.0, device="cuda")
            g0 = None if null_g else torch.zeros(H, W, C, device="cuda")
            _lib.check(L.texir_adam_step_tex(_lib.ptr(p), _lib.ptr(g0), None, _lib.ptr(g1), None, _lib.ptr(m), _lib.ptr(v), _lib.ptr(rest), H, W, C, 3e-2, 0.9, 0.999,
                                             1e-8, 2, 1e-2, 0.8, _lib.stream_ptr()))
            _lib.check(L.texir_mip_build(_lib.ptr(p), _lib.ptr(rest), H, W, C, levels, 1, _lib.stream_ptr()))      # levels 2.. from the fused level 1
            full = torch.empty(n_rest, device="cuda")
            _lib.check(L.texir_mip_build(_lib.ptr(p), _lib.ptr(full), H, W, C, levels, 0, _lib.stream_ptr()))
            assert torch.equal(rest, full), (H, W, C, null_g)
            res.append((p, m, v, rest))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        assert not torch.equal(res[0][0], p0)
        # (c) a sparse level-0 gradient: garbage everywhere except at the masked texels == the dense gradient that is zero elsewhere
        torch.manual_seed(3)
        touched = torch.rand(H * W, device="cuda") < 0.01
        gd = torch.randn(H, W, C, device="cuda") * touched.reshape(H, W, 1)
        gs = torch.where(touched.reshape(H, W, 1), gd, torch.full_like(gd, 1e30))           # never-cleared buffer
        bits = torch.zeros(((H * W + 31) // 32) * 32, device="cuda", dtype=torch.int64)
        bits[: H * W] = touched.long()
        w = (bits.view(-1, 32) << torch.arange(32, device="cuda")).sum(1)
        mask = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
        outs = []
        for g0, mk in ((gd, None), (gs, mask)):
            p, m, v = p0.clone(), m0.clone(), v0.clone()
            _lib.check(L.texir_adam_step_tex(_lib.ptr(p), _lib.ptr(g0.contiguous()), _lib.ptr(mk), _lib.ptr(g1), None, _lib.ptr(m), _lib.ptr(v), None, H, W, C, 3e-2, 0.9,
                                             0.999, 1e-8, 2, 1e-2, 0.8, _lib.stream_ptr()))
            outs.append((p, m, v))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        # (d) the 16-byte-access kernel and the one-float-per-access kernel agree bit for bit (rows that are not 16-byte multiples use the latter)
        both = []
        for scalar in (False, True):
            if scalar:
                monkeypatch.setenv("TEXIR_ADAM_SCALAR", "1")
            else:
                monkeypatch.delenv("TEXIR_ADAM_SCALAR", raising=False)
            p, m, v = p0.clone(), m0.clone(), v0.clone()
            l1 = torch.zeros((H // 2) * (W // 2) * C, device="cuda")
            _lib.check(L.texir_adam_step_tex(_lib.ptr(p), _lib.ptr(gs.contiguous()), _lib.ptr(mask), _lib.ptr(g1), None, _lib.ptr(m), _lib.ptr(v), _lib.ptr(l1), H, W, C,
                                             3e-2, 0.9, 0.999, 1e-8, 5, 0.0, 0.8, _lib.stream_ptr()))
            both.append((p, m, v, l1))
        monkeypatch.delenv("TEXIR_ADAM_SCALAR", raising=False)
        for a, b in zip(*both):
            assert torch.equal(a, b)


def test_multi_level_mip_kernels_match_the_per_level_reference(tx, monkeypatch):
    """the pyramid kernels (five mip levels per launch through LDS; all folds of a gradient stack in one launch) produce the same
    bits as the first implementation (one launch per level, TEXIR_MIP_PER_LEVEL=1), build and folds (complete and deferred), incl.
    non-square sizes and stacks shallower than one tile"""
    from texir_code_amd import _lib
    L = _lib.lib()
    for (H, W, C, mx) in ((4096, 4096, 1, 13), (1024, 2048, 3, 13), (64, 64, 3, 13), (96, 160, 2, 13), (256, 256, 4, 3), (32, 32, 1, 1), (2, 2, 3, 13)):
        levels = int(L.texir_mip_levels(H, W, mx))
        n_rest = int(L.texir_mip_elems(H, W, C, levels))
        torch.manual_seed(H + W + C)
        tex = torch.rand(H, W, C, device="cuda")
        g0s, grs = torch.randn(H, W, C, device="cuda"), torch.randn(n_rest, device="cuda")
        dummy = torch.zeros(8, device="cuda")
        outs = []
        for per_level in ("1", "0"):
            monkeypatch.setenv("TEXIR_MIP_PER_LEVEL", per_level)
            rest = torch.zeros(n_rest, device="cuda")
            _lib.check(L.texir_mip_build(_lib.ptr(tex), _lib.ptr(rest), H, W, C, levels, 0, _lib.stream_ptr()))
            full0, full_r = g0s.clone(), grs.clone()          # P = 0: no scatter, the folds only
            _lib.check(L.texir_tex_fetch_backward(_lib.ptr(full0), _lib.ptr(full_r), H, W, C, levels, _lib.ptr(dummy), _lib.ptr(dummy), 1, 0, _lib.ptr(dummy),
                                                  _lib.stream_ptr()))
            def0, def_r = g0s.clone(), grs.clone()
            if levels >= 2:
                _lib.check(L.texir_tex_fetch_backward_deferred(_lib.ptr(def0), _lib.ptr(def_r), H, W, C, levels, _lib.ptr(dummy), _lib.ptr(dummy), 0,
                                                               _lib.ptr(dummy), _lib.stream_ptr()))
            outs.append((rest, full0, def0, def_r[:max(1, (H // 2) * (W // 2) * C)].clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b), (H, W, C)
        if levels > 1:
            assert not torch.equal(outs[1][1], g0s)             # the folds did something


def test_full_size_c2_properties(c2_workload, tx):
    """size-independent properties at BASELINE full size (the oracle cannot run 6.4 G rays): exact linearity in the radiance
    texture, superposition, determinism, shard-union == whole, and the closed-room constant-radiance limit E -> pi*L"""
    from texir_code_amd import dist_util
    sc0, sc, pos, nrm, shift, ids, valid = c2_workload
    N = 2048
    base = sc.irt_generate(pos, nrm, shift, N, "uniform", texel_ids=ids)
    v = torch.from_numpy(valid.reshape(-1) > 0).cuda()
    assert torch.isfinite(base).all() and bool((base[~v] == 0).all()) and float(base[v].min()) >= 0
    # determinism
    again = sc.irt_generate(pos, nrm, shift, N, "uniform", texel_ids=ids)
    assert torch.equal(base, again)
    # exact linearity: scaling the texture by a power of two scales every partial sum exactly
    hdr = torch.from_numpy(sc0["hdr"]).cuda()
    sc.set_texture(hdr * 2.0)
    twice = sc.irt_generate(pos, nrm, shift, N, "uniform", texel_ids=ids)
    assert torch.equal(twice, base * 2.0)
    # superposition: irr(A) + irr(B) == irr(A + B) up to float rounding
    A = hdr.clone()
    A[:, : hdr.shape[1] // 2] = 0
    B = hdr - A
    sc.set_texture(A)
    ia = sc.irt_generate(pos, nrm, shift, N, "uniform", texel_ids=ids)
    sc.set_texture(B)
    ib = sc.irt_generate(pos, nrm, shift, N, "uniform", texel_ids=ids)
    assert rel_l2((ia + ib).cpu().numpy(), base.cpu().numpy()) < 1e-6
    # closed room, constant radiance L: E = (2 pi / N) * L * sum(ndl) -> pi * L
    sc.set_texture(torch.full_like(hdr, 0.25))
    const = sc.irt_generate(pos, nrm, shift, N, "uniform", texel_ids=ids)[v]
    assert abs(float(const.mean()) - math.pi * 0.25) < 1e-3
    # (a few texels sit where a displaced patch meets a flat border, e.g. box sides just under the displaced floor: the synthetic
    # mesh is not watertight there and part of their hemisphere escapes -- the oracle shows the same outliers)
    assert float(((const - math.pi * 0.25).abs() < 0.03).float().mean()) > 0.999
    sc.set_texture(hdr)
    # multi-GPU partition: the union of the block-cyclic shards reproduces the single-rank texture bit for bit
    acc = torch.zeros_like(base)
    for r in range(4):
        part = dist_util.shard_block_cyclic(ids, r, 4, 4096)
        sc.irt_generate(pos, nrm, shift, N, "uniform", texel_ids=part, out=acc)
    assert torch.equal(acc, base)
    # parity at the configuration's own size and sample count: 300 random valid texels of the 2048-spp texture against the C oracle
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(np.argwhere(valid.reshape(-1) > 0)[:, 0], 300, replace=False))
    tp = torch.from_numpy(pick).cuda()
    osc = O.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    ref = osc.irt_generate(pos[tp].cpu().numpy(), nrm[tp].cpu().numpy(), None, shift[tp].cpu().numpy(), N, "uniform", tracer="bvh")
    e = rel_l2(base[tp].cpu().numpy(), ref)
    print("c2 at 2048 spp, 300 texels vs oracle: rel-L2 %.2e" % e)
    assert e < 1e-3 and e < 1e-4, e


def test_irradiance_at_random_mesh_points_vs_oracle(room):
    """NIrF ground-truth generation (models/tracer_o3d_irrf.py:90-122) is the same sample+trace+integrate kernel evaluated at
    arbitrary surface points instead of texel centres"""
    g, sc, osc = room
    rng = np.random.default_rng(8)
    tris = g["tris"][rng.integers(0, g["tris"].shape[0], 200)]
    V = g["verts"]
    w = rng.dirichlet([1, 1, 1], 200).astype(np.float32)
    p = (V[tris[:, 0]] * w[:, :1] + V[tris[:, 1]] * w[:, 1:2] + V[tris[:, 2]] * w[:, 2:3]).astype(np.float32)
    n = np.cross(V[tris[:, 1]] - V[tris[:, 0]], V[tris[:, 2]] - V[tris[:, 0]])
    n = (n / np.linalg.norm(n, axis=-1, keepdims=True)).astype(np.float32)
    p = p + 1e-2 * n
    shift = rng.uniform(0, 1, (200, 2)).astype(np.float32)
    irr = sc.irt_generate(torch.from_numpy(p), torch.from_numpy(n), torch.from_numpy(shift), 2048, "uniform").cpu().numpy()   # env_res 32x64 = 2048 dirs
    ref = osc.irt_generate(p, n, None, shift, 2048, "uniform", tracer="bvh")
    assert rel_l2(irr, ref) < 1e-4


def test_pathological_meshes_vs_bruteforce(tx):
    """builder/traversal robustness: coincident centroids (forced median splits), long thin slivers, degenerate triangles,
    a single triangle -- closest hits must still agree with the f64 brute-force oracle"""
    from oracle import oracle as O
    rng = np.random.default_rng(42)
    hdr = rng.uniform(0.1, 2.0, (8, 8, 3)).astype(np.float32)
    cases = {}
    # (a) 3000 copies of (nearly) the same triangle stacked along z: every centroid coincides in x,y
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    v = np.concatenate([base + np.array([0, 0, 1e-3 * k], np.float32) for k in range(3000)])
    cases["stack"] = (v, np.arange(9000, dtype=np.int32).reshape(-1, 3))
    # (b) a fan of long thin slivers sharing one apex
    n = 2000
    ang = np.linspace(0, 2 * np.pi, n + 1)
    ring = np.stack([50 * np.cos(ang), 50 * np.sin(ang), np.zeros_like(ang)], -1).astype(np.float32)
    v = np.concatenate([np.zeros((1, 3), np.float32), ring])
    cases["fan"] = (v, np.stack([np.zeros(n, np.int32), np.arange(1, n + 1, dtype=np.int32), np.arange(2, n + 2, dtype=np.int32)], -1))
    # (c) random soup with zero-area and duplicated triangles mixed in
    v = rng.uniform(-1, 1, (600, 3)).astype(np.float32)
    t = rng.integers(0, 600, (1500, 3)).astype(np.int32)
    t[::50, 1] = t[::50, 0]                      # degenerate (two equal corners)
    t[1::97] = t[0]                              # exact duplicates
    cases["soup"] = (v, t)
    # (d) one triangle
    cases["single"] = (base.copy(), np.array([[0, 1, 2]], np.int32))
    for name, (verts, tris) in cases.items():
        uvs = rng.uniform(0, 1, (3 * tris.shape[0], 2)).astype(np.float32)
        sc = tx.Scene(verts, tris, uvs, hdr)
        osc = O.Scene(verts, tris, uvs, hdr)
        lo, hi = verts.min(0), verts.max(0)
        R = 3000
        org = (lo + (hi - lo) * rng.uniform(-0.2, 1.2, (R, 3))).astype(np.float32) + np.array([0, 0, 2.0], np.float32)
        tgt = (lo + (hi - lo) * rng.uniform(0, 1, (R, 3))).astype(np.float32)
        d = tgt - org
        rad, t_gpu, pid, uv = sc.trace_shade(torch.from_numpy(org), torch.from_numpy(d), return_hits=True)
        t_ref, pid_ref, uv_ref = osc.cast_rays(org, d, tracer="brute")
        tg = t_gpu.cpu().numpy()
        hit_ref = np.isfinite(t_ref)
        hit_gpu = np.isfinite(tg)
        assert (hit_ref == hit_gpu).mean() > 0.995, name
        both = hit_ref & hit_gpu
        if both.any():
            # duplicates / stacked copies make the primitive id ambiguous; the hit distance is not
            assert np.abs(tg[both] - t_ref[both]).max() < 1e-3 * max(1.0, float(np.abs(t_ref[both]).max())), name
        assert torch.isfinite(rad).all(), name


def test_full_size_c3_material_pixels_vs_oracle(tx):
    """BASELINE.json configs[2] geometry at full size (1M-triangle mesh, 6x128^2 pixels x 16 GGX samples = 1.57 M traced rays):
    G-buffer by primary-ray casting, then the fused specular forward against the CPU oracle on the SAME G-buffer."""
    from texir_code_amd import synth, cameras, gbuffer as GB, scene as S
    from oracle import oracle as O
    sc0 = synth.make_scene(1000000, seed=666, tex_res=512)
    sc = tx.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    osc = O.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    mvp, cam = cameras.cube_mvps(cameras.grid_cameras(4)[5])
    gb = GB.cast_gbuffer(sc, mvp, 128, flip_v=True)
    assert float(gb["mask"].mean()) > 0.999                       # closed room
    P = 6 * 128 * 128
    n = gb["normal"].reshape(P, 3)
    pts = (gb["position"] + 1e-2 * gb["normal"]).reshape(P, 3)
    g = torch.Generator().manual_seed(1)
    alb = torch.rand(P, 3, generator=g).cuda()
    r = (torch.rand(P, generator=g) * 0.79 + 0.01).cuda()
    irr = (torch.rand(P, 3, generator=g) * 2).cuda()
    shift = torch.rand(P, 2, generator=g).cuda()
    rgb = S.spec_render(sc, n, alb, r, pts, irr, cam.cuda(), shift, 16)
    c = lambda t: t.detach().cpu().numpy()
    ref = osc.spec_forward(c(n), c(alb), c(r), c(pts), c(irr), cam.numpy(), c(shift), 16, tracer="bvh")
    assert rel_l2(c(rgb), ref) < 1e-3
    assert rel_l2(c(rgb), ref) < 2e-4
    # hit points of the G-buffer lie on the mesh: re-cast from the eye and compare distances with the f32 BVH oracle
    d = (gb["position"].reshape(P, 3) - cam.cuda()).cpu().numpy()
    t_ref, _, _ = osc.cast_rays(np.tile(cam.numpy(), (4096, 1)), d[::24][:4096], tracer="bvh")
    assert np.abs(t_ref - 1.0).max() < 1e-3


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_fused_loss_recorded_alone_replays_identically(golden, tx, stage):
    """the loss's workspace must be cleared by every replay of a recorded graph (round 4: cleared by hipMemsetAsync, the stage-1 graph faulted on its
    second replay -- the memset node stopped taking effect and the scatter cursors ran past the workspace; the clear is a kernel now)"""
    from texir_code_amd.loss import RenderLoss
    g = golden("render_loss.npz")
    t = lambda k: torch.from_numpy(g[k]).cuda()
    L = RenderLoss(loss_type="L1", w_gradient=1, lazy_item=True)
    gt, gm, fm, seg, room, empty = t("gt"), t("gt_mask"), t("floor_max_mask"), t("seg_mask"), t("room_seg_mask"), t("empty_mask")
    src = {k: t(k) for k in ("rgb", "albedo", "roughness", "roughness_womipmap")}
    hold = {}

    def body():
        leaves = {k: v.clone().requires_grad_(True) for k, v in src.items()}
        out = L(gt, dict(leaves, empty_mask=empty), gm, fm, seg, stage, room)
        got = torch.autograd.grad(out[0], list(leaves.values()), torch.ones((), device="cuda"), allow_unused=True)
        hold["loss"], hold["grads"] = out[0].detach(), [x for x in got if x is not None]

    body()
    torch.cuda.synchronize()
    want = (float(hold["loss"]), [x.clone() for x in hold["grads"]])
    assert abs(want[0] - float(g["L1_s%d_loss" % stage])) < 1e-5 * max(1.0, abs(float(g["L1_s%d_loss" % stage])))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            body()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(4):
        gr.replay()
        torch.cuda.synchronize()
        assert float(hold["loss"]) == want[0]
        for a, b in zip(hold["grads"], want[1]):
            assert torch.equal(a, b)


def test_spec_backward_on_kept_derivatives_equals_the_recomputing_backward(room, tx):
    """round 4: the training forward keeps d w_i / d roughness (texir_spec_forward_train) and the backward streams over (Ls, dw, d rgb)
    (texir_spec_backward_ws) instead of recomputing the GGX sample chain (texir_spec_backward): same rgb bits, same gradients"""
    from texir_code_amd import scene as S
    g, sc, _ = room
    rng = np.random.default_rng(31)
    v = np.argwhere(g["valid"].reshape(-1) > 0)[:, 0][:3001]
    P = v.size
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    nrm, pts = t(g["nrm"].reshape(-1, 3)[v]), t(g["pos"].reshape(-1, 3)[v])
    alb0, r0 = t(rng.uniform(0, 1, (P, 3))), t(rng.uniform(0.02, 0.7, P))
    irr, sh, G = t(rng.uniform(0, 2, (P, 3))), t(rng.uniform(0, 1, (P, 2))), t(rng.normal(size=(P, 3)))
    cam = torch.tensor([4.0, 1.5, 3.0], device="cuda")
    out = {}
    for form in (True, False):
        S._TRAIN_FORM[0] = form
        try:
            a, r = alb0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            rgb = S.spec_render(sc, nrm, a, r, pts, irr, cam, sh, 16)
            (rgb * G).sum().backward()
            out[form] = (rgb.detach().clone(), a.grad.clone(), r.grad.clone())
        finally:
            S._TRAIN_FORM[0] = True
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    assert float(out[False][2].abs().max()) > 0
    assert rel_l2(out[True][2].cpu().numpy(), out[False][2].cpu().numpy()) < 1e-6
