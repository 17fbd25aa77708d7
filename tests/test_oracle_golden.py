"""Pins the CPU oracle (oracle/texir_oracle.c) against golden vectors captured from the reference's own
functions (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import oracle as O


def test_hammersley(golden):
    g = golden("gen_dir.npz")
    for N in [1, 16, 64, 2048, 100]:
        assert np.array_equal(O.hammersley(N), g["ham_%d" % N])


def test_generate_dir_all_modes(golden):
    g = golden("gen_dir.npz")
    nrm, rough = g["normals"], g["roughness"]
    for c in range(int(g["n_cases"])):
        mode, N = str(g["c%d_mode" % c]), int(g["c%d_N" % c])
        L = O.generate_dir(nrm, N, mode, g["c%d_shift" % c], rough if mode == "importance" else None)
        ref = g["c%d_L" % c]
        assert L.shape == ref.shape
        # sin/cos/sqrt differ by an ulp between libm and torch's vectorised kernels
        assert np.abs(L - ref).max() < 2e-6, (mode, N, np.abs(L - ref).max())
        # zero normal -> zero direction (seam texels)
        assert np.all(L[8] == 0)


def test_query_irf_post_intersection(golden):
    g = golden("query_irf.npz")
    T = g["tri_uvs"].shape[0] // 3
    # geometry is irrelevant for the shading half; make a dummy mesh with T triangles
    verts = np.zeros((3, 3), np.float32)
    tris = np.zeros((T, 3), np.int32)
    tris[:] = [0, 1, 2]
    verts[1, 0] = verts[2, 1] = 1
    s = O.Scene(verts, tris, g["tri_uvs"].astype(np.float32), g["tex"])
    out = s.shade_hits(g["t_hit"], g["prim_id"], g["prim_uv"]).reshape(g["radiance"].shape)
    assert np.abs(out - g["radiance"]).max() < 2e-5 * max(1.0, np.abs(g["radiance"]).max())
    # explicit miss cases
    assert np.all(out[0, 0] == 0) and np.all(out[0, 1] == 0) and np.all(out[0, 2] == 0) and np.any(out[0, 3] != 0)


@pytest.mark.parametrize("name,tol", [("irt_box.npz", 1e-5), ("irt_room.npz", 1e-5)])
def test_irt_forward_bruteforce(golden, name, tol):
    g = golden(name)
    s = O.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    irr = s.irt_generate(g["pos"], g["nrm"], g["valid"], g["shift"], int(g["N"]), str(g["mode"]), tracer="brute")
    ref = g["irr"].reshape(-1, 3)
    assert rel_l2(irr, ref) < tol
    assert np.all(irr[g["valid"].reshape(-1) == 0] == 0)


@pytest.mark.parametrize("name", ["irt_box.npz", "irt_room.npz"])
def test_irt_forward_bvh_matches(golden, name):
    """canonical BVH2 (f32) against the reference loop driven by the f64 brute-force tracer"""
    g = golden(name)
    s = O.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    c = O.new_counters()
    irr = s.irt_generate(g["pos"], g["nrm"], g["valid"], g["shift"], int(g["N"]), str(g["mode"]), tracer="bvh", counters=c)
    assert rel_l2(irr, g["irr"].reshape(-1, 3)) < 1e-3
    assert c[2] == int(g["valid"].sum()) * int(g["N"])
    assert c[0] > c[2] and c[1] > 0


def test_oracle_spec_forward_on_reference_lighting(golden):
    """direct pin of the C oracle's render + specular_reflectance: the fixture's traced radiance `Ls` (what the reference's query_irf
    returned) goes in as the lighting, the reference's own `rgb` must come out (mat_nvdiffrast.py:201-249,260-279)"""
    g = golden("spec_render.npz")
    sc = O.Scene(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.int32),
                 np.zeros((3, 2), np.float32), np.ones((2, 2, 3), np.float32))           # (never traced: the lighting is given)
    rgb = sc.spec_forward(g["normal"], g["albedo"], g["roughness"].reshape(-1), g["points"], g["irr"], g["cam"], g["shift"], int(g["S"]), lighting=g["Ls"])
    assert rel_l2(rgb, g["rgb"]) < 1e-5


def test_spec_forward_matches_reference_render(golden):
    g = golden("spec_render.npz")
    P, S = g["normal"].shape[0], int(g["S"])
    # feed the fixture's Ls through a scene whose every ray returns ... not possible; instead check the
    # closed form against rgb using the oracle's own sampler + BRDF on the stored Ls
    import math
    n, a, r, pts, irr, cam, Ls = g["normal"], g["albedo"], g["roughness"].reshape(-1), g["points"], g["irr"], g["cam"], g["Ls"]
    h = O.generate_dir(n, S, "importance", g["shift"], r)
    assert np.abs(h - g["h"]).max() < 2e-6
    v = cam[None] - pts
    v = v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-4)
    vdh = np.clip((h * v[:, None]).sum(-1), 0, 1)
    l = 2 * vdh[..., None] * h - v[:, None]
    assert np.abs(l - g["l"]).max() < 1e-5
    ndl = np.clip((n[:, None] * l).sum(-1), 0, 1)
    ndh = np.clip((n[:, None] * h).sum(-1), 0, 1)
    ndv = np.clip((n * v).sum(-1), 0, 1)[:, None]
    f = 0.04 + 0.96 * np.power(2.0, (-5.55472 * vdh - 6.98316) * vdh)
    k = ((r + 1) ** 2 / 8)[:, None]
    g1v = ndv / np.maximum(ndv * (1 - k) + k, 1e-14)
    g1l = ndl / np.maximum(ndl * (1 - k) + k, 1e-14)
    brdf = f * g1l * g1v / np.maximum(4 * ndl * ndv, 1e-14)
    w = brdf * ndl * 4 * vdh / np.maximum(ndh, 1e-14)
    rgb = irr * a / math.pi + (Ls * w[..., None]).sum(1) / S
    assert rel_l2(rgb, g["rgb"]) < 1e-5


def test_oracle_diffuse_reflectance_matches_reference(golden):
    """diffuse_reflectance(query_irf(...), l, n, albedo, type) / N of the reference (mat_nvdiffrast.py:252-258, both sample types) ==
    the oracle's per-point irradiance integral x albedo / pi"""
    import math
    g = golden("diffuse.npz")
    s = O.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    for mode in ("uniform", "cosine"):
        E = s.irt_generate(g["points"], g["normal"], None, g["shift_" + mode], int(g["N"]), mode, tracer="brute", cosine_estimator=mode == "cosine")
        assert rel_l2(E * g["albedo"] / math.pi, g["diffuse_" + mode]) < 1e-5, mode


def test_oracle_evaluation_render_matches_reference(golden):
    """the evaluation model's render (models/test_nvdiffrast.py:256-304: BRDF denominators floored at 1e-6, diffuse term traced when
    relighting) at S = 256, with its own query_irf"""
    import math
    g = golden("test_render.npz")
    s = O.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    P = g["normal"].reshape(-1, 3).shape[0]
    f = lambda k, c: g[k].reshape(P, c)
    for tag in ("plain", "relight"):
        irr = f("irr", 3)
        if tag == "relight":
            irr = s.irt_generate(f("points", 3), f("normal", 3), None, g["shift_diff_relight"], int(g["N0"]), "uniform", tracer="brute")
        rgb = s.spec_forward(f("normal", 3), f("albedo", 3), f("roughness", 1).reshape(-1), f("points", 3), irr, g["cam"], g["shift_spec_" + tag], int(g["S"]),
                             tracer="brute", clamp_eps=1e-6)
        assert rel_l2(rgb, g["rgb_" + tag].reshape(P, 3)) < 2e-5, tag
