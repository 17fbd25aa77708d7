"""GPU: the product's G-buffer (texir_gbuffer_cast: primary-ray casting through the BVH) against the rasteriser-semantics oracle
(oracle/raster.py: clip-space edge functions, top-left rule, z-buffer, perspective-correct barycentrics, analytic rast_db) -- the
semantics of the nvdiffrast calls it replaces (models/mat_nvdiffrast.py:119-128, models/tracer_o3d_irt.py:99-112), restated without
sharing a code path with ray casting.  Reports the disagreeing-pixel fraction and the rendered-RGB error that fraction causes."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _compare(tx, sc0, c, eye, cn_seed=1):
    from oracle import raster as R
    from texir_code_amd import cameras, gbuffer as GB
    sc = tx.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    rng = np.random.default_rng(cn_seed)
    cn = rng.normal(size=(3 * sc0["tris"].shape[0], 3)).astype(np.float32)
    GB.set_corner_normals(sc, cn)
    E = np.eye(4, dtype=np.float32)
    E[:3, 3] = eye
    mvp, cam = cameras.cube_mvps(E)
    out = GB.cast_gbuffer(sc, mvp, c, flip_v=True)
    ref = R.gbuffer(sc0["verts"], sc0["tris"], sc0["tri_uvs"], mvp.numpy(), c, corner_normals=cn, flip_v=True)
    tri = out["tri_id"].reshape(-1).cpu().numpy()
    same = tri == ref["tri_id"]
    return sc, mvp, cam, out, ref, tri, same


def _rendered_rgb(sc, sc0, gb, cam, c, seed=3):
    """stage-2 render (diffuse irradiance x albedo / pi + traced GGX specular) of a G-buffer given as a dict of tensors"""
    from texir_code_amd import conf as C
    from texir_code_amd.models import MaterialModel
    conf = C.parse_string("train{ pano_img_res = [%d,%d]\n sample_light = [64,16]\n hdr_exposure = 0 }\nmodels{ render{ sample_type = [uniform, importance] } }" % (2 * c, 4 * c))
    g = torch.Generator().manual_seed(seed)
    m = MaterialModel.from_arrays(sc, sc0["hdr"], torch.rand(64, 64, 3, generator=g) + 0.3, conf, albedo_res=256, roughness_res=256)
    with torch.no_grad():
        m.materials_a.copy_((0.2 + 0.6 * torch.rand(256, 256, 3, generator=g)).cuda())
        m.materials_r.copy_((0.1 + 0.5 * torch.rand(256, 256, 1, generator=g)).cuda())
    m._gbuffer = lambda mvp, vid: gb
    m._static_shift = torch.rand(6 * c * c, 2, generator=g).cuda()
    with torch.no_grad():
        return m(None, "v", cam, 2)["rgb"].reshape(-1, 3).cpu().numpy()


def _as_gb(ref, c):
    t = lambda a, k: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda().reshape(6, c, c, k)
    return {"position": t(ref["position"], 3), "normal": t(ref["normal"], 3), "mask": t(ref["mask"], 1), "uv": t(ref["uv"], 2), "uv_da": t(ref["uv_da"], 4),
            "tri_id": torch.from_numpy(ref["tri_id"].astype(np.int32)).cuda().reshape(6, c, c)}


def test_gbuffer_vs_raster_semantics_room(golden, tx):
    g = golden("irt_room.npz")
    sc0 = {k: g[k] for k in ("verts", "tris", "tri_uvs", "hdr")}
    c = 32
    sc, mvp, cam, out, ref, tri, same = _compare(tx, sc0, c, [3.1, 1.4, 2.2])
    frac = 1.0 - same.mean()
    assert frac < 5e-3, frac                                         # measured: see the printed line
    assert (tri > 0).mean() > 0.99
    for k, tol in (("position", 1e-5), ("normal", 1e-4), ("uv", 1e-5), ("uv_da", 1e-3)):
        a = out[k].reshape(tri.size, -1).cpu().numpy()[same]
        assert rel_l2(a, ref[k].reshape(tri.size, -1)[same]) < tol, (k, rel_l2(a, ref[k].reshape(tri.size, -1)[same]))
    # what the disagreeing pixels cost in the rendered image: same materials, same lighting, same GGX shifts, the two G-buffers
    rgb_a = _rendered_rgb(sc, sc0, {k: (v if k != "tri_id" else v) for k, v in out.items()}, cam, c)
    rgb_b = _rendered_rgb(sc, sc0, _as_gb(ref, c), cam, c)
    err_all, err_same = rel_l2(rgb_a, rgb_b), rel_l2(rgb_a[same], rgb_b[same])
    print("room 20k: %.4f %% of %d pixels pick another triangle; rendered RGB rel-L2 %.2e (agreeing pixels only: %.2e)" % (100 * frac, tri.size, err_all, err_same))
    assert err_same < 1e-3                                            # north-star bar on the pixels both algorithms assign to the same triangle
    assert err_all < 2e-2


def test_gbuffer_vs_raster_semantics_silhouettes(tx):
    """rotated boxes floating in the room, a blind of thin slats, two openings (background pixels): silhouette edges cross pixel centres
    everywhere -- the case where ray casting and rasterisation could disagree (tie-breaks on shared edges, grazing triangles)"""
    from texir_code_amd import synth
    sc0 = synth.make_scene(6000, seed=666, tex_res=128, style="scan")
    c = 48
    sc, mvp, cam, out, ref, tri, same = _compare(tx, sc0, c, [4.2, 1.3, 2.9])
    frac = 1.0 - same.mean()
    bg_a, bg_b = (tri == 0), (ref["tri_id"] == 0)
    assert bg_b.sum() > 20                                            # the openings show background
    assert (bg_a != bg_b).mean() < 2e-3                               # coverage itself (hit vs background) agrees
    assert frac < 1e-2, frac
    for k, tol in (("position", 1e-5), ("normal", 1e-4), ("uv", 1e-5), ("uv_da", 1e-3)):
        a = out[k].reshape(tri.size, -1).cpu().numpy()[same]
        assert rel_l2(a, ref[k].reshape(tri.size, -1)[same]) < tol, (k, rel_l2(a, ref[k].reshape(tri.size, -1)[same]))
    rgb_a = _rendered_rgb(sc, sc0, out, cam, c)
    rgb_b = _rendered_rgb(sc, sc0, _as_gb(ref, c), cam, c)
    err_all, err_same = rel_l2(rgb_a, rgb_b), rel_l2(rgb_a[same], rgb_b[same])
    print("scan 6k: %.4f %% of %d pixels pick another triangle; rendered RGB rel-L2 %.2e (agreeing pixels only: %.2e)" % (100 * frac, tri.size, err_all, err_same))
    assert err_same < 1e-3
    assert err_all < 5e-2
