"""CPU: the rasteriser-semantics oracle (oracle/raster.py: clip-space edge functions at pixel centres, top-left rule, z-buffer,
perspective-correct barycentrics, analytic rast_db) against closed-form cases and against the ray-casting restatement
(oracle/ref_torch.gbuffer) -- two independent algorithms for the same G-buffer."""
import numpy as np
import torch

from conftest import rel_l2


def _mvp(eye=(3.1, 1.4, 2.2)):
    from texir_code_amd import cameras
    E = np.eye(4, dtype=np.float32)
    E[:3, 3] = eye
    mvp, cam = cameras.cube_mvps(E)
    return mvp.numpy().astype(np.float64), cam.numpy()


def test_single_triangle_closed_form():
    """one triangle in front of the +z face of a camera at the origin: coverage, perspective-correct barycentrics, z/w and analytic
    derivatives against direct formulas"""
    from oracle import raster as R
    mvp, _ = _mvp((0.0, 0.0, 0.0))
    v = np.array([[-0.6, -0.5, 1.0], [0.7, -0.4, 2.0], [0.1, 0.8, 3.0]])
    t = np.array([[0, 1, 2]])
    c = 16
    r = R.rasterize(v, t, mvp, c)
    f = 1                                            # face 1 keeps the extrinsic's own axes (front = +z)
    tri = r["tri_id"].reshape(6, c, c)[f]
    assert tri.sum() > 10 and set(np.unique(r["tri_id"])) == {0, 1}
    bar = r["bary"].reshape(6, c, c, 3)[f]
    db = r["bary_dxy"].reshape(6, c, c, 3, 2)[f]
    n, fa = 1e-4, 100.0
    for (i, j) in np.argwhere(tri > 0):
        x, y = (j + 0.5) / c * 2 - 1, (i + 0.5) / c * 2 - 1
        # the ray through the pixel: direction (x, y, 1) (fov 90); intersect with the triangle's plane
        d = np.array([x, y, 1.0])
        nrm = np.cross(v[1] - v[0], v[2] - v[0])
        s = (nrm @ v[0]) / (nrm @ d)
        p = s * d
        A = np.stack([v[0], v[1], v[2]], 1)
        b = np.linalg.solve(A, p)                   # p = sum b_i v_i with sum b = 1 on the plane
        assert abs(b.sum() - 1) < 1e-9 and (b > -1e-12).all()
        assert np.abs(bar[i, j] - b).max() < 1e-9
        zw = ((fa + n) / (fa - n) * p[2] - 2 * fa * n / (fa - n)) / p[2]
        assert abs(r["zw"].reshape(6, c, c)[f][i, j] - zw) < 1e-6          # (the projection matrix is float32, cameras.projection)
        # derivative of the barycentrics by central differences of the same closed form
        h = 1e-6
        for a, (dx, dy) in enumerate(((h, 0.0), (0.0, h))):
            bs = []
            for sgn in (1, -1):
                dd = np.array([x + sgn * dx * 2 / c, y + sgn * dy * 2 / c, 1.0])
                bs.append(np.linalg.solve(A, (nrm @ v[0]) / (nrm @ dd) * dd))
            fd = (bs[0] - bs[1]) / (2 * h)
            assert np.abs(db[i, j, :, a] - fd).max() < 1e-5
    # pixels outside: the point where the pixel ray meets the plane lies outside the triangle
    for (i, j) in np.argwhere(tri == 0)[::7]:
        d = np.array([(j + 0.5) / c * 2 - 1, (i + 0.5) / c * 2 - 1, 1.0])
        nrm = np.cross(v[1] - v[0], v[2] - v[0])
        b = np.linalg.solve(np.stack([v[0], v[1], v[2]], 1), (nrm @ v[0]) / (nrm @ d) * d)
        assert (b < 1e-12).any()


def test_depth_test_near_clipping_and_shared_edges():
    from oracle import raster as R
    mvp, _ = _mvp((0.0, 0.0, 0.0))
    # two quads facing the camera at z = 2 (nearer, smaller) and z = 4 (farther, larger), each two triangles sharing a diagonal; a third
    # triangle straddling the camera plane (one vertex behind the eye)
    def quad(z, s, base):
        return [[-s, -s, z], [s, -s, z], [s, s, z], [-s, s, z]], [[base, base + 1, base + 2], [base, base + 2, base + 3]]
    v0, t0 = quad(2.0, 0.5, 0)
    v1, t1 = quad(4.0, 3.0, 4)
    v = np.array(v0 + v1 + [[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, -0.1, -1.0]], np.float64)
    t = np.array(t0 + t1 + [[8, 9, 10]])
    c = 32
    r = R.rasterize(v, t, mvp, c)
    tri = r["tri_id"].reshape(6, c, c)[1]
    # every pixel of the front face is covered exactly once by the winner; the near quad (ids 1, 2) wins where it projects: |x|,|y| < 0.25
    jj, ii = np.meshgrid(np.arange(c), np.arange(c))
    x, y = (jj + 0.5) / c * 2 - 1, (ii + 0.5) / c * 2 - 1
    inner = (np.abs(x) < 0.25) & (np.abs(y) < 0.25)
    assert np.isin(tri[inner], (1, 2)).all()
    far_only = (np.abs(x) < 0.75) & (np.abs(y) < 0.75) & ~((np.abs(x) <= 0.25 + 2.0 / c) & (np.abs(y) <= 0.25 + 2.0 / c)) & (y > -0.2)
    assert np.isin(tri[far_only], (3, 4)).all()
    # the shared diagonal: no pixel lost, none claimed by both (each pixel has ONE id by construction; coverage is complete)
    assert (tri[(np.abs(x) < 0.7) & (np.abs(y) < 0.7) & (y > -0.2)] > 0).all()
    # the straddling triangle is visible only where it is in front of the near plane, and never wraps around through w < 0
    strad = tri == 5
    assert strad.any() and (y[strad] < 0).all()
    zw = r["zw"].reshape(6, c, c)[1]
    assert ((zw[tri > 0] >= -1) & (zw[tri > 0] <= 1)).all()


def test_raster_oracle_vs_ray_cast_restatement(golden):
    """independent algorithms, same G-buffer: edge-function rasterisation vs f64 brute-force ray casting on the 20 k-triangle room"""
    from oracle import oracle as O, raster as R, ref_torch as RT
    g = golden("irt_room.npz")
    mvp, _ = _mvp()
    rng = np.random.default_rng(1)
    cn = rng.normal(size=(3 * g["tris"].shape[0], 3)).astype(np.float32)
    c = 16
    a = R.gbuffer(g["verts"], g["tris"], g["tri_uvs"], mvp, c, corner_normals=cn, flip_v=True)
    osc = O.Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"])
    b = RT.gbuffer(osc, g["verts"], g["tris"], g["tri_uvs"], mvp, c, corner_normals=cn, flip_v=True)
    same = a["tri_id"] == b["tri_id"]
    assert same.mean() > 0.995, same.mean()
    assert (a["tri_id"] > 0).mean() > 0.99
    for k, tol in (("position", 1e-6), ("normal", 1e-5), ("uv", 1e-6), ("uv_da", 1e-4)):
        assert rel_l2(a[k][same], b[k][same]) < tol, (k, rel_l2(a[k][same], b[k][same]))
