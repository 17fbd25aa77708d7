"""Rasteriser-semantics oracle for the G-buffer seam (TEST INFRASTRUCTURE ONLY, see texir_oracle.c's header).

The reference obtains its per-pixel G-buffer from nvdiffrast (models/mat_nvdiffrast.py:119-128, models/tracer_o3d_irt.py:99-112):
    rast, rast_db = dr.rasterize(ctx, pos_clip, tri, resolution)          # clip-space triangles -> (u, v, z/w, triangle id + 1) per pixel
    attr, attr_da = dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs='all')
nvdiffrast is a compiled third-party extension that is not part of /root/reference and cannot be installed here, so this file restates
the RASTERISATION rules it documents ("Modular Primitives for High-Performance Differentiable Rendering", Laine et al. 2020, and the
package documentation) -- and shares no code path with ray casting, which is what the product's texir_gbuffer_cast and the older
checker oracle/ref_torch.gbuffer both do:

  * triangles are processed in CLIP space; a pixel centre (col + 0.5, row + 0.5) maps to ndc ((col + .5) / c * 2 - 1, (row + .5) / c * 2 - 1),
    row 0 at clip y = -1;
  * coverage by edge functions in 2D homogeneous coordinates (Olano & Greer 1997): with M = [[x0,y0,w0],[x1,y1,w1],[x2,y2,w2]] the row
    vector e = (px, py, 1) M^-1 holds the three unnormalised perspective-correct weights; the pixel is covered iff all e_i share the sign
    of their sum and 1 / sum(e) = w at the pixel is positive (this is exact for triangles that cross the w = 0 plane: no explicit clipping
    of x, y); a pixel centre exactly ON an edge belongs to the triangle only if that edge is a top or a left edge (top-left fill rule);
  * depth test on z/w (LESS, range [-1, 1] = near / far clipping); among equal depths the triangle drawn first (lowest index) stays;
  * barycentrics are perspective-correct (b = e / sum e); rast_db holds their ANALYTIC screen-space derivatives per pixel;
  * interpolate: attr = sum b_i attr_i; attr_da = d attr / d(X, Y) in pixels through rast_db; background pixels carry zeros from
    nvdiffrast, over which the reference lays its own background values (mat_nvdiffrast.py:125: position = normal = (1, 0, 0)).
Parity status: UNPINNED against nvdiffrast itself (it cannot run here); this restatement is independent of the product's algorithm,
so a disagreement between ray casting and rasterisation (tie-breaks on shared edges, near-plane clipping, depth precision, analytic vs
finite-difference derivatives) shows up in tests/test_raster_oracle.py and tests/test_gpu_raster.py instead of cancelling out."""
import numpy as np


def rasterize(verts, tris, mvp, c, chunk=256):
    """verts [V,3], tris [T,3], mvp [6,4,4] (row-vector convention: clip = [x,y,z,1] @ mvp) -> dict of [6*c*c, ...] arrays:
    tri_id (int64, triangle + 1, 0 = background), bary [.,3] (perspective-correct weights of the triangle's corners 0, 1, 2),
    bary_dxy [.,3,2] (d bary / d(X, Y), pixels), zw (z/w of the visible surface)"""
    verts = np.asarray(verts, np.float64)
    tris = np.asarray(tris, np.int64)
    mvp = np.asarray(mvp, np.float64).reshape(6, 4, 4)
    T = tris.shape[0]
    P1 = c * c
    out_tri = np.zeros(6 * P1, np.int64)
    out_b = np.zeros((6 * P1, 3))
    out_db = np.zeros((6 * P1, 3, 2))
    out_z = np.full(6 * P1, np.inf)
    jj, ii = np.meshgrid(np.arange(c), np.arange(c))
    px = ((jj.reshape(-1) + 0.5) / c * 2 - 1)
    py = ((ii.reshape(-1) + 0.5) / c * 2 - 1)
    vh = np.concatenate([verts, np.ones((verts.shape[0], 1))], 1)
    for f in range(6):
        clip = vh @ mvp[f]                                        # [V,4]
        cv = clip[tris]                                           # [T,3,4]
        M = cv[:, :, [0, 1, 3]]                                   # rows (x_i, y_i, w_i)
        det = np.linalg.det(M)
        ok = np.abs(det) > 1e-300
        Minv = np.zeros_like(M)
        Minv[ok] = np.linalg.inv(M[ok])
        zc = cv[:, :, 2]                                          # clip z per corner
        # trivial reject: triangles entirely behind the eye
        ok &= (cv[:, :, 3] > 0).any(1)
        idx = np.flatnonzero(ok)
        if idx.size == 0:
            continue
        Mi, zi, cvi = Minv[idx], zc[idx], cv[idx]
        # projected corners (for the top-left rule; only used where all three w > 0)
        for p0 in range(0, P1, chunk):
            p1 = min(P1, p0 + chunk)
            x, y = px[p0:p1, None, None], py[p0:p1, None, None]
            e = x * Mi[None, :, 0, :] + y * Mi[None, :, 1, :] + Mi[None, :, 2, :]      # [p, t, 3]
            S = e.sum(-1)
            with np.errstate(divide="ignore", invalid="ignore"):
                b = e / S[..., None]
            inside = (S > 0) & (b >= 0).all(-1)
            # pixel centres exactly on an edge: keep only for top / left edges
            on_edge = inside & (b == 0).any(-1)
            if on_edge.any():
                pp, tt = np.nonzero(on_edge)
                for q, t in zip(pp, tt):
                    keep = True
                    w3 = cvi[t, :, 3]
                    if (w3 <= 0).any():
                        keep = False
                    else:
                        s = cvi[t, :, :2] / w3[:, None]                          # ndc corners
                        area = (s[1, 0] - s[0, 0]) * (s[2, 1] - s[0, 1]) - (s[2, 0] - s[0, 0]) * (s[1, 1] - s[0, 1])
                        for k in range(3):
                            if b[q, t, k] == 0:
                                a_, b_ = s[(k + 1) % 3], s[(k + 2) % 3]          # the edge opposite corner k
                                d = (b_ - a_) * (1.0 if area > 0 else -1.0)     # counter-clockwise orientation (y up)
                                # OpenGL-style window (y up): a left edge goes down (d.y < 0), a top edge is horizontal and goes left (d.x < 0)
                                top_left = (d[1] < 0) or (d[1] == 0 and d[0] < 0)
                                keep &= top_left
                    inside[q, t] = keep
            with np.errstate(invalid="ignore", over="ignore"):
                z = (np.where(inside[..., None], b, 0.0) * zi[None]).sum(-1) * S      # clip z at the pixel is sum b_i z_i; w at the pixel is 1 / S
            zw = np.where(inside, z, np.inf)                               # z / w = (sum b z) / (1 / S)
            zw = np.where((zw >= -1.0) & (zw <= 1.0), zw, np.inf)          # near / far clipping
            best = np.argmin(zw, axis=1)                                    # first minimum = lowest triangle index among equal depths
            zbest = zw[np.arange(p1 - p0), best]
            hit = np.isfinite(zbest)
            rows = np.arange(p0, p1)[hit] + f * P1
            tb = best[hit]
            out_tri[rows] = idx[tb] + 1
            bb = b[np.arange(p1 - p0)[hit], tb]
            out_b[rows] = bb
            out_z[rows] = zbest[hit]
            # analytic derivatives: e is affine in (px, py): de/dpx = Minv[0], de/dpy = Minv[1]; b = e / S
            Sb = S[np.arange(p1 - p0)[hit], tb][:, None]
            for a in range(2):
                de = Mi[tb, a, :]
                db = (de - bb * de.sum(-1, keepdims=True)) / Sb
                out_db[rows, :, a] = db * (2.0 / c)                        # d ndc / d pixel = 2 / c
    return {"tri_id": out_tri, "bary": out_b, "bary_dxy": out_db, "zw": out_z}


def gbuffer(verts, tris, tri_uvs, mvp, c, corner_normals=None, flip_v=False, chunk=256):
    """rasterize + interpolate as the reference's MaterialModel.forward uses them (mat_nvdiffrast.py:119-128): position, normal, uv with
    its pixel derivatives (du/dX, du/dY, dv/dX, dv/dY), mask, tri_id -- the layout of texir_gbuffer_cast / oracle.ref_torch.gbuffer"""
    verts = np.asarray(verts, np.float64)
    tris = np.asarray(tris, np.int64)
    r = rasterize(verts, tris, mvp, c, chunk)
    hit = r["tri_id"] > 0
    t = np.where(hit, r["tri_id"] - 1, 0)
    b = r["bary"]
    corners = verts[tris[t]]                                              # [P,3,3]
    pos = (b[..., None] * corners).sum(1)
    tuv = np.asarray(tri_uvs, np.float64).reshape(-1, 3, 2)[t]
    uv = (b[..., None] * tuv).sum(1)
    duv = np.einsum("pka,pkc->pca", r["bary_dxy"], tuv)                    # [P, (u,v), (X,Y)]
    da = np.stack([duv[:, 0, 0], duv[:, 0, 1], duv[:, 1, 0], duv[:, 1, 1]], -1)
    if corner_normals is not None:
        cn = np.asarray(corner_normals, np.float64).reshape(-1, 3, 3)[t]
        nrm = (b[..., None] * cn).sum(1)
    else:
        nrm = np.cross(corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 0])
        nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-300)
    if flip_v:
        uv[:, 1] = 1 - uv[:, 1]
        da[:, 2:] = -da[:, 2:]
    bg = np.array([1.0, 0.0, 0.0])
    pos = np.where(hit[:, None], pos, bg)
    nrm = np.where(hit[:, None], nrm, bg)
    uv = np.where(hit[:, None], uv, 0.0)
    da = np.where(hit[:, None], da, 0.0)
    return {"position": pos, "normal": nrm, "mask": hit.astype(np.float64), "uv": uv, "uv_da": da, "tri_id": r["tri_id"], "zw": r["zw"]}
