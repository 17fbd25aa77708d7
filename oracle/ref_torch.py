"""torch-CPU / numpy restatements used ONLY as checkers (tests, smoke): nvdiffrast-style texture fetch, primary-ray
G-buffer with finite-difference pixel derivatives, Cube2Pano.  TEST INFRASTRUCTURE ONLY (see texir_oracle.c header)."""
import numpy as np
import torch
import torch.nn.functional as F


def mip_stack(tex, max_mip_level):
    """tex [H,W,C] -> list of levels by 2x2 box filter"""
    levels = [tex]
    t = tex
    while len(levels) <= max_mip_level and t.shape[0] % 2 == 0 and t.shape[1] % 2 == 0 and t.shape[0] >= 2 and t.shape[1] >= 2:
        t = F.avg_pool2d(t.permute(2, 0, 1).unsqueeze(0), 2).squeeze(0).permute(1, 2, 0)
        levels.append(t)
    return levels


def _bilinear_wrap(t, u, v):
    H, W, _ = t.shape
    u = u - torch.floor(u)
    v = v - torch.floor(v)
    x = u * W - 0.5
    y = v * H - 0.5
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    fx = (x - x0).unsqueeze(-1)
    fy = (y - y0).unsqueeze(-1)
    x0 = x0.long() % W
    y0 = y0.long() % H
    x1 = (x0 + 1) % W
    y1 = (y0 + 1) % H
    return t[y0, x0] * (1 - fx) * (1 - fy) + t[y0, x1] * fx * (1 - fy) + t[y1, x0] * (1 - fx) * fy + t[y1, x1] * fx * fy


def texture(tex, uv, uv_da=None, filter_mode="linear", max_mip_level=13):
    """differentiable (autograd) restatement of nvdiffrast texture(): tex [H,W,C], uv [P,2], uv_da [P,4]"""
    u, v = uv[:, 0], uv[:, 1]
    if filter_mode == "linear":
        return _bilinear_wrap(tex, u, v)
    levels = mip_stack(tex, max_mip_level)
    H, W, _ = tex.shape
    dsdx, dsdy, dtdx, dtdy = uv_da[:, 0] * W, uv_da[:, 1] * W, uv_da[:, 2] * H, uv_da[:, 3] * H
    A = dsdx * dsdx + dtdx * dtdx
    B = dsdy * dsdy + dtdy * dtdy
    Cc = dsdx * dsdy + dtdx * dtdy
    major = 0.5 * (A + B) + torch.sqrt(0.25 * (A - B) ** 2 + Cc * Cc)
    lv = torch.clamp(0.5 * torch.log2(major), 0, len(levels) - 1)
    lv = torch.nan_to_num(lv, nan=0.0)
    l0 = torch.floor(lv).long()
    l1 = torch.clamp(l0 + 1, max=len(levels) - 1)
    f = (lv - l0.float()).unsqueeze(-1)
    out = torch.zeros(uv.shape[0], tex.shape[2], dtype=tex.dtype)
    for l, t in enumerate(levels):
        s = _bilinear_wrap(t, u, v)
        out = out + s * ((l0 == l).float().unsqueeze(-1) * (1 - f) + ((l1 == l) & (l1 != l0)).float().unsqueeze(-1) * f)
    return out


def gbuffer(oscene, verts, tris, tri_uvs, mvp, c, corner_normals=None, flip_v=False):
    """numpy float64 restatement of the primary-ray G-buffer: unproject the near and far points of each pixel through
    inverse(mvp), brute-force closest hit, pixel derivatives of uv by central differences on the hit triangle's plane."""
    minv = np.linalg.inv(np.asarray(mvp, np.float64))
    jj, ii = np.meshgrid(np.arange(c), np.arange(c))

    def rays(dx=0.0, dy=0.0):
        x = (jj + 0.5 + dx) / c * 2 - 1
        y = (ii + 0.5 + dy) / c * 2 - 1
        O, D = [], []
        for f in range(6):
            a = np.stack([x, y, -np.ones_like(x), np.ones_like(x)], -1) @ minv[f]
            b = np.stack([x, y, np.ones_like(x), np.ones_like(x)], -1) @ minv[f]
            a = a[..., :3] / a[..., 3:]
            b = b[..., :3] / b[..., 3:]
            O.append(a)
            D.append(b - a)
        return np.stack(O).reshape(-1, 3), np.stack(D).reshape(-1, 3)

    O, D = rays()
    t, pid, buv = oscene.cast_rays(O.astype(np.float32), D.astype(np.float32), tracer="brute")
    hit = np.isfinite(t)
    P = O.shape[0]
    verts = np.asarray(verts, np.float64)
    tri_uvs = np.asarray(tri_uvs, np.float64).reshape(-1, 3, 2)
    pidc = np.where(hit, pid, 0).astype(np.int64)
    v0, v1, v2 = verts[tris[pidc, 0]], verts[tris[pidc, 1]], verts[tris[pidc, 2]]

    def bary(Oe, De):
        e1, e2 = v1 - v0, v2 - v0
        pv = np.cross(De, e2)
        det = (e1 * pv).sum(-1)
        tv = Oe - v0
        u = (tv * pv).sum(-1) / det
        q = np.cross(tv, e1)
        v = (De * q).sum(-1) / det
        return u, v

    u, v = bary(O, D)
    w = 1 - u - v
    pos = v0 * w[:, None] + v1 * u[:, None] + v2 * v[:, None]
    uv = tri_uvs[pidc, 0] * w[:, None] + tri_uvs[pidc, 1] * u[:, None] + tri_uvs[pidc, 2] * v[:, None]
    h = 1e-3
    da = np.zeros((P, 4))
    for k, (dx, dy) in enumerate([(h, 0), (0, h)]):
        up, vp = bary(*rays(dx, dy))
        um, vm = bary(*rays(-dx, -dy))
        du, dv = (up - um) / (2 * h), (vp - vm) / (2 * h)
        duv = (tri_uvs[pidc, 1] - tri_uvs[pidc, 0]) * du[:, None] + (tri_uvs[pidc, 2] - tri_uvs[pidc, 0]) * dv[:, None]
        da[:, 0 + k] = duv[:, 0]
        da[:, 2 + k] = duv[:, 1]
    if corner_normals is not None:
        cn = np.asarray(corner_normals, np.float64).reshape(-1, 3, 3)
        nrm = cn[pidc, 0] * w[:, None] + cn[pidc, 1] * u[:, None] + cn[pidc, 2] * v[:, None]
    else:
        nrm = np.cross(v1 - v0, v2 - v0)
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    if flip_v:
        uv[:, 1] = 1 - uv[:, 1]
        da[:, 2:] = -da[:, 2:]
    bg = np.array([1.0, 0.0, 0.0])
    pos = np.where(hit[:, None], pos, bg)
    nrm = np.where(hit[:, None], nrm, bg)
    uv = np.where(hit[:, None], uv, 0.0)
    da = np.where(hit[:, None], da, 0.0)
    return {"position": pos, "normal": nrm, "mask": hit.astype(np.float64), "uv": uv, "uv_da": da, "tri_id": np.where(hit, pid.astype(np.int64) + 1, 0)}
