"""Cube-map cameras exactly as the reference builds them (datasets/dataset.py:402-469 read_extrinsic, :544-549 projection):
six world->camera matrices derived from one camera-to-world extrinsic (front = +z), projected with fov 90, n=1e-4, f=100
and stored TRANSPOSED, so that clip = [x,y,z,1] @ mvp (row-vector convention, models/mat_nvdiffrast.py:119)."""
import math

import numpy as np
import torch


def projection(fov=90.0, n=1e-4, f=100.0):
    t = 1.0 / math.tan(fov * math.pi / 180.0 / 2.0)
    return np.array([[t, 0, 0, 0], [0, t, 0, 0], [0, 0, (f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, 1, 0]], np.float32)


def cube_mvps(c2w):
    """c2w [4,4] (columns: right, up, front, position) -> (mvp [6,4,4] float32, cam_position [3])"""
    E = torch.as_tensor(c2w, dtype=torch.float32)
    R, Fv = E[0:3, 0].clone(), E[0:3, 2].clone()
    Up = torch.linalg.cross(R, Fv)
    # (column 0, column 1, column 2) replacements per face; None keeps the extrinsic's own column
    table = [(Fv, None, -R), (None, None, None), (-Fv, None, R), (-R, None, -Fv), (None, Fv, Up), (None, -Fv, -Up)]
    w2c = []
    for c0, c1, c2 in table:
        M = E.clone()
        if c0 is not None:
            M[0:3, 0] = c0
        if c1 is not None:
            M[0:3, 1] = c1
        if c2 is not None:
            M[0:3, 2] = c2
        w2c.append(torch.inverse(M))
    w2c = torch.stack(w2c, 0)
    proj = torch.from_numpy(projection()).expand(6, 4, 4)
    mvp = torch.einsum("ijk,ikl->ijl", proj, w2c).transpose(1, 2).contiguous()
    return mvp.float(), E[0:3, 3].clone()


def grid_cameras(n_side=4, room=(8.0, 3.0, 6.0), height=1.5):
    """n_side x n_side panoramic cameras on a grid at y = height, identity orientation"""
    out = []
    for a in range(n_side):
        for b in range(n_side):
            E = np.eye(4, dtype=np.float32)
            E[0, 3] = room[0] * (a + 0.5) / n_side + 0.013 * (b + 1)
            E[1, 3] = height
            E[2, 3] = room[2] * (b + 0.5) / n_side + 0.017 * (a + 1)
            out.append(E)
    return out
