"""Cube-map G-buffers by primary-ray casting (replaces dr.rasterize + dr.interpolate, models/mat_nvdiffrast.py:119-128)."""
import numpy as np
import torch

from . import _lib


def set_corner_normals(scene, corner_normals):
    """corner_normals [3T,3] = normals[indices] (pyredner.load_obj order, tracer_o3d_irt.py:61)"""
    a = np.ascontiguousarray(corner_normals, np.float32).reshape(-1, 3)
    if a.shape[0] != 3 * scene.n_tris:
        raise ValueError("corner_normals must be [3T,3]")
    _lib.check(_lib.lib().texir_scene_set_corner_normals(scene.h, _lib.ptr(a)))


def cast_gbuffer(scene, mvp, cube_res, flip_v=False):
    """mvp [6,4,4] (row-vector convention, datasets/dataset.py:464-465) -> dict of [6,c,c,k] tensors:
    position, normal, mask, uv (texc), uv_da (texd), tri_id"""
    dev = scene.device
    m = np.ascontiguousarray(mvp.detach().to("cpu", torch.float32).numpy().reshape(6, 4, 4))
    c = int(cube_res)
    P = 6 * c * c
    f = lambda k: torch.empty((P, k), device=dev, dtype=torch.float32)
    pos, nrm, mask, uv, uvda = f(3), f(3), f(1), f(2), f(4)
    tri = torch.empty((P,), device=dev, dtype=torch.int32)
    _lib.check(_lib.lib().texir_gbuffer_cast(scene.h, _lib.ptr(m), c, 1 if flip_v else 0, _lib.ptr(pos), _lib.ptr(nrm), _lib.ptr(mask),
                                             _lib.ptr(uv), _lib.ptr(uvda), _lib.ptr(tri), _lib.stream_ptr()))
    r = lambda t, k: t.reshape(6, c, c, k)
    return {"position": r(pos, 3), "normal": r(nrm, 3), "mask": r(mask, 1), "uv": r(uv, 2), "uv_da": r(uvda, 4), "tri_id": tri.reshape(6, c, c)}
