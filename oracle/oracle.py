"""ctypes binding of oracle/libtexir_oracle.so (plain-C restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY -- see the header of texir_oracle.c.  The product (texir_code_amd)
never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODES = {"uniform": 0, "cosine": 1, "importance": 2}


def build(force=False):
    so = os.path.join(_HERE, "libtexir_oracle.so")
    src = os.path.join(_HERE, "texir_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtexir_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        L.txo_scene_create.restype = vp
        L.txo_scene_create.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32]
        L.txo_scene_destroy.argtypes = [vp]
        L.txo_scene_info.argtypes = [vp, vp]
        L.txo_cast_rays_bruteforce.argtypes = [vp, vp, vp, i64, vp, vp, vp]
        L.txo_cast_rays_bvh.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
        L.txo_shade_hits.argtypes = [vp, vp, vp, vp, i64, vp]
        L.txo_trace_shade.argtypes = [vp, vp, vp, i64, i32, vp, vp]
        L.txo_hammersley.argtypes = [i32, vp]
        L.txo_generate_dir.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        L.txo_irt_generate.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp]
        L.txo_spec_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, C.c_float]
        L.txo_num_threads.restype = i32
        L.txo_set_num_threads.argtypes = [i32]
        L.txo_set_num_threads.restype = None
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def hammersley(n):
    out = np.empty((n, 2), np.float32)
    lib().txo_hammersley(n, _p(out))
    return out


def generate_dir(normals, n, mode, shift, roughness=None):
    normals = _f32(normals).reshape(-1, 3)
    b = normals.shape[0]
    shift = _f32(shift).reshape(b, 2)
    r = None if roughness is None else _f32(roughness).reshape(b)
    out = np.empty((b, n, 3), np.float32)
    lib().txo_generate_dir(_p(normals), b, n, MODES[mode], _p(r), _p(shift), _p(out))
    return out


class Scene:
    """verts [V,3] f32, tris [T,3] i32, tri_uvs [3T,2] (Open3D triangle_uvs order),
    hdr [Ht,Wt,3] f32 already flipped + exposure scaled (tracer_o3d_irt.py:77-81)."""

    def __init__(self, verts, tris, tri_uvs, hdr):
        self.verts = _f32(verts).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
        self.tri_uvs = _f32(tri_uvs).reshape(-1, 2)
        self.hdr = _f32(hdr)
        assert self.tri_uvs.shape[0] == 3 * self.tris.shape[0]
        Ht, Wt, c = self.hdr.shape
        assert c == 3
        self.h = lib().txo_scene_create(_p(self.verts), self.verts.shape[0], _p(self.tris), self.tris.shape[0],
                                        _p(self.tri_uvs), _p(self.hdr), Ht, Wt)

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            try:
                _LIB.txo_scene_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def info(self):
        out = np.zeros(3, np.int64)
        lib().txo_scene_info(self.h, _p(out))
        return {"n_nodes": int(out[0]), "max_depth": int(out[1]), "T": int(out[2])}

    def cast_rays(self, org, dir, tracer="bvh", counters=None):
        org = _f32(org).reshape(-1, 3)
        dir = _f32(dir).reshape(-1, 3)
        R = org.shape[0]
        t = np.empty(R, np.float32)
        pid = np.empty(R, np.uint32)
        uv = np.empty((R, 2), np.float32)
        if tracer == "brute":
            lib().txo_cast_rays_bruteforce(self.h, _p(org), _p(dir), R, _p(t), _p(pid), _p(uv))
        else:
            lib().txo_cast_rays_bvh(self.h, _p(org), _p(dir), R, _p(t), _p(pid), _p(uv), _p(counters))
        return t, pid, uv

    def shade_hits(self, t, pid, uv):
        t = _f32(t).reshape(-1)
        pid = np.ascontiguousarray(pid, np.uint32).reshape(-1)
        uv = _f32(uv).reshape(-1, 2)
        out = np.empty((t.shape[0], 3), np.float32)
        lib().txo_shade_hits(self.h, _p(t), _p(pid), _p(uv), t.shape[0], _p(out))
        return out

    def trace_shade(self, org, dir, tracer="bvh", counters=None):
        org = _f32(org).reshape(-1, 3)
        dir = _f32(dir).reshape(-1, 3)
        out = np.empty_like(org)
        lib().txo_trace_shade(self.h, _p(org), _p(dir), org.shape[0], 0 if tracer == "brute" else 1, _p(out), _p(counters))
        return out

    def irt_generate(self, pos, nrm, valid, shift, n, mode="uniform", tracer="bvh", counters=None, cosine_estimator=False):
        """cosine_estimator: (pi / n) sum L instead of (2 pi / n) sum L n.l -- the cosine branch of diffuse_reflectance"""
        pos = _f32(pos).reshape(-1, 3)
        nrm = _f32(nrm).reshape(-1, 3)
        Nt = pos.shape[0]
        shift = _f32(shift).reshape(Nt, 2)
        v = None if valid is None else np.ascontiguousarray(valid, np.uint8).reshape(Nt)
        out = np.zeros((Nt, 3), np.float32)
        lib().txo_irt_generate(self.h, _p(pos), _p(nrm), _p(v), _p(shift), Nt, n, MODES[mode] | (4 if cosine_estimator else 0),
                               0 if tracer == "brute" else 1, _p(out), _p(counters))
        return out

    def spec_forward(self, normal, albedo, rough, points, irr, cam, shift, S, tracer="bvh", return_ls=False, lighting=None, clamp_eps=1e-14):
        normal = _f32(normal).reshape(-1, 3)
        P = normal.shape[0]
        albedo = _f32(albedo).reshape(P, 3)
        rough = _f32(rough).reshape(P)
        points = _f32(points).reshape(P, 3)
        irr = _f32(irr).reshape(P, 3)
        cam = _f32(cam).reshape(3)
        shift = _f32(shift).reshape(P, 2)
        rgb = np.empty((P, 3), np.float32)
        ls = np.empty((P, S, 3), np.float32) if return_ls else None
        lin = None if lighting is None else _f32(lighting).reshape(P, S, 3)
        lib().txo_spec_forward(self.h, _p(normal), _p(albedo), _p(rough), _p(points), _p(irr), _p(cam), _p(shift),
                               P, S, 0 if tracer == "brute" else 1, _p(rgb), _p(ls), _p(lin), C.c_float(clamp_eps))
        return (rgb, ls) if return_ls else rgb


def new_counters():
    """[node records fetched, triangles tested, rays, hits]"""
    return np.zeros(4, np.uint64)


def num_threads():
    return lib().txo_num_threads()


def set_num_threads(n):
    """OpenMP threads of the following oracle calls (torch.set_num_threads() lowers the process-wide default)"""
    lib().txo_set_num_threads(int(n))
