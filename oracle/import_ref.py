"""Stub-import harness for the reference's pure-torch arithmetic.  BUILD CONTAINER ONLY.

/root/reference does not exist on the GPU box and never travels (source, bytecode or otherwise).
This module is used only by oracle/make_golden.py, run by hand in the build container, to capture
small input/output vectors into tests/golden/.  It installs sys.modules stubs for the third-party
packages the reference imports but this image lacks (cv2, torchvision, open3d, nvdiffrast,
pyredner, pyhocon, pytorch_msssim, siren_pytorch, tinycudann, tensorboardX, GPUtil) and the three
shims the reference needs on numpy 2 / a CPU-only torch (np.math, np.int, Tensor.cuda).
"""
import math
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


class _Anything(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Anything(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return _Anything(self.__name__ + "()")


def install():
    for name in ["cv2", "torchvision", "torchvision.transforms", "open3d", "open3d.core", "nvdiffrast",
                 "nvdiffrast.torch", "pyredner", "pyhocon", "pytorch_msssim", "siren_pytorch", "tinycudann",
                 "tensorboardX", "GPUtil", "skimage", "imageio"]:
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["nvdiffrast"].torch = sys.modules["nvdiffrast.torch"]
    sys.modules["open3d"].core = sys.modules["open3d.core"]
    if not hasattr(np, "math"):
        np.math = math
    if not hasattr(np, "int"):
        np.int = int
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)


class FakeO3dTensor:
    def __init__(self, a):
        self.a = np.asarray(a)

    def numpy(self):
        return self.a


class FakeScene:
    """Stands in for o3d.t.geometry.RaycastingScene in TracerO3d.query_irf: cast_rays is answered by the
    oracle's double-precision brute-force tracer (Embree itself cannot run here: parity unpinned there)."""

    def __init__(self, oracle_scene, tracer="brute"):
        self.s = oracle_scene
        self.tracer = tracer

    def cast_rays(self, rays):
        r = np.asarray(rays.a if isinstance(rays, FakeO3dTensor) else rays, np.float32)
        shp = r.shape[:-1]
        t, pid, uv = self.s.cast_rays(r[..., 0:3].reshape(-1, 3), r[..., 3:6].reshape(-1, 3), tracer=self.tracer)
        return {"t_hit": FakeO3dTensor(t.reshape(shp)), "primitive_ids": FakeO3dTensor(pid.reshape(shp)),
                "primitive_uvs": FakeO3dTensor(uv.reshape(shp + (2,)))}


def patch_cv2_rodrigues():
    """utils/Pano2Cube.py:41-47 calls cv2.Rodrigues on axis-angle vectors: answer it with scipy's rotation-vector conversion
    (an implementation independent of texir_code_amd.pano2cube.rodrigues)"""
    from scipy.spatial.transform import Rotation
    sys.modules["cv2"].Rodrigues = lambda v: (Rotation.from_rotvec(np.asarray(v, np.float64)).as_matrix().astype(np.float32), None)


def patch_o3d_tensor():
    """o3d.core.Tensor(ndarray, dtype=...) -> FakeO3dTensor"""
    o3d = sys.modules["open3d"]
    o3d.core.Tensor = lambda a, dtype=None: FakeO3dTensor(a)
    o3d.core.Dtype = types.SimpleNamespace(Float32=None)
