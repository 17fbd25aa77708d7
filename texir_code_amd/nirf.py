"""Neural irradiance field (NIrF) slice: SURVEY.md 8(f) row 4.

  models.tracer_o3d_irrf.TracerO3d -> TracerO3dIrrF   (models/tracer_o3d_irrf.py:30-136)
  models.incidentNet.MatNetwork    -> MatNetwork      (models/incidentNet.py:103-142; PE from models/embedder.py:6-53)
  models.loss.IRFLoss              -> IRFLoss         (models/loss.py:28-52)

The ground truth the MLP is fitted to is the IrT estimator evaluated at random mesh points (same trace + integrate as
models/tracer_o3d_irt.py:156-173, uniform hemisphere sampling, one Cranley-Patterson shift per point), so it runs on
the IrT kernel of libtexir_hip.so; the MLP itself stays stock PyTorch-ROCm (hipBLASLt GEMMs), as in the reference.
"""
import math

import numpy as np
import torch
from torch import nn

from . import io_formats as IO
from .scene import Scene


class Embedder:
    """NeRF positional encoding (models/embedder.py:6-38): [x, sin(2^k x), cos(2^k x)] for k = 0..multires-1"""

    def __init__(self, multires, input_dims=3, include_input=True):
        self.include_input = include_input
        self.freq_bands = 2.0 ** torch.linspace(0.0, multires - 1, multires)
        self.out_dim = input_dims * ((1 if include_input else 0) + 2 * multires)

    def embed(self, x):
        out = [x] if self.include_input else []
        for f in self.freq_bands:
            f = f.to(x.device)
            out.append(torch.sin(x * f))
            out.append(torch.cos(x * f))
        return torch.cat(out, -1)


def get_embedder(multires):
    """models/embedder.py:41-53"""
    eo = Embedder(multires)
    return eo.embed, eo.out_dim


class MatNetwork(nn.Module):
    """models/incidentNet.py:103-142: PE -> (Linear + LeakyReLU(0.01)) x len(dims) -> Linear, kaiming-uniform(relu) weights, zero bias.
    state_dict keys equal the reference's (`vis_layer.{2k}.weight/bias`), so its checkpoints load."""

    def __init__(self, points_multires=10, p_input_dim=3, p_out_dim=4, dims=(128, 128, 128, 128), AABB=None):
        super().__init__()
        self.p_embed_fn = None
        if points_multires > 0:
            self.p_embed_fn, p_input_dim = get_embedder(points_multires)
        self.actv_fn = nn.LeakyReLU(0.01, inplace=False)
        layers, dim = [], p_input_dim
        for d in dims:
            layers += [nn.Linear(dim, d), self.actv_fn]
            dim = d
        layers.append(nn.Linear(dim, p_out_dim))
        self.vis_layer = nn.Sequential(*layers)
        self.vis_layer.apply(self._init_weights)

    def forward(self, points):
        if self.p_embed_fn is not None:
            points = self.p_embed_fn(points)
        return self.vis_layer(points)

    @staticmethod
    def _init_weights(m):
        if type(m) == nn.Linear:
            nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
            if hasattr(m.bias, "data"):
                m.bias.data.fill_(0.0)


class IRFLoss(nn.Module):
    """models/loss.py:28-52: L1 / L2 between ln(1 + gt) (utils/general.py:61-66) and the prediction"""

    def __init__(self, loss_type="L1"):
        super().__init__()
        if loss_type == "L1":
            print("Using L1 loss for comparing radiance!")
            self.rgb_loss = nn.L1Loss(reduction="mean")
        elif loss_type == "L2":
            print("Using L2 loss for comparing radiance!")
            self.rgb_loss = nn.MSELoss(reduction="mean")
        else:
            raise Exception("Unknown loss_type!")

    def forward(self, res):
        return self.rgb_loss(torch.log(res["gt"] + 1) / math.log(math.e), res["pred"])


def hdr_recover(img, base=math.e):
    """utils/general.py:68-73"""
    return torch.pow(base, img) - 1


class TracerO3dIrrF(nn.Module):
    """models/tracer_o3d_irrf.py:30-136.  forward(points [b,3], normals [b,3], resolution [h,w], isnot_first_val) ->
    {'gt': [b,3] traced irradiance (absent when isnot_first_val), 'pred': [b,3], 'pred_jit': [b,3]}"""

    def __init__(self, conf, AABB=None, is_hdr_texture=False, scene=None):
        super().__init__()
        self.ir_radiance_network = MatNetwork(**conf.get_config("models.irrf_network"), AABB=AABB)
        self.path_traced_mesh = conf.get_string("train.path_mesh_open3d")
        self.std_jit = conf.get_float("train.std_jit")
        self.device = torch.device("cuda", torch.cuda.current_device())
        if scene is not None:
            self.scene = scene
            return
        obj = IO.load_obj(self.path_traced_mesh)
        if is_hdr_texture:
            # :52-57  hdr_texture.hdr, vertical flip, 2^exposure
            tex = IO.read_hdr(self.path_traced_mesh.replace("out1.obj", "hdr_texture.hdr"))
            tex = np.ascontiguousarray(tex[::-1]) * np.float32(2 ** conf.get_float("train.hdr_exposure"))
        else:
            # :58-60  the mesh's LDR texture map, (x/255)^2.2 (Open3D keeps image rows as stored: no flip)
            path_tex = IO.obj_texture_path(self.path_traced_mesh)
            if path_tex is None:
                raise ValueError("%s names no map_Kd texture; set train.is_hdr_texture = True" % self.path_traced_mesh)
            tex = IO.read_ldr(path_tex)
            tex = (tex.astype(np.float32) / np.float32(np.iinfo(tex.dtype).max)) ** np.float32(2.2)
        self.scene = Scene(obj["vertices"], obj["indices"], IO.triangle_uvs_open3d(obj), tex, device=self.device.index)

    def trace_gt(self, points, normals, resolution, shift=None):
        """:86-122  E = (2 pi / N) sum_i L(p, d_i) clamp(n . d_i, 0, 1), N = h*w uniform-hemisphere Hammersley samples"""
        b = points.shape[0]
        if shift is None:
            shift = torch.rand(b, 1, 1, 2)                       # :193 CPU generator, then .cuda()
        return self.scene.irt_generate(points.contiguous(), normals.contiguous(), shift.reshape(b, 2).to(points.device),
                                       int(resolution[0]) * int(resolution[1]), "uniform")

    def forward(self, points, normals, resolution, isnot_first_val=False):
        res = {}
        if not isnot_first_val:
            res["gt"] = self.trace_gt(points, normals, resolution)
        res["pred"] = self.ir_radiance_network(points)
        jit = torch.normal(mean=0.0, std=self.std_jit, size=points.shape, device=points.device)
        res["pred_jit"] = self.ir_radiance_network(points + jit)
        return res
