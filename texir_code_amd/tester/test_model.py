"""Drop-in for models.test_nvdiffrast.MaterialModel (models/test_nvdiffrast.py:36-367): the evaluation-side material model the
tester runners instantiate (tester/test_editing.py:110, test_novel.py:109, test_relighting.py:109, test_error.py:109).

Differences from the training model (models.py MaterialModel) that the reference has and this class keeps:
  * conf keys come from the `test{}` block; the material textures are the newest `mat_albedo-1_<iter>.hdr` /
    `mat_roughness-1_<iter>.hdr` pair of a training run's plots directory (sort_res, :112-124), frozen;
  * a class-id texture `0_seg_gray.png` is fetched bilinearly and drives the material edits (:160-238);
  * relighting=True recolours the light sources of the radiance texture (:98-101) and TRACES the diffuse term
    (diffuse_reflectance live, :268-274) instead of reading the pre-computed irradiance texture;
  * specular_reflectance floors its denominators at TINY_NUMBER = 1e-6 (:320-333), not 1e-14.
All arithmetic is the same HIP path: texir_gbuffer_cast / tex fetch / texir_diffuse_irradiance / texir_spec_forward."""
import functools
import glob
import os

import numpy as np
import torch
from torch import nn

from .. import gbuffer as GB, io_formats as IO
from ..models import _sibling, get_mip_level, rgb_to_intensity
from ..scene import Scene, diffuse_irradiance, spec_render
from ..texture import texture as tex_fetch

TINY_NUMBER = 1e-6
RELIGHT_COLOUR = (2.14, 1.38, 0.2)              # test_nvdiffrast.py:101


def sort_res(checkpoint_material):
    """newest material plot of a run: the file with the largest trailing iteration number (test_nvdiffrast.py:112-124)"""
    def compare(A, B):
        a, b = int(os.path.splitext(A)[0].split("_")[-1]), int(os.path.splitext(B)[0].split("_")[-1])
        return -1 if a > b else 1
    paths = glob.glob("{}/mat_albedo-1_*".format(checkpoint_material))
    if not paths:
        raise FileNotFoundError("no mat_albedo-1_*.hdr under %s (run --trainstage Mat first)" % checkpoint_material)
    paths.sort(key=functools.cmp_to_key(compare))
    return paths[0]


class MaterialModel(nn.Module):
    def __init__(self, conf, cam_position_list, checkpoint_material, gt_irf=True, relighting=False, gt_irrt=True):
        super().__init__()
        self.path_traced_mesh = conf.get_string("test.path_mesh_open3d")
        self.pano_res = conf.get_list("test.pano_img_res", default=[1000, 2000])
        self.cube_res = int(self.pano_res[1] / 4)
        self.sample_l = conf.get_list("test.sample_light", default=[64, 64])
        self.sample_type = conf.get_list("models.render.sample_type", default=["uniform", "importance"])
        self.conf, self.checkpoint_material, self.relighting, self.gt_irrt = conf, checkpoint_material, relighting, gt_irrt
        self.device = torch.device("cuda", torch.cuda.current_device())
        albedo_path = sort_res(checkpoint_material)
        a = IO.read_hdr(albedo_path)                                                    # RGB (cv2.imread(-1)[:, :, ::-1])
        r = IO.read_hdr(albedo_path.replace("albedo", "roughness"))[:, :, 0:1]          # cv2's channel 0 of a grey .hdr
        self.materials_a = nn.Parameter(torch.from_numpy(np.ascontiguousarray(a, np.float32)), requires_grad=False)
        self.materials_r = nn.Parameter(torch.from_numpy(np.ascontiguousarray(r, np.float32)), requires_grad=False)
        irt = IO.read_hdr(_sibling(self.path_traced_mesh, "irt.hdr"))
        self.irrt = nn.Parameter(torch.from_numpy(np.ascontiguousarray(irt, np.float32)), requires_grad=False)
        seg = IO.read_png(_sibling(self.path_traced_mesh, "0_seg_gray.png"))
        seg = seg[..., 0:1] if seg.ndim == 3 else seg[..., None]
        self.texture_seg = nn.Parameter(torch.from_numpy(np.ascontiguousarray(seg, np.float32)), requires_grad=False)
        self.max_mip_level = get_mip_level(irt.shape[0])                               # :86 (`texture` is the irradiance texture there)
        # tracing scene (:92-110)
        exposure = conf.get_float("test.hdr_exposure")
        obj = IO.load_obj(self.path_traced_mesh)
        tex = IO.read_hdr(_sibling(self.path_traced_mesh, "hdr_texture.hdr"))
        tex = np.ascontiguousarray(tex[::-1]) * np.float32(2 ** exposure)
        if relighting:
            inten = 0.299 * tex[..., 0] + 0.587 * tex[..., 1] + 0.114 * tex[..., 2]
            inten = inten * np.float32(2 ** -exposure)
            tex = np.where(inten[..., None] > 0.5, np.array(RELIGHT_COLOUR, np.float32) * np.float32(2 ** exposure), tex).astype(np.float32)
        self.scene = Scene(obj["vertices"], obj["indices"], IO.triangle_uvs_open3d(obj), tex, device=self.device.index)
        GB.set_corner_normals(self.scene, IO.corner_normals(obj))
        self.texture = torch.from_numpy(tex).permute(2, 0, 1).unsqueeze(0).float()
        self._gb_cache = {}

    def _gbuffer(self, mvp, view_id):
        key = (str(view_id), hash(mvp.cpu().numpy().tobytes()))            # novel views all come with id 0 (tester/test_novel.py:181)
        gb = self._gb_cache.get(key)
        if gb is None:
            gb = GB.cast_gbuffer(self.scene, mvp, self.cube_res, flip_v=True)
            self._gb_cache[key] = gb
        return gb

    def forward(self, mvp, id, cam_position, editing=True, albedo_floor=None, albedo_wall=None, roughness_floor=None):
        """test_nvdiffrast.py:127-246.  (The runners also pass a stage number or False in the `editing` slot: any truthy value edits.)"""
        gb = self._gbuffer(mvp, id)
        texc, texd = gb["uv"], gb["uv_da"]
        albedo = tex_fetch(self.materials_a, texc, texd, "linear-mipmap-linear", self.max_mip_level)
        seg = tex_fetch(self.texture_seg, texc, texd, "linear")
        roughness = tex_fetch(self.materials_r, texc, texd, "linear-mipmap-linear", self.max_mip_level)
        irr = tex_fetch(self.irrt, texc, texd, "linear-mipmap-linear", self.max_mip_level)
        if editing:
            dev = albedo.device
            if albedo_floor is not None:
                albedo = torch.where(seg == 46.0, torch.as_tensor(albedo_floor, dtype=torch.float32, device=dev), albedo)
                albedo = torch.where(seg == 45.0, torch.as_tensor(albedo_wall, dtype=torch.float32, device=dev), albedo)
            if roughness_floor is not None:
                roughness = torch.where(seg == 46.0, torch.as_tensor(roughness_floor, dtype=torch.float32, device=dev), roughness)
        nrm, pos = gb["normal"], gb["position"]
        res = self.render(nrm, albedo, roughness, pos + 1e-2 * nrm, cam_position.to(self.device), irr)
        res.update({"roughness": roughness, "empty_mask": gb["mask"]})
        return res

    def render(self, normal, albedo, roughness, points, cam_position, irr):
        """test_nvdiffrast.py:256-304"""
        face, h, w, _ = normal.shape
        P = face * h * w
        normal, albedo, roughness, points, irr = (normal.reshape(P, 3), albedo.reshape(P, 3), roughness.reshape(P), points.reshape(P, 3), irr.reshape(P, 3))
        if self.sample_type[1] != "importance":
            raise NotImplementedError("specular sample_type %r: the reference path uses 'importance'" % (self.sample_type[1],))
        if self.relighting:
            # :268-274 -- diffuse_reflectance(query_irf(points, generate_dir(normal, N0, type)), ...) / N0 == E * albedo / pi
            shift_d = torch.rand(P, 1, 2).reshape(P, 2).to(self.device)                   # generate_dir's draw (sample_util.py:102)
            irr = diffuse_irradiance(self.scene, points, normal, shift_d, int(self.sample_l[0]), self.sample_type[0])
        shift_s = torch.rand(P, 1, 2).reshape(P, 2).to(self.device)
        rgb = spec_render(self.scene, normal, albedo, roughness, points, irr, cam_position, shift_s, int(self.sample_l[1]), clamp_eps=TINY_NUMBER)
        return {"rgb": rgb.reshape(face, h, w, 3), "albedo": albedo.reshape(face, h, w, 3), "normal": normal.reshape(face, h, w, 3).detach(),
                "position": (points + 2e-2 * normal).reshape(face, h, w, 3).detach()}

    def query_irf(self, points, directions, num_sample):
        b, n, _ = points.shape
        return self.scene.trace_shade(points.reshape(-1, 3), directions.reshape(-1, 3)).reshape(b, n, 3)
