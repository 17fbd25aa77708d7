"""Drop-in model classes for the reference's class-name plugin seam (utils/general.py:12-18 get_class):

  models.tracer_o3d_irt.TracerO3d      -> TracerO3d      (models/tracer_o3d_irt.py:35-180)
  models.mat_nvdiffrast.MaterialModel  -> MaterialModel  (models/mat_nvdiffrast.py:35-320)

Same constructor arguments, attributes the trainers touch (materials_a, materials_r, sample_l, texture) and forward()
contracts; all arithmetic runs in libtexir_hip.so.  Open3D/Embree, nvdiffrast, pyredner and cv2 are replaced by the BVH
scene handle, the ray-cast G-buffer + texture kernels and the loaders in io_formats.py.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from . import dist_util, gbuffer as GB, io_formats as IO
from .runlog import phases
from .cube2pano import Cube2Pano
from .scene import Scene, generate_dir, spec_render
from .texture import texture as tex_fetch, texture_batch as tex_fetch_batch

TINY_NUMBER = 1e-6


def get_mip_level(n):
    """utils/general.py:88-93"""
    count = 0
    while not (n & 1 or n == 1):
        n >>= 1
        count += 1
    return count


def rgb_to_intensity(t, dim=-1):
    """utils/general.py:95-112"""
    r, g, b = t.unbind(dim)
    return (0.29900 * r + 0.58700 * g + 0.11400 * b).unsqueeze(dim)


def hdr_scale(img, base=math.e):
    """utils/general.py:61-66"""
    return torch.log(img + 1) / math.log(base)


def _sibling(path_mesh, name):
    # the reference derives every asset path with str.replace("out1.obj", name)
    return path_mesh.replace("out1.obj", name) if "out1.obj" in path_mesh else os.path.join(os.path.dirname(path_mesh), name)


def _load_scene(conf, device):
    """mesh + radiance texture -> Scene (tracer_o3d_irt.py:75-89 / mat_nvdiffrast.py:87-101)"""
    path_mesh = conf.get_string("train.path_mesh_open3d")
    with phases.phase("load_obj", sync=False):
        obj = IO.load_obj(path_mesh)                                         # (parsed once per run: the dataset's load is served from the same cache)
        tri_uvs = IO.triangle_uvs_open3d(obj)
    with phases.phase("load_hdr_texture", sync=False):
        tex = IO.read_hdr(_sibling(path_mesh, "hdr_texture.hdr"))           # RGB
        expo = np.float32(2 ** conf.get_float("train.hdr_exposure"))
        tex = np.ascontiguousarray(tex[::-1])                                # cv2.flip(texture, 0)
        if expo != 1.0:
            tex *= expo
    with phases.phase("scene_build"):
        scene = Scene(obj["vertices"], obj["indices"], tri_uvs, tex, device=device)
        GB.set_corner_normals(scene, IO.corner_normals(obj))
    return scene, obj, torch.from_numpy(tex)


def seam_texels(index_texture):
    """texels the reference zeroes: `index_texture[:,:,0] + index_texture[:,:,1] + index_texture[:,:,2] == 0` evaluated in the image's own
    uint16 arithmetic (tracer_o3d_irt.py:137,176): a code triple whose sum wraps to 65536 counts as a seam there, and so it does here"""
    idx = np.asarray(index_texture)
    if idx.dtype == np.uint16:
        return ((idx[..., 0].astype(np.uint32) + idx[..., 1] + idx[..., 2]) & 0xFFFF) == 0
    if idx.dtype == np.uint8:
        return ((idx[..., 0].astype(np.uint32) + idx[..., 1] + idx[..., 2]) & 0xFF) == 0
    return idx.astype(np.int64).sum(-1) == 0


class TracerO3d(nn.Module):
    """Irradiance-texture model: forward() -> [H,W,3] irradiance (cuda tensor)."""

    def __init__(self, conf, ids, extrinsics, optim_cam=False, gt_irf=True):
        super().__init__()
        self.resolution = conf.get_list("train.env_res", default=[8, 16])
        self.path_traced_mesh = conf.get_string("train.path_mesh_open3d")
        self.pano_res = conf.get_list("train.pano_img_res", default=[1000, 2000])
        self.sample_l = conf.get_list("train.sample_light", default=[64, 64])
        self.sample_type = conf.get_list("models.render.sample_type", default=["uniform", "importance"])
        self.optim_cam, self.ids, self.extrinsics, self.conf = optim_cam, ids, extrinsics, conf
        self.cube_res = 256
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.scene, self.obj, self.texture = _load_scene(conf, self.device.index)
        # The reference resizes 0.png to 1024 x 1024 (tracer_o3d_irt.py:95) -- train.irt_res keeps that default size; `native` (or 0) opts out.
        # train.irt_resize (new optional key):
        #   nearest (default)  the flag the reference passes -- a true nearest-neighbour pick of the (row, column, panorama) codes;
        #   reference          what its call computes: cv2.INTER_NEAREST sits in the `dst` slot, so cv2 interpolates the uint16 codes with its default
        #                      INTER_LINEAR -- blended codes between texels of different panoramas included; identical to `nearest` when 0.png
        #                      already has the target size, bit-for-bit the reference's texture otherwise (imgops.resize_u16_as_cv2_default).
        with phases.phase("load_index_texture", sync=False):
            idx = IO.read_index_texture(_sibling(self.path_traced_mesh, "0.png"))
        res = conf.get("train.irt_res", 1024)
        how = str(conf.get("train.irt_resize", "nearest")).lower()
        if how not in ("nearest", "reference"):
            raise ValueError("train.irt_resize must be nearest or reference, got %r" % how)
        if res not in (None, 0, "0", "native"):
            res = int(res)
            if (res, res) != idx.shape[:2]:
                if how == "reference":
                    from .imgops import resize_u16_as_cv2_default
                    if idx.dtype != np.uint16:
                        raise ValueError("train.irt_resize = reference restates cv2's 16-bit path; 0.png is %s" % idx.dtype)
                    idx = resize_u16_as_cv2_default(np.ascontiguousarray(idx), (res, res))
                else:
                    ry = (np.arange(res) * idx.shape[0] // res)
                    rx = (np.arange(res) * idx.shape[1] // res)
                    idx = idx[ry][:, rx]
        self.index_texture = np.ascontiguousarray(idx)
        # optional exact texel G-buffer written by the synthetic generator (bypasses the panorama gather)
        self.texel_gbuffer_path = _sibling(self.path_traced_mesh, "texel_gbuffer.npz")
        self.use_texel_gbuffer = conf.get("train.texel_gbuffer", "auto")

    # -- tracer_o3d_irt.py:99-112 -----------------------------------------------------------------------------------
    def generate_positions(self):
        self.position_normal_list = []
        c2p = Cube2Pano(pano_width=1024, pano_height=512, cube_lenth=self.cube_res, cube_channel=6, is_cuda=True)
        for i in range(len(self.ids)):
            gb = GB.cast_gbuffer(self.scene, self.extrinsics[i], self.cube_res, flip_v=False)
            g = torch.cat([gb["position"], gb["normal"]], dim=-1)              # bg already (1,0,0,1,0,0)
            g = torch.cat([g[..., 0:3] + 1e-2 * g[..., 3:6], g[..., 3:6]], dim=-1)
            res = c2p.ToPano(g.permute(0, 3, 1, 2).reshape(1, -1, self.cube_res, self.cube_res))[0].permute(1, 2, 0)
            self.position_normal_list.append(res)

    # -- tracer_o3d_irt.py:115-142 (host numpy in the reference; same integer arithmetic on the device) ------------
    def calcute_position_normal_texture(self):
        idx = torch.from_numpy(self.index_texture.astype(np.int64)).to(self.device)       # [H,W,3] (row code, col code, pano id)
        H, W, _ = idx.shape
        pos = torch.zeros((H, W, 3), device=self.device)
        nrm = torch.zeros((H, W, 3), device=self.device)
        for hdr_id in torch.unique(idx[..., 2]).tolist():
            if hdr_id >= len(self.position_normal_list):
                raise ValueError("index texture references panorama %d but only %d views exist" % (hdr_id, len(self.position_normal_list)))
            sel = idx[..., 2] == hdr_id
            pano = self.position_normal_list[hdr_id]
            h, w, _ = pano.shape
            col = torch.clamp((idx[..., 1][sel].double() / 50000 * w).long(), 0, w - 1)
            row = torch.clamp((idx[..., 0][sel].double() / 50000 * h).long(), 0, h - 1)
            pos[sel] = pano[row, col, 0:3]
            nrm[sel] = pano[row, col, 3:6]
        seam = torch.from_numpy(seam_texels(self.index_texture)).to(self.device)
        pos[seam] = 0
        nrm[seam] = 0
        self.position_texture, self.normal_texture = pos, nrm

    def _load_texel_gbuffer(self):
        z = np.load(self.texel_gbuffer_path)
        self.position_texture = torch.from_numpy(z["position"]).to(self.device)
        self.normal_texture = torch.from_numpy(z["normal"]).to(self.device)

    # -- tracer_o3d_irt.py:145-180 ----------------------------------------------------------------------------------
    def forward(self):
        use_file = self.use_texel_gbuffer in (True, "file") or (self.use_texel_gbuffer == "auto" and os.path.exists(self.texel_gbuffer_path))
        # The per-texel shifts (33.5 M floats from the CPU generator at 4096^2: 0.2 s) are drawn on a helper thread WHILE the texel G-buffer is loaded / ray-cast:
        # nothing else touches the generator in between, so the stream is the reference's (one [nt,1,2] draw = its torch.rand(512,1,2) per 512-texel batch,
        # sample_util.py:102).  The texel count is the index texture's; if the G-buffer turns out to have another size the generator is put back and the draw redone.
        import threading
        nt0 = int(self.index_texture.shape[0]) * int(self.index_texture.shape[1])
        rng_before = torch.get_rng_state()
        early = {}
        th = threading.Thread(target=lambda: early.__setitem__("shift", torch.rand(nt0, 1, 2)), name="texir-irt-shifts", daemon=True)
        th.start()
        with phases.phase("texel_gbuffer"):
            if use_file:
                self._load_texel_gbuffer()
            else:
                self.generate_positions()
                self.calcute_position_normal_texture()
        print("Finish precomputing model!")
        H, W, _ = self.position_texture.shape
        with phases.phase("shifts_and_texel_list"):
            pos = self.position_texture.reshape(-1, 3).contiguous()
            nrm = self.normal_texture.reshape(-1, 3).contiguous()
            nt = pos.shape[0]
            # shifts: the reference draws torch.rand(512,1,2) per 512-texel batch from the CPU generator (sample_util.py:102);
            # one [nt,1,2] draw consumes the same stream in the same order
            th.join()
            if nt != nt0 or "shift" not in early:
                torch.set_rng_state(rng_before)
                early["shift"] = torch.rand(nt, 1, 2)
            shift = early.pop("shift").reshape(nt, 2).to(self.device, non_blocking=True)
            seam = torch.from_numpy(seam_texels(self.index_texture).reshape(-1)).to(self.device)
            ids = dist_util.morton_order(torch.nonzero(~seam)[:, 0].to(torch.int32), W)
            rank, world, _ = dist_util.world_info()
            ids_all = ids
            ids = dist_util.shard_block_cyclic(ids_all, rank, world)
            irr = torch.zeros((nt, 3), device=self.device)
        with phases.phase("irt_kernel"):
            self.scene.irt_generate(pos, nrm, shift, int(self.sample_l[0]), self.sample_type[0], texel_ids=ids, out=irr)
        with phases.phase("assemble_shards"):
            dist_util.assemble_shards(irr, ids_all)             # (one all_gather of the ranks' own texel values; no-op for one rank)
        self.ir_texture = irr.reshape(H, W, 3)
        return self.ir_texture

    def query_irf(self, points, directions, num_sample):
        """tracer_o3d_irt.py:240-269: points [b,n,3], directions [b,n,1,3] -> radiance [b,n,3]"""
        b, n, _ = points.shape
        return self.scene.trace_shade(points.reshape(-1, 3), directions.reshape(-1, 3)).reshape(b, n, 3)


class MaterialModel(nn.Module):
    """Material model: forward(mvp, id, cam_position, stage) -> dict(rgb, albedo, normal, position, empty_mask,
    roughness_womipmap, roughness)  (mat_nvdiffrast.py:107-190)."""

    def __init__(self, conf, ids, extrinsics, optim_cam=False, gt_irf=True, gt_irrt=True):
        super().__init__()
        self.resolution = conf.get_list("train.env_res", default=[8, 16])
        self.path_traced_mesh = conf.get_string("train.path_mesh_open3d")
        self.pano_res = conf.get_list("train.pano_img_res", default=[1000, 2000])
        self.cube_res = int(self.pano_res[1] / 4)
        self.sample_l = conf.get_list("train.sample_light", default=[64, 64])
        self.sample_type = conf.get_list("models.render.sample_type", default=["uniform", "importance"])
        self.optim_cam, self.ids, self.extrinsics, self.conf = optim_cam, ids, extrinsics, conf
        self.max_mip_level = get_mip_level(8192)
        # texture sizes: reference hard-codes 2048^2 x 3 and 4096^2 x 1 (mat_nvdiffrast.py:68-69); optional keys make them configurable
        ra = int(conf.get("train.albedo_res", 2048))
        rr = int(conf.get("train.roughness_res", 4096))
        self.materials_a = nn.Parameter(torch.ones((ra, ra, 3)) * 0.5, requires_grad=True)
        self.materials_r = nn.Parameter(torch.ones((rr, rr, 1)) * 0.1, requires_grad=True)
        self.gt_irrt = gt_irrt
        if gt_irrt:
            irt = IO.read_hdr(_sibling(self.path_traced_mesh, "irt.hdr"))        # NOT flipped (mat_nvdiffrast.py:73-76)
            self.irrt = nn.Parameter(torch.from_numpy(np.ascontiguousarray(irt)), requires_grad=False)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.scene, self.obj, tex = _load_scene(conf, self.device.index)
        self.texture = tex.permute(2, 0, 1).unsqueeze(0).float()                   # [1,3,H,W] like the reference attribute
        self._gb_cache = {}

    @classmethod
    def from_arrays(cls, scene, hdr_texture, irrt, conf, albedo_res=2048, roughness_res=4096):
        """build the model around an existing Scene and in-memory textures (synthetic benches / tests; no files).
        hdr_texture [Ht,Wt,3] in the tracer layout (flipped + exposed), irrt [H,W,3] in file orientation."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.pano_res = conf.get_list("train.pano_img_res", default=[1000, 2000])
        self.cube_res = int(self.pano_res[1] / 4)
        self.sample_l = conf.get_list("train.sample_light", default=[64, 64])
        self.sample_type = conf.get_list("models.render.sample_type", default=["uniform", "importance"])
        self.conf, self.max_mip_level, self.gt_irrt = conf, get_mip_level(8192), True
        self.device = scene.device
        self.materials_a = nn.Parameter(torch.ones((albedo_res, albedo_res, 3), device=self.device) * 0.5, requires_grad=True)
        self.materials_r = nn.Parameter(torch.ones((roughness_res, roughness_res, 1), device=self.device) * 0.1, requires_grad=True)
        self.irrt = nn.Parameter(torch.as_tensor(irrt, dtype=torch.float32).to(self.device).contiguous(), requires_grad=False)
        self.scene = scene
        self.texture = torch.as_tensor(hdr_texture, dtype=torch.float32).permute(2, 0, 1).unsqueeze(0)
        self._gb_cache = {}
        return self

    def _gbuffer(self, mvp, view_id):
        key = str(view_id)
        gb = self._gb_cache.get(key)
        if gb is None:
            gb = GB.cast_gbuffer(self.scene, mvp, self.cube_res, flip_v=True)     # pyredner-style uvs for the nvdiffrast-side textures
            self._gb_cache[key] = gb
        return gb

    def _fetch_materials(self, gb, womipmap=True):
        """the four dr.texture fetches of mat_nvdiffrast.py:131-139 (womipmap=False leaves the un-mipmapped roughness out: only the
        stage-1 loss reads it, train_material.py via loss.py:98-104)"""
        texc, texd = gb["uv"], gb["uv_da"]
        # (cache=gb: the view's fetch coordinates never change, so the backward is a gather over tap lists sorted once per view)
        roughness_womipmap = tex_fetch(self.materials_r, texc, texd, "linear", cache=gb) if womipmap else None
        # the two trilinear fetches as ONE node: one launch per kind of kernel over both textures, forward and backward (texture.texture_batch).  Made after
        # the un-mipmapped fetch, so that autograd runs its backward first: the trilinear fetch of the roughness texture decides between the sparse and the
        # dense level-0 form by `grad is None` (texture._bwd_prepare), as it did when it was a node of its own.
        # _fan_roughness (set by forward() for stage 2; an attribute, not an argument: subclasses override this method with its reference-shaped signature):
        # the mip-mapped roughness goes to the specular term AND to the loss; handed out as two tensors, its two gradients meet inside the gather instead of
        # in an autograd add launch.  `roughness` is then the pair (for render, for the loss).
        albedo, roughness = tex_fetch_batch([self.materials_a, self.materials_r], texc, texd, "linear-mipmap-linear", self.max_mip_level, cache=gb,
                                            fanout=[1, 2] if getattr(self, "_fan_roughness", False) else None)
        # the irradiance texture is frozen and the view's uvs are constant: fetch once per view
        irr = gb.get("_irr")
        if irr is None or gb.get("_irr_version") != self.irrt._version or self.irrt.requires_grad:
            irr = tex_fetch(self.irrt, texc, texd, "linear-mipmap-linear", self.max_mip_level)
            if not self.irrt.requires_grad:
                gb["_irr"], gb["_irr_version"] = irr.detach(), self.irrt._version
        return albedo, roughness_womipmap, roughness, irr

    @staticmethod
    def _view_consts(gb):
        """per-view constants kept with the cached G-buffer: the offset ray origins (mat_nvdiffrast.py:179,182) and the position the render reports"""
        if "_points" not in gb:
            gb["_points"] = gb["position"] + 1e-2 * gb["normal"]
            gb["_position_out"] = (gb["_points"] + 2e-2 * gb["normal"]).detach()
        return gb

    def forward(self, mvp, id, cam_position, stage=1):
        gb = self._gbuffer(mvp, id)
        pos, nrm, mask = gb["position"], gb["normal"], gb["mask"]
        self._view_consts(gb)
        # `lean_outputs` (set by the trainers' optimisation step): res["roughness_womipmap"] is None in the stages whose loss does not read it
        self._fan_roughness = bool(stage == 2 and torch.is_grad_enabled())
        try:
            albedo, roughness_womipmap, roughness, irr = self._fetch_materials(gb, womipmap=(stage == 1 or not getattr(self, "lean_outputs", False)))
        finally:
            self._fan_roughness = False
        roughness_loss = roughness
        if isinstance(roughness, tuple):
            roughness, roughness_loss = roughness
        cam_position = cam_position.to(self.device)
        if stage == -1:
            # light-source-only radiance texture (mat_nvdiffrast.py:141-150)
            src = self.texture[0].permute(1, 2, 0).to(self.device)
            inten = rgb_to_intensity(src * (2 ** -self.conf.get_float("train.hdr_exposure")))
            self.scene.set_texture(torch.where(inten >= 0.5, src, torch.zeros_like(src)).contiguous())
            try:
                res = self.render(nrm, torch.zeros_like(albedo), torch.ones_like(roughness) * 0.01, gb["_points"], cam_position, irr, gb["_position_out"])
            finally:
                self.scene.set_texture(src.contiguous())
        elif stage == 0:
            if "_position_out0" not in gb:                   # (a constant of the view: computed once, not twice per recorded step)
                gb["_position_out0"] = pos + 1e-1 * nrm
            res = {"rgb": irr * albedo / np.pi, "albedo": albedo, "normal": nrm, "position": gb["_position_out0"]}
        elif stage == 1:
            res = self.render(nrm, albedo.detach(), roughness_womipmap, gb["_points"], cam_position, irr, gb["_position_out"])
        elif stage == 2:
            res = self.render(nrm, albedo, roughness, gb["_points"], cam_position, irr, gb["_position_out"])
        else:
            raise ValueError("MaterialModel.forward: unknown stage %r" % (stage,))
        res.update({"empty_mask": mask, "roughness_womipmap": roughness_womipmap, "roughness": roughness_loss})
        return res

    def render(self, normal, albedo, roughness, points, cam_position, irr, position_out=None):
        """mat_nvdiffrast.py:201-249: fused GGX-importance sampling + trace + BRDF (texir_spec_forward/backward)"""
        face, h, w, _ = normal.shape
        P = face * h * w
        S = int(self.sample_l[1])
        if self.sample_type[1] != "importance":
            raise NotImplementedError("specular sample_type %r: the reference path uses 'importance'" % (self.sample_type[1],))
        static = getattr(self, "_static_shift", None)
        if static is not None:
            shift = static                      # hipGraph replay: the caller refreshes this buffer from the CPU generator each step
        else:
            shift = torch.rand(P, 1, 2).reshape(P, 2).to(self.device)           # sample_util.py:102 (CPU generator)
        rgb = spec_render(self.scene, normal.reshape(P, 3), albedo.reshape(P, 3), roughness.reshape(P), points.reshape(P, 3),
                          irr.reshape(P, 3), cam_position, shift, S)
        return {"rgb": rgb.reshape(face, h, w, 3), "albedo": albedo.reshape(face, h, w, 3), "normal": normal.reshape(face, h, w, 3).detach(),
                "position": (position_out if position_out is not None else (points + 2e-2 * normal).detach()).reshape(face, h, w, 3)}

    # -- mat_nvdiffrast.py:252-258 (call sites commented out at :166-169, 221-226; live in the evaluation model's relighting branch) ------
    def traced_diffuse(self, points, normal, albedo, shift=None):
        """the traced diffuse term the commented-out call sites compute: diffuse_reflectance(query_irf(points, l), l, n, albedo, type) / N
        with l = generate_dir(normal, N, mode=type) -- fused in texir_diffuse_irradiance (sample + trace + shade + reduce), times albedo/pi"""
        from .scene import diffuse_irradiance
        P = points.reshape(-1, 3).shape[0]
        if shift is None:
            shift = torch.rand(P, 1, 2).reshape(P, 2)                            # generate_dir's draw (sample_util.py:102)
        E = diffuse_irradiance(self.scene, points.reshape(P, 3), normal.reshape(P, 3), shift, int(self.sample_l[0]), self.sample_type[0])
        return E * albedo.reshape(P, 3) / np.pi

    def diffuse_reflectance(self, lighting, l, n, albedo, sample_type="uniform"):
        ndl = torch.clamp(torch.sum(n.unsqueeze(1) * l, dim=-1, keepdim=True), 0.0, 1.0)
        brdf = albedo.unsqueeze(1) / np.pi
        if sample_type == "cosine":
            return torch.sum(lighting * brdf * np.pi, dim=1)
        return torch.sum(lighting * brdf * ndl * 2 * np.pi, dim=1)

    def query_irf(self, points, directions, num_sample):
        b, n, _ = points.shape
        return self.scene.trace_shade(points.reshape(-1, 3), directions.reshape(-1, 3)).reshape(b, n, 3)
