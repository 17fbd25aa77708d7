"""Composite oracle of ONE textured material-estimation step in plain torch autograd on the CPU.  TEST INFRASTRUCTURE ONLY
(see texir_oracle.c's header): only tests/ may import this module; the product never does.

What it restates, end to end and sharing no code with texir_code_amd (no HIP kernel, no tap list, no mip fold, no fused Adam):

    G-buffer        oracle/raster.py  (rasteriser rules of the dr.rasterize / dr.interpolate calls, models/mat_nvdiffrast.py:119-128)
    texture fetch   oracle/ref_torch.texture  (dr.texture: bilinear / trilinear-mip, wrap; models/mat_nvdiffrast.py:131-139) -- autograd
                    carries the gradient through the torch mip stack (avg_pool2d) down to the level-0 parameter
    stage 0/1/2     models/mat_nvdiffrast.py:151-190: Lambertian only / render on detached albedo + un-mipmapped roughness / full render
    render          models/mat_nvdiffrast.py:201-249 + specular_reflectance :260-279 + generate_dir(importance) utils/sample_util.py:63-146,
                    restated in `spec_render` below (pinned on CPU against tests/golden/spec_render.npz = the reference's own
                    values AND autograd gradients, tests/test_mat_step_oracle.py)
    lighting        the specular samples' radiance `Ls` comes from the C oracle's tracer (txo_spec_forward on its canonical BVH2 /
                    brute force): the reference detaches the directions before query_irf (mat_nvdiffrast.py:239), so Ls is a constant
                    of the step and the gradient flows through the sample weights only
    loss            models/loss.py:81-115 (RenderLoss) + :214-295 (SegLoss), L1 / L2, restated in `render_loss` below (pinned on CPU
                    against tests/golden/render_loss.npz = the reference's values and gradients)
    optimiser       torch.optim.Adam + the trainer's clamps (trainer/train_material.py:448-458, 519-525, 587-593)
"""
import math

import numpy as np
import torch

from . import raster as R
from . import ref_torch as RT

TINY = 1e-6          # utils/sample_util.py:25
TINY_TINY = 1e-14    # utils/sample_util.py:26


def _unit(x):
    # sample_util.py:86-87
    return x / (torch.norm(x, dim=-1, keepdim=True) + TINY)


def hammersley_points(S):
    """sample_util.py:28-41 as arrays: (i / S, bit-reversed i * 2^-32), float32"""
    i = np.arange(S, dtype=np.uint64)
    rev = np.zeros(S, np.uint64)
    for b in range(32):
        rev |= ((i >> np.uint64(b)) & np.uint64(1)) << np.uint64(31 - b)
    return np.stack([(i.astype(np.float64) / S), rev.astype(np.float64) * 2.3283064365386963e-10], -1).astype(np.float32)


def ggx_half_vectors(normal, rough, shift, S):
    """generate_dir(normal, S, None, 'importance', rough) with the Cranley-Patterson shift given: normal [P,3], rough [P,1] (autograd),
    shift [P,2] -> h [P,S,3]"""
    P = normal.shape[0]
    n = normal.unsqueeze(1).expand(P, S, 3)
    ex = torch.tensor([1.0, 0.0, 0.0])
    ey = torch.tensor([0.0, 1.0, 0.0])
    axis = torch.where(torch.abs(n[:, :, 0:1]) > 0.99, ey, ex)          # :84, decided on the RAW normal
    n = _unit(n)
    U = _unit(torch.linalg.cross(axis, n))
    V = _unit(torch.linalg.cross(n, U))
    s = torch.from_numpy(hammersley_points(S)).unsqueeze(0).repeat(P, 1, 1) + shift.reshape(P, 1, 2)
    s = torch.where(s > 1.0, s - 1.0, s)                                # :104-107
    s = torch.where(s < 0.0, s + 1.0, s)
    s = torch.clamp(s, TINY, 1 - TINY)
    a = (rough * rough).unsqueeze(1).expand(P, S, 1)
    phi = 2 * np.pi * s[:, :, 1:2] - np.pi
    ct = torch.sqrt((1.0 - s[:, :, 0:1]) / (1.0 + (a * a - 1) * s[:, :, 0:1]))
    ct = torch.clamp(ct, min=-1.0 + TINY, max=1.0 - TINY)
    st = torch.clamp(torch.sqrt(1.0 - ct * ct), min=-1.0 + TINY, max=1.0 - TINY)
    return V * (torch.sin(phi) * st) + n * ct + U * -(torch.cos(phi) * st)


def spec_render(normal, albedo, rough, points, irr, cam, shift, S, lighting, ceps=TINY_TINY):
    """render (mat_nvdiffrast.py:201-249) with specular_reflectance (:260-279) on GIVEN per-sample lighting [P,S,3] (what query_irf
    returned for the step's reflected directions): rgb [P,3].  Also returns the reflected directions l [P,S,3] (detached)."""
    dot = lambda a, b: torch.clamp(torch.sum(a * b, dim=-1, keepdim=True), 0.0, 1.0)
    v = torch.nn.functional.normalize(cam.unsqueeze(0) - points, eps=1e-4)
    diffuse = irr * albedo / np.pi
    h = ggx_half_vectors(normal, rough, shift, S)
    vb, nb = v.unsqueeze(1), normal.unsqueeze(1)
    vdh = dot(h, vb)
    l = 2 * vdh * h - vb
    ndl, ndh, ndv = dot(nb, l), dot(nb, h), dot(nb, vb)
    fres = 0.04 + 0.96 * torch.pow(2.0, (-5.55472 * vdh - 6.98316) * vdh)
    k = (rough.unsqueeze(1) + 1) * (rough.unsqueeze(1) + 1) / 8
    g = (ndl / torch.clamp(ndl * (1 - k) + k, min=ceps)) * (ndv / torch.clamp(ndv * (1 - k) + k, min=ceps))
    brdf = fres * g / torch.clamp(4 * ndl * ndv, min=ceps)
    spec = torch.sum(lighting * brdf * ndl * 4 * vdh / torch.clamp(ndh, ceps), dim=1) / S
    return diffuse + spec, l.detach()


def _seg_loss(img, img_womip, seg, fm, mode, room=None):
    """models/loss.py:214-295.  img [b,h,w,c]; seg / fm [C,b,h,w,1]; room [R,b,h,w,1]"""
    b, h, w, c = img.shape
    C = seg.shape[0]
    seg = seg.reshape(C, b, h * w, 1)
    fm = fm.reshape(C, b, h * w, 1)
    x = img.reshape(1, b, h * w, c).expand(C, -1, -1, -1)
    l1 = torch.nn.functional.l1_loss
    if mode == 0:
        mean = (x * seg).reshape(C, -1, c).sum(1, keepdim=True) / (seg.reshape(C, -1, 1).sum(1, keepdim=True) + TINY)
        return l1(x * seg, mean.unsqueeze(1) * seg)
    if mode == 1:
        xw = img_womip.reshape(1, b, h * w, c).expand(C, -1, -1, -1).detach()
        npx = fm.reshape(C, -1, 1).sum(1, keepdim=True)                  # [C,1,1]
        target = torch.ones((C, 1, c))
        for i in range(C):
            if npx[i, 0, 0].item() == 0:
                target[i] = 0
                continue
            sel = fm.reshape(C, -1, 1)[i, :, 0].bool()
            q = torch.quantile(xw.reshape(C, -1, c)[i][sel], 0.4, dim=0, keepdim=True)
            target[i] = torch.ones_like(q) * 0.8 if i == 43 else q
        wgt = (seg - fm) * (npx / (npx + TINY)).unsqueeze(1)
        return l1(x * wgt, target.unsqueeze(1) * wgt)
    Rn = room.shape[0]
    rm = room.reshape(Rn, 1, b, h * w, 1)
    both = seg.unsqueeze(0) * rm                                          # [R,C,b,hw,1]
    mean = (x.unsqueeze(0) * both).reshape(Rn, C, -1, c).sum(2, keepdim=True) / (both.reshape(Rn, C, -1, 1).sum(2, keepdim=True) + TINY)
    return l1(x.unsqueeze(0) * both, mean.unsqueeze(2) * both)


def render_loss(gt, preds, gt_mask, fm, seg, stage, room=None, loss_type="L1"):
    """models/loss.py:81-115 -> (loss, seg term)"""
    f = torch.nn.functional.l1_loss if loss_type == "L1" else torch.nn.functional.mse_loss
    hs = lambda x: torch.log(x + 1) / math.log(math.e)
    pred = preds["rgb"] * preds["empty_mask"]
    if stage == 0:
        direct = f(hs(pred * gt_mask), hs(gt * gt_mask))
        s = _seg_loss(preds["albedo"], None, seg, fm, 0) * 20
    elif stage == 1:
        direct = f(hs(gt.unsqueeze(0) * fm * seg), hs(pred.unsqueeze(0) * fm * seg)) * (pred.shape[1] * pred.shape[2])
        s = _seg_loss(preds["roughness"], preds["roughness_womipmap"], seg, fm, 1)
    else:
        direct = f(hs(gt.unsqueeze(0) * seg), hs(pred.unsqueeze(0) * seg))
        s = _seg_loss(preds["roughness"], None, seg, fm, 2, room) * 0.2
    return direct + s, s


class MaterialStepOracle:
    """The reference's MaterialModel + one trainer step, on the CPU.  Textures are torch leaf tensors (self.a [Ha,Wa,3], self.r [Hr,Wr,1])."""

    def __init__(self, oscene, verts, tris, tri_uvs, corner_normals, irrt, albedo0, rough0, cube_res, S, max_mip_level=13, tracer="bvh"):
        self.osc, self.verts, self.tris, self.tri_uvs, self.cn = oscene, verts, tris, tri_uvs, corner_normals
        self.c, self.S, self.max_mip, self.tracer = int(cube_res), int(S), int(max_mip_level), tracer
        self.irrt = torch.as_tensor(irrt, dtype=torch.float32)
        self.a = torch.as_tensor(albedo0, dtype=torch.float32).clone().requires_grad_(True)
        self.r = torch.as_tensor(rough0, dtype=torch.float32).clone().requires_grad_(True)
        self._gb = {}

    def gbuffer(self, key, mvp):
        gb = self._gb.get(key)
        if gb is None:
            g = R.gbuffer(self.verts, self.tris, self.tri_uvs, np.asarray(mvp, np.float64), self.c, corner_normals=self.cn, flip_v=True)
            t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], np.float32))
            gb = {k: t(k) for k in ("position", "normal", "mask", "uv", "uv_da")}
            gb["tri_id"] = g["tri_id"]
            gb["irr"] = RT.texture(self.irrt, gb["uv"], gb["uv_da"], "linear-mipmap-linear", self.max_mip)
            self._gb[key] = gb
        return gb

    def forward(self, key, mvp, cam, stage, shift):
        """-> preds dict of [6,c,c,k] tensors (autograd attached to self.a / self.r)"""
        gb, c, S = self.gbuffer(key, mvp), self.c, self.S
        uv, da = gb["uv"], gb["uv_da"]
        albedo = RT.texture(self.a, uv, da, "linear-mipmap-linear", self.max_mip)
        rough_wo = RT.texture(self.r, uv, da, "linear")
        rough = RT.texture(self.r, uv, da, "linear-mipmap-linear", self.max_mip)
        irr, nrm, pos = gb["irr"], gb["normal"], gb["position"]
        cam = torch.as_tensor(cam, dtype=torch.float32)
        if stage == 0:
            rgb = irr * albedo / np.pi
        else:
            alb_in, r_in = (albedo.detach(), rough_wo) if stage == 1 else (albedo, rough)
            pts = pos + 1e-2 * nrm
            # the step's lighting: trace the reflected directions of the CURRENT roughness with the C oracle (HIP-independent)
            _, Ls = self.osc.spec_forward(nrm.numpy(), alb_in.detach().numpy(), r_in.detach().numpy().reshape(-1), pts.numpy(), irr.numpy(),
                                          cam.numpy(), np.asarray(shift, np.float32), S, tracer=self.tracer, return_ls=True)
            rgb, _ = spec_render(nrm, alb_in, r_in, pts, irr, cam, torch.as_tensor(shift, dtype=torch.float32), S, torch.from_numpy(Ls))
        sh = lambda t, k: t.reshape(6, c, c, k)
        return {"rgb": sh(rgb, 3), "albedo": sh(albedo, 3), "roughness": sh(rough, 1), "roughness_womipmap": sh(rough_wo, 1),
                "empty_mask": sh(gb["mask"], 1)}

    def loss(self, key, mvp, cam, stage, shift, gt, gt_mask, fm, seg, room, loss_type="L1"):
        preds = self.forward(key, mvp, cam, stage, shift)
        return render_loss(gt, preds, gt_mask, fm, seg, stage, room, loss_type)[0], preds

    def grads(self, *a, **k):
        """(loss, d loss / d albedo texture, d loss / d roughness texture) at the current textures"""
        for p in (self.a, self.r):
            p.grad = None
        loss, _ = self.loss(*a, **k)
        loss.backward()
        z = lambda p: torch.zeros_like(p) if p.grad is None else p.grad.clone()
        return float(loss.detach()), z(self.a), z(self.r)

    def make_optimizer(self, stage, lr):
        """fresh Adam per stage over the parameters the stage trains (train_material.py:416-417, 472-479, 539-545)"""
        self.a.requires_grad_(stage in (0, 2))
        self.r.requires_grad_(stage in (1, 2))
        self.stage = stage
        self.opt = torch.optim.Adam([self.a, self.r], lr=lr)
        return self.opt

    def step(self, *a, **k):
        self.opt.zero_grad()
        loss, _ = self.loss(*a, **k)
        loss.backward()
        self.opt.step()
        with torch.no_grad():
            if self.stage == 2:
                self.a.clamp_(min=0.0)                    # :592
            self.r.clamp_(1e-2, 0.8)                      # :458 / :525 / :593
        return float(loss.detach())
