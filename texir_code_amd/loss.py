"""Drop-in for models.loss.RenderLoss (+ SegLoss, hdr_scale) -- models/loss.py:55-115, 214-295, utils/general.py:61-66.

Same constructor and forward signature/return tuple as the reference class; the arithmetic runs in the fused HIP
kernels behind texir_loss_forward.  The reference's [C,6,h,w,1] one-hot mask tensors are accepted as-is and
compacted (once per mask tensor, cached) to 1-byte class / highlight / room ids.
"""
import torch
from torch import nn

from . import _lib

NO_CLASS = 255
_LOSS_TYPES = {"L1": 0, "L2": 1}
# the other image terms of models/loss.py:65-73: evaluated with stock torch ops on top of the fused SegLoss (kernel loss_type 2)
_VARIANTS = ("psnr", "ssim", "msssim")


def compact_masks(seg_mask, floor_max_mask=None, room_seg_mask=None):
    """[C,...,1] one-hot float masks -> (seg_id u8 [P], hl u8 [P] | None, room_id u8 [P] | None, C, R).
    Raises if the masks are not what trainer/train_material.py:255-296 constructs (one-hot seg / room masks,
    floor_max_mask a subset of seg_mask): the compact kernels would silently compute something else."""
    C = seg_mask.shape[0]
    s = seg_mask.reshape(C, -1)
    cnt = s.sum(0)
    if not bool(((s == 0) | (s == 1)).all()) or float(cnt.max()) > 1:
        raise ValueError("seg_mask must be a one-hot {0,1} mask over the class dimension")
    seg_id = torch.where(cnt > 0, s.argmax(0), torch.full_like(cnt, NO_CLASS, dtype=torch.long)).to(torch.uint8)
    hl = None
    if floor_max_mask is not None:
        f = floor_max_mask.reshape(C, -1)
        if not bool(((f == 0) | (f == 1)).all()) or bool((f * (1 - s)).any()):
            raise ValueError("floor_max_mask must be a {0,1} subset of seg_mask")
        hl = (f.sum(0) > 0).to(torch.uint8)
    room_id, R = None, 0
    if room_seg_mask is not None:
        R = room_seg_mask.shape[0]
        r = room_seg_mask.reshape(R, -1)
        rc = r.sum(0)
        if not bool(((r == 0) | (r == 1)).all()) or float(rc.max()) > 1:
            raise ValueError("room_seg_mask must be a one-hot {0,1} mask over the room dimension")
        room_id = torch.where(rc > 0, r.argmax(0), torch.full_like(rc, NO_CLASS, dtype=torch.long)).to(torch.uint8)
    return seg_id.contiguous(), hl, room_id, C, R


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, albedo, rough, rough_womip, gt, empty, gtm, seg_id, hl, room_id, stage, loss_type, C, R, hw, unit_upstream=False):
        dev = rgb.device
        P = seg_id.numel()
        f = lambda t, n: None if t is None else t.detach().to(device=dev, dtype=torch.float32).reshape(P, n).contiguous()
        rgb_, alb_, gt_ = f(rgb, 3), f(albedo, 3), f(gt, 3)
        r_, rw_, e_, m_ = f(rough, 1), f(rough_womip, 1), f(empty, 1), f(gtm, 1)
        L = _lib.lib()
        ws = torch.empty(int(L.texir_loss_workspace_bytes(P, C, R)), device=dev, dtype=torch.uint8)
        out = torch.empty(2, device=dev, dtype=torch.float32)
        d_rgb = torch.empty((P, 3), device=dev, dtype=torch.float32)
        d_alb = torch.empty((P, 3), device=dev, dtype=torch.float32) if stage == 0 else None
        d_r = torch.empty((P,), device=dev, dtype=torch.float32) if stage != 0 else None
        _lib.check(L.texir_loss_forward(stage, loss_type, _lib.ptr(gt_), _lib.ptr(rgb_), _lib.ptr(alb_), _lib.ptr(r_), _lib.ptr(rw_), _lib.ptr(e_),
                                        _lib.ptr(m_), _lib.ptr(seg_id), _lib.ptr(hl), _lib.ptr(room_id), P, C, R, hw, _lib.ptr(ws), _lib.ptr(out),
                                        _lib.ptr(d_rgb), _lib.ptr(d_alb), _lib.ptr(d_r), _lib.stream_ptr()))
        ctx.grads = (d_rgb, d_alb, d_r)
        ctx.unit_upstream = unit_upstream
        ctx.shapes = (rgb.shape, None if albedo is None else albedo.shape, None if rough is None else rough.shape)
        # two 0-dim outputs (views of the kernel's result pair): the loss, and the segmentation term the reference returns as .item()
        loss, seg = out[0], out[1]
        ctx.mark_non_differentiable(seg)
        ctx.set_materialize_grads(False)          # (no zero tensor filled per step for the non-differentiable output)
        return loss, seg

    @staticmethod
    def backward(ctx, g0, _g_seg=None):
        if g0 is None:
            return (None,) * 16
        d_rgb, d_alb, d_r = ctx.grads
        s_rgb, s_alb, s_r = ctx.shapes
        if ctx.unit_upstream:
            # the loss is the root of the backward pass (trainer: loss.backward()): the upstream gradient is exactly 1 and the kernels'
            # gradients are handed on as they are -- three elementwise launches fewer per step
            sc = lambda t: t
        else:
            sc = lambda t: t * g0
        grgb = sc(d_rgb).reshape(s_rgb) if ctx.needs_input_grad[0] else None
        galb = sc(d_alb).reshape(s_alb) if (d_alb is not None and ctx.needs_input_grad[1]) else None
        gr = sc(d_r).reshape(s_r) if (d_r is not None and ctx.needs_input_grad[2]) else None
        return (grgb, galb, gr) + (None,) * 13


class RenderLoss(nn.Module):
    """models.loss.RenderLoss(loss_type='L1', w_gradient=0).forward(gt_img, preds, gt_mask, floor_max_mask, seg_mask,
    stage, room_seg_mask) -> (loss, seg_loss_item[, 0])  (loss.py:56,81-115)."""

    def __init__(self, loss_type="L1", w_gradient=0, lazy_item=False, unit_upstream=False):
        """lazy_item=True returns the seg term as a 0-dim device tensor instead of calling .item() (the reference's
        `seg_loss.item()` forces a host sync every step, which also forbids hipGraph capture of the step).
        unit_upstream=True promises that the returned loss is back-propagated as the root (`loss.backward()`, what the trainers do):
        the fused kernels' gradients are then passed on without the multiplication by the upstream gradient (= 1)."""
        super().__init__()
        self.lazy_item = lazy_item
        self.unit_upstream = unit_upstream
        if loss_type not in _LOSS_TYPES and loss_type not in _VARIANTS:
            raise Exception("Unknown loss_type!")                          # loss.py:76
        print("Using %s loss for comparing re-rendered radiance!" % {"msssim": "ms-ssim", "psnr": "PSNR"}.get(loss_type, loss_type))
        self.loss_type = loss_type
        self.w_gradient = w_gradient
        self._cache = {}

    def _compact(self, seg_mask, floor_max_mask, room_seg_mask, dev):
        key = tuple((t.data_ptr(), tuple(t.shape), t._version) if t is not None else None for t in (seg_mask, floor_max_mask, room_seg_mask))
        hit = self._cache.get(key)
        if hit is None:
            if len(self._cache) > 256:
                self._cache.clear()
            seg_id, hl, room_id, C, R = compact_masks(seg_mask, floor_max_mask, room_seg_mask)
            mv = lambda t: None if t is None else t.to(dev).contiguous()
            hit = (mv(seg_id), mv(hl), mv(room_id), C, R)
            self._cache[key] = hit
        return hit

    def forward(self, gt_img, preds, gt_mask, floor_max_mask, seg_mask, stage=0, room_seg_mask=None):
        if stage not in (0, 1, 2):
            raise ValueError("RenderLoss: stage must be 0, 1 or 2")
        rgb = preds["rgb"]
        dev = rgb.device
        seg_id, hl, room_id, C, R = self._compact(seg_mask, floor_max_mask, room_seg_mask if stage == 2 else None, dev)
        hw = int(rgb.shape[1] * rgb.shape[2])
        if self.loss_type in _VARIANTS:
            return self._variant(gt_img, preds, gt_mask, seg_id, hl, C, stage, hw)
        out = _LossFn.apply(rgb, preds["albedo"] if stage == 0 else None, preds["roughness"] if stage != 0 else None,
                            preds["roughness_womipmap"] if stage == 1 else None, gt_img, preds["empty_mask"], gt_mask if stage == 0 else None,
                            seg_id, hl, room_id if stage == 2 else None, stage, _LOSS_TYPES[self.loss_type], C, R if stage == 2 else 0, hw,
                            self.unit_upstream)
        loss, seg_item = out[0], (out[1].detach() if self.lazy_item else out[1].item())
        if stage == 0:
            return loss, seg_item
        return loss, seg_item, (0. if stage == 1 else 0)

    def _variant(self, gt_img, preds, gt_mask, seg_id, hl, C, stage, hw):
        """loss_type psnr / ssim / msssim (models/loss.py:65-73,117-140): the image term in stock torch, SegLoss in the fused kernel.
        The reference's PSNRLoss / SSIMLoss / MSSSIMLoss permute their inputs as [b,h,w,c] images, which only stage 0's tensors are
        (stages 1 and 2 pass [49,6,h,w,3] products and fail inside permute): the same restriction is kept, with a clear message."""
        from . import metrics as M
        from .models import hdr_scale
        if stage != 0:
            raise ValueError("RenderLoss(loss_type=%r) supports stage 0 only (the reference's image-shaped losses cannot take the "
                             "[classes,6,h,w,3] tensors of stages 1 and 2: models/loss.py:101,111,124,132)" % self.loss_type)
        rgb = preds["rgb"]
        a = hdr_scale(rgb * preds["empty_mask"] * gt_mask).permute(0, 3, 1, 2)
        b = hdr_scale(gt_img * gt_mask).permute(0, 3, 1, 2)
        if self.loss_type == "psnr":
            direct = -M.mse_to_psnr(torch.mean((b - a) ** 2))
        elif self.loss_type == "ssim":
            direct = 1.0 - M.ssim(b, a)
        else:
            direct = 1.0 - M.ms_ssim(b, a)
        out = _LossFn.apply(rgb.detach(), preds["albedo"], None, None, gt_img, preds["empty_mask"], gt_mask, seg_id, hl, None, 0, 2, C, 0, hw)
        loss, seg_item = direct + out[0], (out[1].detach() if self.lazy_item else out[1].item())
        return loss, seg_item
