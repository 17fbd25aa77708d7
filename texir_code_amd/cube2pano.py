"""Cube-map -> equirectangular warp with the reference's conventions (utils/Cube2Pano.py:22-144): face order
[left, front, right, back, top, bottom], bilinear grid_sample with border padding, align_corners=False.
One-time G-buffer plumbing (models/tracer_o3d_irt.py:111), so it stays on stock torch ops."""
import numpy as np
import torch
import torch.nn.functional as F

# (divisor axis, u axis, u sign, v axis, v sign, hemisphere axis, hemisphere sign)
_FACES = [(0, 2, 1.0, 1, -1.0, 0, -1), (2, 0, 1.0, 1, -1.0, 2, 1), (0, 2, -1.0, 1, -1.0, 0, 1),
          (2, 0, -1.0, 1, -1.0, 2, -1), (1, 0, 1.0, 2, 1.0, 1, 1), (1, 0, 1.0, 2, -1.0, 1, -1)]


class Cube2Pano:
    def __init__(self, batch_size=1, pano_width=256, pano_height=128, cube_lenth=128, cube_channel=3, is_cuda=False, cube_padding_size=0):
        self.pano_width, self.pano_height, self.batch_size = pano_width, pano_height, batch_size
        self.cube_lenth, self.cube_channel, self.is_cuda, self.cube_padding_size = cube_lenth, cube_channel, is_cuda, cube_padding_size
        theta, phi = np.meshgrid(np.linspace(-np.pi, np.pi, pano_width, dtype=np.float32),
                                 np.linspace(0.5 * np.pi, -0.5 * np.pi, pano_height, dtype=np.float32))
        theta, phi = torch.from_numpy(theta), torch.from_numpy(phi)
        xyz = torch.stack([torch.cos(phi) * torch.sin(theta), torch.sin(phi), torch.cos(phi) * torch.cos(theta)], dim=2)
        grids, masks = [], []
        for (da, ua, us, va, vs, ha, hs) in _FACES:
            tmp = xyz / torch.abs(xyz[:, :, da:da + 1])
            u, v = us * tmp[:, :, ua], vs * tmp[:, :, va]
            inside = (u >= -1) * (u <= 1) * (v >= -1) * (v <= 1) * ((tmp[:, :, ha] < 0) if hs < 0 else (tmp[:, :, ha] > 0))
            grids.append(torch.stack([u, v], dim=2))
            masks.append(inside.unsqueeze(-1).float())
        self.grid = torch.stack(grids, 0).repeat(batch_size, 1, 1, 1)
        self.mask = torch.stack(masks, 0).repeat(batch_size, 1, 1, 1)
        if is_cuda and torch.cuda.is_available():
            self.grid, self.mask = self.grid.cuda(), self.mask.cuda()

    def ToPano(self, input, mode="bilinear"):
        """input [b, c*6, h, w] (face-major) -> [b, c, pano_h, pano_w]"""
        assert mode in ("bilinear", "nearest", "bicubic")
        L, pad = self.cube_lenth, self.cube_padding_size
        image = input.reshape(6 * self.batch_size, -1, L + 2 * pad, L + 2 * pad)
        g = self.grid.to(image.device) * self.mask.to(image.device).expand_as(self.grid)
        g = (g + 1) / 2
        g = (g * L + pad) / (L + 2 * pad) * 2 - 1.0
        out = F.grid_sample(image, g, mode=mode, padding_mode="border", align_corners=False)
        out = out * self.mask.to(image.device).permute(0, 3, 1, 2)
        out = out.unsqueeze(1).reshape(self.batch_size, 6, -1, self.pano_height, self.pano_width)
        return torch.sum(out, dim=1)
