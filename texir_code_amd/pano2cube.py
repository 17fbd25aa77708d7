"""Pano2Cube -- equirectangular panorama -> six cube faces by a precomputed grid_sample grid (utils/Pano2Cube.py:24-102).
Face order 0-left, 1-front, 2-right, 3-back, 4-top, 5-bottom; used by the dataset adapters (datasets/dataset.py:361,516)."""
import numpy as np
import torch
import torch.nn.functional as F


def rodrigues(rvec):
    """rotation vector -> 3x3 matrix (cv2.Rodrigues: R = cos*I + (1-cos)*r r^T + sin*[r]x, evaluated in double)"""
    rvec = np.asarray(rvec, np.float64)
    theta = np.linalg.norm(rvec)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3, dtype=np.float32)
    r = rvec / theta
    c, s = np.cos(theta), np.sin(theta)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return (c * np.eye(3) + (1 - c) * np.outer(r, r) + s * K).astype(np.float32)


class Pano2Cube:
    def __init__(self, batch_size=1, pano_width=256, pano_height=128, cube_lenth=128, cube_channel=3, is_cuda=False):
        self.pano_width, self.pano_height, self.batch_size = pano_width, pano_height, batch_size
        self.cube_lenth, self.cube_channel, self.is_cuda = cube_lenth, cube_channel, is_cuda
        horizon = np.array([-90.0, 0.0, 90.0, 180.0]) / 180.0 * np.pi                 # :37 about y
        vertical = np.array([-90.0, 90.0]) / 180.0 * np.pi                            # :38 about x
        rots = [rodrigues(h * np.array([0, 1, 0], np.float32)) for h in horizon] + \
               [rodrigues(v * np.array([1, 0, 0], np.float32)) for v in vertical]
        self.rorate_list = [torch.from_numpy(r) for r in rots]
        sx, sy = np.meshgrid(np.linspace(-1.0, 1.0, cube_lenth), np.linspace(1.0, -1.0, cube_lenth))     # :52-56
        r = np.sqrt(sy * sy + sx * sx + 1)
        sx = sx / r
        sy = sy / r
        sz = np.sqrt(1 - sy * sy - sx * sx)
        xyz = torch.from_numpy(np.array([sx, sy, sz], dtype=np.float32)).view(3, cube_lenth * cube_lenth)
        self.uv = []
        for R in self.rorate_list:
            t = torch.matmul(R, xyz).permute(1, 0)
            azimuth = torch.atan2(t[:, 0], t[:, 2]).view(1, cube_lenth, cube_lenth, 1)
            elevation = torch.asin(t[:, 1]).view(1, cube_lenth, cube_lenth, 1)
            u = azimuth / np.pi                                                        # :80-81 grid_sample coordinates
            v = -elevation / (np.pi / 2)
            self.uv.append(torch.cat([u.repeat(batch_size, 1, 1, 1), v.repeat(batch_size, 1, 1, 1)], dim=3))

    def Tocube(self, input, mode="bilinear"):
        """[b, c, h, w] -> [b, 6*c, cube, cube] (no wrap-around interpolation at the back face's seam, as in the reference)"""
        assert mode in ["bilinear", "nearest"]
        out = []
        for i in range(6):
            uv = self.uv[i].to(input.device)
            out.append(F.grid_sample(input, uv, mode=mode, padding_mode="border", align_corners=False))
        return torch.cat(out, dim=1)
