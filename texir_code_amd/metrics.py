"""Image metrics the reference takes from utils/general.py and pytorch_msssim (models/loss.py:117-140, trainer/train_material_syn.py:453-481,
tester/test_error.py:186-188): tonemapping, MSE <-> PSNR, SSIM and MS-SSIM.  pytorch_msssim is not installable here: its published algorithm
(Gaussian window 11, sigma 1.5, K = (0.01, 0.03), `valid` convolution, per-channel mean, 5-scale weights) is restated -- parity unpinned."""
import math

import torch
import torch.nn.functional as F


def tonemapping(img):
    """utils/general.py:79-85"""
    return torch.clamp(img ** (1 / 2.2), 0.0, 1.0)


def mse_to_psnr(mse):
    """utils/general.py:71-73 (maximum pixel value 1)"""
    return -10.0 / math.log(10.0) * torch.log(mse)


def scale_compute(gt, prediction):
    """utils/general.py:128-130: least-squares scalar s minimising |s * prediction - gt|"""
    p, g = prediction.flatten().double(), gt.flatten().double()
    return (torch.dot(p, g) / torch.dot(p, p)).float().detach()


def _gauss(size=11, sigma=1.5, device=None):
    c = torch.arange(size, dtype=torch.float32, device=device) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _ssim_cs(x, y, data_range=1.0, win=None):
    """x, y [B,C,H,W] -> (ssim per channel [B,C], cs per channel [B,C])"""
    C = x.shape[1]
    if win is None:
        win = _gauss(device=x.device)
    k = win.numel()

    def blur(t):
        t = F.conv2d(t, win.reshape(1, 1, k, 1).expand(C, 1, k, 1), groups=C) if t.shape[2] >= k else t
        return F.conv2d(t, win.reshape(1, 1, 1, k).expand(C, 1, 1, k), groups=C) if t.shape[3] >= k else t

    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = blur(x), blur(y)
    s1, s2, s12 = blur(x * x) - mu1 * mu1, blur(y * y) - mu2 * mu2, blur(x * y) - mu1 * mu2
    cs = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs
    return ssim.flatten(2).mean(-1), cs.flatten(2).mean(-1)


def ssim(x, y, data_range=1.0, nonnegative_ssim=True):
    """pytorch_msssim.SSIM(data_range=1, size_average=True, channel=3, nonnegative_ssim=True) on [B,C,H,W]"""
    s, _ = _ssim_cs(x, y, data_range)
    if nonnegative_ssim:
        s = torch.relu(s)
    return s.mean()


def ms_ssim(x, y, data_range=1.0):
    """pytorch_msssim.MS_SSIM(data_range=1, size_average=True, channel=3) on [B,C,H,W] (smaller side > 160)"""
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("ms_ssim: image side must exceed %d pixels" % ((11 - 1) * 2 ** 4))
    weights = torch.tensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333], device=x.device)
    win = _gauss(device=x.device)
    mcs = []
    for i in range(5):
        s, cs = _ssim_cs(x, y, data_range, win)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [d % 2 for d in x.shape[2:]]
            x, y = F.avg_pool2d(x, 2, padding=pad), F.avg_pool2d(y, 2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(s)], 0)                       # [5,B,C]
    return torch.prod(vals ** weights.reshape(-1, 1, 1), 0).mean()
