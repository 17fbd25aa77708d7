"""The asset step between IrT generation and material estimation (tools/padding_texture.py:49-87): zero texels of the
irradiance texture (seams / gutters) take the value of their nearest non-zero texel (Euclidean distance transform), so that
mip-mapped fetches near chart borders do not bleed black.  One-time CPU step in the reference (scipy + torch grid_sample); kept
on the same ops here, including grid_sample's nearest-rounding quirk.  The reference then pipes the padded texture through the
external Open Image Denoise binary (:86-87); `denoise_atrous` is a stand-in for that call (an edge-avoiding a-trous wavelet
filter, Dammertz et al. 2010, colour edge-stopping only like OIDN's image-only mode) -- NOT a re-implementation of OIDN's network.

    python -m texir_code_amd.tools pad <.../0_irr_texture.hdr> [<.../irt.hdr>] [--denoise]
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

from . import io_formats as IO


def padding_texture(img):
    """img [H,W,3] float32 -> padded copy"""
    from scipy import ndimage
    img = np.asarray(img, np.float32)
    h, w, _ = img.shape
    mask = np.asarray((img[:, :, 0] + img[:, :, 1] + img[:, :, 2]) == 0.0, dtype=np.uint8)
    if mask.all():
        return img.copy()
    _, indices = ndimage.distance_transform_edt(mask, return_indices=True)
    indices = torch.from_numpy(indices).permute(1, 2, 0).reshape(-1, 2)
    img_t = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
    uv = torch.zeros((h * w, 2), dtype=torch.float32)
    m = torch.from_numpy(mask.reshape(-1).astype(bool))
    uv[m] = indices[m][:, [1, 0]].float() / torch.tensor([w, h]).unsqueeze(0) * 2.0 - 1.0
    res = F.grid_sample(img_t, uv.reshape(1, h, w, 2), mode="nearest", align_corners=False)[0].permute(1, 2, 0).numpy()
    mf = mask.astype(np.float32)[:, :, None]
    return res * mf + img * (1 - mf)


def denoise_atrous(img, iterations=3, sigma_c=0.5, device=None):
    """edge-avoiding a-trous filter on log(1+x): `iterations` passes of the 5x5 B3-spline kernel with hole sizes 1, 2, 4, ... and the
    edge-stopping weight exp(-|dc|^2 / sigma_c^2) (sigma halves every pass).  Zero texels (unpadded seams) neither contribute nor change.
    Runs on `device` (default: the GPU when present); a 4k x 4k texture takes a few tens of milliseconds there."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    x = torch.as_tensor(np.asarray(img, np.float32), device=device)
    valid = (x.sum(-1, keepdim=True) != 0).float()
    c = torch.log1p(x.clamp(min=0))
    k1 = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0], device=device) / 16.0
    H, W, _ = c.shape
    for it in range(iterations):
        step, s2 = 1 << it, (sigma_c * 0.5 ** it) ** 2
        acc, wsum = torch.zeros_like(c), torch.zeros((H, W, 1), device=device)
        pad = 2 * step
        cp = F.pad(c.permute(2, 0, 1)[None], (pad, pad, pad, pad), mode="replicate")[0].permute(1, 2, 0)
        vp = F.pad(valid.permute(2, 0, 1)[None], (pad, pad, pad, pad), mode="constant", value=0.0)[0].permute(1, 2, 0)
        for dy in range(5):
            for dx in range(5):
                q = cp[dy * step:dy * step + H, dx * step:dx * step + W]
                w = k1[dy] * k1[dx] * torch.exp(-((q - c) ** 2).sum(-1, keepdim=True) / s2) * vp[dy * step:dy * step + H, dx * step:dx * step + W]
                acc += q * w
                wsum += w
        c = torch.where(valid > 0, acc / wsum.clamp(min=1e-20), c)
    return (torch.expm1(c) * valid).cpu().numpy()


def main(argv):
    flags = [a for a in argv if a.startswith("--")]
    argv = [a for a in argv if not a.startswith("--")]
    if len(argv) < 2 or argv[0] != "pad":
        print(__doc__)
        return 2
    src = argv[1]
    dst = argv[2] if len(argv) > 2 else src.replace("0_irr_texture", "irt")
    out = padding_texture(IO.read_hdr(src))
    if "--denoise" in flags:
        out = denoise_atrous(out)
    IO.write_hdr(dst, out)
    print("wrote", dst)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
