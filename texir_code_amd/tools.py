"""The asset step between IrT generation and material estimation (tools/padding_texture.py:49-87): zero texels of the
irradiance texture (seams / gutters) take the value of their nearest non-zero texel (Euclidean distance transform), so that
mip-mapped fetches near chart borders do not bleed black.  One-time CPU step in the reference (scipy + torch grid_sample); kept
on the same ops here, including grid_sample's nearest-rounding quirk.  The external OIDN denoiser call that follows it in the
reference (:86-87) is not reproduced.

    python -m texir_code_amd.tools pad <.../0_irr_texture.hdr> [<.../irt.hdr>]
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


def main(argv):
    if len(argv) < 2 or argv[0] != "pad":
        print(__doc__)
        return 2
    src = argv[1]
    dst = argv[2] if len(argv) > 2 else src.replace("0_irr_texture", "irt")
    IO.write_hdr(dst, padding_texture(IO.read_hdr(src)))
    print("wrote", dst)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
