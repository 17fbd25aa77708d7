"""nvdiffrast `dr.texture` replacement (models/mat_nvdiffrast.py:131-139): bilinear ('linear') and trilinear
('linear-mipmap-linear') fetch with wrap boundary, differentiable wrt the texture.  Semantics restated from
nvdiffrast's public documentation (parity unpinned: nvdiffrast is not installable here)."""
import torch

from . import _lib

FILTER = {"linear": 0, "linear-mipmap-linear": 1}


class _TexFetch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, uv_da, mode, max_mip_level):
        L = _lib.lib()
        H, W, C = tex.shape
        P = uv.shape[0]
        levels = int(L.texir_mip_levels(H, W, max_mip_level)) if mode == 1 else 1
        n = int(L.texir_mip_elems(H, W, C, levels))
        if levels > 1:
            mips = torch.empty(n, device=tex.device, dtype=torch.float32)
            mips[: H * W * C].copy_(tex.detach().reshape(-1))
            _lib.check(L.texir_mip_build(_lib.ptr(mips), H, W, C, levels, _lib.stream_ptr()))
        else:
            mips = tex.detach().contiguous().reshape(-1)
        out = torch.empty((P, C), device=tex.device, dtype=torch.float32)
        _lib.check(L.texir_tex_fetch_forward(_lib.ptr(mips), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, P, _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(uv, uv_da)
        ctx.meta = (H, W, C, levels, mode, n)
        return out

    @staticmethod
    def backward(ctx, d_out):
        uv, uv_da = ctx.saved_tensors
        H, W, C, levels, mode, n = ctx.meta
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        g = torch.zeros(n, device=d_out.device, dtype=torch.float32)
        d_out = d_out.contiguous()
        _lib.check(_lib.lib().texir_tex_fetch_backward(_lib.ptr(g), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, uv.shape[0],
                                                       _lib.ptr(d_out), _lib.stream_ptr()))
        return g[: H * W * C].reshape(H, W, C), None, None, None, None


def texture(tex, uv, uv_da=None, filter_mode="linear", max_mip_level=13):
    """tex [H,W,C] (or [1,H,W,C]); uv [...,2]; uv_da [...,4] (du/dX,du/dY,dv/dX,dv/dY) -> [...,C]"""
    if tex.dim() == 4:
        tex = tex[0]
    lead = uv.shape[:-1]
    uvf = uv.reshape(-1, 2).to(torch.float32).contiguous()
    daf = None if uv_da is None else uv_da.reshape(-1, 4).to(torch.float32).contiguous()
    mode = FILTER[filter_mode]
    if mode == 1 and daf is None:
        raise ValueError("linear-mipmap-linear needs uv_da")
    out = _TexFetch.apply(tex.to(torch.float32), uvf, daf, mode, int(max_mip_level))
    return out.reshape(*lead, tex.shape[-1])
