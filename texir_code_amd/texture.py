"""nvdiffrast `dr.texture` replacement (models/mat_nvdiffrast.py:131-139): bilinear ('linear') and trilinear
('linear-mipmap-linear') fetch with wrap boundary, differentiable wrt the texture.  Semantics restated from
nvdiffrast's public documentation (parity unpinned: nvdiffrast is not installable here).

Level 0 of the mip stack is the texture tensor itself; levels 1.. are built into a side buffer.  The side buffer is cached
per (storage, version): a frozen texture (the irradiance texture) builds its stack once, and several fetches of the same
parameter inside one step share one build."""
import torch

from . import _lib

FILTER = {"linear": 0, "linear-mipmap-linear": 1}
_MIP_CACHE = {}


def _mips_for(tex, levels):
    """levels 1.. of `tex` (a contiguous [H,W,C] float32 CUDA tensor), cached on (data_ptr, version, shape)"""
    H, W, C = tex.shape
    key = (tex.data_ptr(), tex._version, H, W, C, levels, tex.device.index)
    hit = _MIP_CACHE.get(tex.data_ptr())
    if hit is not None and hit[0] == key:
        return hit[1]
    L = _lib.lib()
    n = int(L.texir_mip_elems(H, W, C, levels))
    rest = hit[1] if (hit is not None and hit[1].numel() == n and hit[1].device == tex.device) else torch.empty(n, device=tex.device, dtype=torch.float32)
    _lib.check(L.texir_mip_build(_lib.ptr(tex), _lib.ptr(rest), H, W, C, levels, _lib.stream_ptr()))
    if len(_MIP_CACHE) > 64:
        _MIP_CACHE.clear()
    _MIP_CACHE[tex.data_ptr()] = (key, rest)
    return rest


class _TexFetch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, uv_da, mode, max_mip_level):
        L = _lib.lib()
        H, W, C = tex.shape
        P = uv.shape[0]
        t0 = tex.detach()
        if not t0.is_contiguous():
            t0 = t0.contiguous()
        levels = int(L.texir_mip_levels(H, W, max_mip_level)) if mode == 1 else 1
        rest = _mips_for(t0, levels) if levels > 1 else None
        out = torch.empty((P, C), device=tex.device, dtype=torch.float32)
        _lib.check(L.texir_tex_fetch_forward(_lib.ptr(t0), _lib.ptr(rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, P, _lib.ptr(out),
                                             _lib.stream_ptr()))
        ctx.save_for_backward(uv, uv_da)
        ctx.meta = (H, W, C, levels, mode)
        return out

    @staticmethod
    def backward(ctx, d_out):
        uv, uv_da = ctx.saved_tensors
        H, W, C, levels, mode = ctx.meta
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        L = _lib.lib()
        d_tex = torch.zeros((H, W, C), device=d_out.device, dtype=torch.float32)
        g_rest = torch.zeros(int(L.texir_mip_elems(H, W, C, levels)), device=d_out.device, dtype=torch.float32) if levels > 1 else None
        d_out = d_out.contiguous()
        _lib.check(L.texir_tex_fetch_backward(_lib.ptr(d_tex), _lib.ptr(g_rest), H, W, C, levels, _lib.ptr(uv), _lib.ptr(uv_da), mode, uv.shape[0],
                                              _lib.ptr(d_out), _lib.stream_ptr()))
        return d_tex, None, None, None, None


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
    out = _TexFetch.apply(tex if tex.dtype == torch.float32 else tex.to(torch.float32), uvf, daf, mode, int(max_mip_level))
    return out.reshape(*lead, tex.shape[-1])
