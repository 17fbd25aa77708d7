"""Scene handle: the MI355X replacement of the Open3D RaycastingScene + CPU radiance texture that
TracerO3d.__init__ / MaterialModel.__init__ build (models/tracer_o3d_irt.py:75-89, models/mat_nvdiffrast.py:87-101)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

MODES = {"uniform": 0, "cosine": 1, "importance": 2}

# hipFree is not allowed while a stream capture (hipGraph) is in progress, and a Scene may be garbage-collected at any time:
# destruction is deferred while this counter is non-zero (see defer_destroy()).
_DEFER = [0]
_PENDING = []


class defer_destroy:
    """context manager: scene handles released inside are destroyed on exit (used around hipGraph capture)"""

    def __enter__(self):
        _DEFER[0] += 1

    def __exit__(self, *a):
        _DEFER[0] -= 1
        if _DEFER[0] == 0 and _lib._LIB is not None:
            while _PENDING:
                _lib._LIB.texir_scene_destroy(_PENDING.pop())


def _dev_f32(t, device):
    if not torch.is_tensor(t):
        t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


class Scene:
    """verts [V,3], tris [T,3] (primitive id = row, Open3D order), tri_uvs [3T,2] (= np.asarray(mesh.triangle_uvs)),
    hdr_texture [Ht,Wt,3] float32 ALREADY RGB + vertically flipped + exposure-scaled (tracer_o3d_irt.py:77-81)."""

    def __init__(self, verts, tris, tri_uvs, hdr_texture, device=None):
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
        tri_uvs = np.ascontiguousarray(tri_uvs, np.float32).reshape(-1, 2)
        hdr = np.ascontiguousarray(hdr_texture, np.float32)
        if tri_uvs.shape[0] != 3 * tris.shape[0]:
            raise ValueError("tri_uvs must be [3T,2]")
        if hdr.ndim != 3 or hdr.shape[2] != 3:
            raise ValueError("hdr_texture must be [Ht,Wt,3]")
        self.n_tris = tris.shape[0]
        self.tex_shape = hdr.shape
        h = C.c_void_p()
        _lib.check(_lib.lib().texir_scene_create(_lib.ptr(verts), verts.shape[0], _lib.ptr(tris), tris.shape[0], _lib.ptr(tri_uvs),
                                                 _lib.ptr(hdr), hdr.shape[0], hdr.shape[1], self.device.index, C.byref(h)))
        self.h = h

    def __del__(self):
        h = getattr(self, "h", None)
        if h and _lib is not None and getattr(_lib, "_LIB", None) is not None:      # (module globals may be gone at interpreter exit)
            try:
                if _DEFER[0] > 0:
                    _PENDING.append(h)
                else:
                    _lib._LIB.texir_scene_destroy(h)
            except Exception:
                pass
            self.h = None

    def info(self):
        out = np.zeros(8, np.int64)
        _lib.check(_lib.lib().texir_scene_info(self.h, _lib.ptr(out)))
        keys = ["inner_nodes", "triangles", "max_depth", "node_bytes", "tri_bytes", "uv_bytes", "tex_bytes", "device"]
        d = dict(zip(keys, (int(x) for x in out)))
        sch = np.zeros(2, np.float64)
        _lib.check(_lib.lib().texir_scene_scheduler(self.h, _lib.ptr(sch)))
        d["sched_weight"], d["node_step_fill"] = int(sch[0]), (None if sch[1] < 0 else round(float(sch[1]), 4))
        d["tex_layout"] = self.texture_layout()
        return d

    def texture_layout(self):
        """layout of the hit shader's texture copy in force: 3 / 4 = 4-byte shared-exponent texels (the texture packs exactly: an RGBE file times a power
        of two), 2 / 1 / 0 = float32 (include/texir_hip.h texir_scene_texture_layout)"""
        out = np.zeros(1, np.int32)
        _lib.check(_lib.lib().texir_scene_texture_layout(self.h, _lib.ptr(out)))
        return int(out[0])

    def reserve_irt_scratch(self, n_ids, n_samples):
        """before RECORDING irt_generate over n_ids listed texels into a hipGraph: reserve the partial-sum scratch the recorded launch will use (a recorded
        graph must not allocate; include/texir_hip.h texir_scene_reserve_scratch).  Eager calls need nothing."""
        _lib.check(_lib.lib().texir_scene_reserve_scratch(self.h, int(n_ids), int(n_samples)))

    def irt_kernel_name(self, n_ids, n_samples):
        """the kernel form one irt_generate call over n_ids listed texels launches (the launcher's own decision)"""
        buf = C.create_string_buffer(96)
        _lib.check(_lib.lib().texir_irt_kernel_name(self.h, int(n_ids), int(n_samples), buf, 96))
        return buf.value.decode()

    def set_texture(self, tex):
        """tex [Ht,Wt,3] tensor (device or host) -- stage -1's temporary light-source-only texture (mat_nvdiffrast.py:141-150)"""
        if torch.is_tensor(tex) and tex.is_cuda:
            tex = tex.to(torch.float32).contiguous()
            _lib.check(_lib.lib().texir_scene_set_texture(self.h, _lib.ptr(tex), tex.shape[0], tex.shape[1], 1, _lib.stream_ptr()))
            torch.cuda.current_stream().synchronize()
        else:
            a = np.ascontiguousarray(tex.numpy() if torch.is_tensor(tex) else tex, np.float32)
            _lib.check(_lib.lib().texir_scene_set_texture(self.h, _lib.ptr(a), a.shape[0], a.shape[1], 0, _lib.stream_ptr()))
            torch.cuda.current_stream().synchronize()

    # -- query_irf (tracer_o3d_irt.py:240-269) ----------------------------------------------------------
    def trace_shade(self, org, dir, t_min=1e-4, return_hits=False):
        org = _dev_f32(org, self.device).reshape(-1, 3)
        dir = _dev_f32(dir, self.device).reshape(-1, 3)
        R = org.shape[0]
        rad = torch.empty((R, 3), device=self.device, dtype=torch.float32)
        t = pid = uv = None
        if return_hits:
            t = torch.empty(R, device=self.device, dtype=torch.float32)
            pid = torch.empty(R, device=self.device, dtype=torch.int32)
            uv = torch.empty((R, 2), device=self.device, dtype=torch.float32)
        if R > 0:                                   # (zero rays: empty results; an empty tensor has no pointer to hand to the library)
            _lib.check(_lib.lib().texir_trace_shade(self.h, _lib.ptr(org), _lib.ptr(dir), R, float(t_min), _lib.ptr(rad), _lib.ptr(t),
                                                    _lib.ptr(pid), _lib.ptr(uv), _lib.stream_ptr()))
        return (rad, t, pid, uv) if return_hits else rad

    # -- TracerO3d.forward hot loop (tracer_o3d_irt.py:156-178) -------------------------------------------
    def irt_generate(self, pos, nrm, shift, n_samples, mode="uniform", texel_ids=None, out=None, stats=False):
        pos = _dev_f32(pos, self.device).reshape(-1, 3)
        nrm = _dev_f32(nrm, self.device).reshape(-1, 3)
        Nt = pos.shape[0]
        shift = _dev_f32(shift, self.device).reshape(Nt, 2)
        if out is None:
            out = torch.zeros((Nt, 3), device=self.device, dtype=torch.float32)
        ids = None
        n_ids = 0
        if texel_ids is not None:
            ids = texel_ids.to(device=self.device, dtype=torch.int32).contiguous()
            n_ids = ids.numel()
        st = torch.zeros(32, device=self.device, dtype=torch.int64) if stats else None         # [0:6] counters; [8:21] cycle probe of a TEXIR_CHAIN_PROBE build
        if Nt == 0 or (texel_ids is not None and n_ids == 0):
            # nothing to do.  (An EMPTY id list must not reach the library: its null data pointer would read as "no list = all texels" --
            # the case of a rank whose shard of a small texel list is empty, dist_util.shard_block_cyclic)
            return (out, st) if stats else out
        if ids is not None and n_ids >= 65536 and not getattr(self, "_tuned", False) and not torch.cuda.is_current_stream_capturing():
            # once per scene, before its first long launch: measure how full the scene's node steps run and pick the phase scheduler's weight
            # (texir_scene_tune: ~2 ms, blocking -- which is why it lives here and not inside the asynchronous texir_irt_generate)
            _lib.check(_lib.lib().texir_scene_tune(self.h, _lib.ptr(pos), _lib.ptr(nrm), _lib.ptr(shift), _lib.ptr(ids), n_ids, int(n_samples),
                                                   MODES[mode], _lib.stream_ptr()))
            self._tuned = int(n_samples) >= 256
        _lib.check(_lib.lib().texir_irt_generate(self.h, _lib.ptr(pos), _lib.ptr(nrm), _lib.ptr(shift), _lib.ptr(ids), n_ids, Nt,
                                                 int(n_samples), MODES[mode], _lib.ptr(out), _lib.ptr(st), _lib.stream_ptr()))
        return (out, st) if stats else out


def generate_dir(normals, num_sample_dir, shift, mode="uniform", roughness=None):
    """utils/sample_util.py:63-146 on the GPU; `shift` [b,2] is the torch.rand(b,1,2) of :102 made explicit."""
    dev = normals.device
    normals = normals.to(torch.float32).contiguous().reshape(-1, 3)
    b = normals.shape[0]
    shift = _dev_f32(shift, dev).reshape(b, 2)
    r = None if roughness is None else roughness.to(torch.float32).contiguous().reshape(b)
    L = torch.empty((b, num_sample_dir, 3), device=dev, dtype=torch.float32)
    if b == 0:
        return L
    _lib.check(_lib.lib().texir_generate_dir(_lib.ptr(normals), _lib.ptr(r), _lib.ptr(shift), b, int(num_sample_dir), MODES[mode],
                                             _lib.ptr(L), _lib.stream_ptr()))
    return L


def spec_forward_raw(scene, normal, albedo, rough, points, irr, cam, shift, S, clamp_eps=1e-14, lighting=None, want_dw=False):
    """texir_spec_forward on contiguous float32 tensors (no autograd): -> (rgb [P,3], Ls [P,S,3] = the traced -- or given -- lighting the backward needs,
    dw [P,S] | None = the sample weights' derivatives wrt roughness when want_dw: the training form, texir_spec_forward_train)"""
    P = normal.shape[0]
    rgb = torch.empty((P, 3), device=normal.device, dtype=torch.float32)
    # lighting given: specular_reflectance on the caller's radiance (no tracing); else traced and kept for the backward
    Ls = torch.empty((P, S, 3), device=normal.device, dtype=torch.float32) if lighting is None else lighting
    dw = torch.empty((P, S), device=normal.device, dtype=torch.float32) if want_dw else None
    if P > 0:
        L = _lib.lib()
        head = (None if scene is None else scene.h, _lib.ptr(normal), _lib.ptr(albedo), _lib.ptr(rough), _lib.ptr(points), _lib.ptr(irr), _lib.ptr(cam), _lib.ptr(shift),
                P, S, float(clamp_eps), 0 if lighting is None else 1, _lib.ptr(rgb), _lib.ptr(Ls))
        if want_dw:
            _lib.check(L.texir_spec_forward_train(*head, _lib.ptr(dw), _lib.stream_ptr()))
        else:
            _lib.check(L.texir_spec_forward(*head, _lib.stream_ptr()))
    return rgb, Ls, dw


def spec_backward_raw(normal, rough, points, irr, cam, shift, Ls, d_rgb, S, clamp_eps=1e-14, need_albedo=True, need_rough=True, dw=None):
    """d rgb -> (d albedo [P,3] | None, d roughness [P] | None), no autograd.  dw (from spec_forward_raw(want_dw=True)): texir_spec_backward_ws, a stream over
    what the forward kept; else texir_spec_backward, which recomputes the sample chain"""
    P = normal.shape[0]
    d_rgb = d_rgb.contiguous()
    d_a = torch.empty((P, 3), device=normal.device, dtype=torch.float32) if need_albedo else None
    d_r = torch.empty((P,), device=normal.device, dtype=torch.float32) if need_rough else None
    if P > 0 and (need_albedo or need_rough):
        if dw is not None:
            _lib.check(_lib.lib().texir_spec_backward_ws(_lib.ptr(irr), _lib.ptr(Ls), _lib.ptr(dw), _lib.ptr(d_rgb), P, S, _lib.ptr(d_a), _lib.ptr(d_r), _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().texir_spec_backward(_lib.ptr(normal), _lib.ptr(rough), _lib.ptr(points), _lib.ptr(irr), _lib.ptr(cam),
                                                      _lib.ptr(shift), _lib.ptr(Ls), _lib.ptr(d_rgb), P, S, float(clamp_eps), _lib.ptr(d_a), _lib.ptr(d_r),
                                                      _lib.stream_ptr()))
    return d_a, d_r


class _SpecRender(torch.autograd.Function):
    """render + specular_reflectance (mat_nvdiffrast.py:201-249,260-279) with its analytic backward."""

    @staticmethod
    def forward(ctx, scene, normal, albedo, rough, points, irr, cam, shift, S, clamp_eps=1e-14, lighting=None):
        # roughness is being optimised: keep the weights' derivatives the forward computes anyway, and the backward needs no second pass over the sample chain
        want_dw = bool(ctx.needs_input_grad[3]) and _TRAIN_FORM[0]
        rgb, Ls, dw = spec_forward_raw(scene, normal, albedo, rough, points, irr, cam, shift, S, clamp_eps, lighting, want_dw)
        ctx.save_for_backward(normal, rough, points, irr, cam, shift, Ls, dw)
        ctx.S, ctx.clamp_eps = S, float(clamp_eps)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        normal, rough, points, irr, cam, shift, Ls, dw = ctx.saved_tensors
        d_a, d_r = spec_backward_raw(normal, rough, points, irr, cam, shift, Ls, d_rgb, ctx.S, ctx.clamp_eps, ctx.needs_input_grad[2], ctx.needs_input_grad[3], dw)
        return None, None, d_a, d_r, None, None, None, None, None, None, None


# A/B switch (TEXIR_SPEC_TRAIN_FORM=0: the backward recomputes the sample chain, texir_spec_backward -- the round-3 form)
_TRAIN_FORM = [__import__("os").environ.get("TEXIR_SPEC_TRAIN_FORM", "1") != "0"]


def spec_shift_arg(shift, P, dev):
    """the `shift` argument of the specular kernels: a PINNED host tensor is handed over as it is (pinned allocations are mapped into the device's address
    space: a recorded hipGraph reads each step's freshly drawn shifts without a staging copy); anything else becomes a float32 device tensor"""
    if torch.is_tensor(shift) and shift.device.type == "cpu" and shift.is_pinned() and shift.dtype == torch.float32 and shift.is_contiguous() and shift.numel() == 2 * P:
        return shift.reshape(P, 2)
    return shift.to(device=dev, dtype=torch.float32).reshape(P, 2).contiguous()


def spec_render(scene, normal, albedo, roughness, points, irr, cam_position, shift, num_samples, clamp_eps=1e-14, lighting=None):
    """rgb [P,3] = irr*albedo/pi + GGX specular; differentiable wrt albedo [P,3] and roughness [P] / [P,1].
    clamp_eps: floor of the BRDF denominators (1e-14 in mat_nvdiffrast.py, 1e-6 in the evaluation model test_nvdiffrast.py).
    lighting [P,S,3]: use the caller's radiance instead of tracing (the specular_reflectance function seam; scene may be None)."""
    dev = scene.device if scene is not None else normal.device
    P = normal.reshape(-1, 3).shape[0]
    f = lambda t, s: t.to(device=dev, dtype=torch.float32).reshape(*s).contiguous()
    S = int(num_samples)
    sh = spec_shift_arg(shift, P, dev)
    return _SpecRender.apply(scene, f(normal, (P, 3)), f(albedo, (P, 3)), f(roughness, (P,)), f(points, (P, 3)), f(irr, (P, 3)),
                             f(cam_position, (3,)), sh, S, float(clamp_eps), None if lighting is None else f(lighting, (P, S, 3)))


def diffuse_irradiance(scene, points, normals, shift, num_samples, sample_type="uniform"):
    """the lighting integral of diffuse_reflectance (mat_nvdiffrast.py:252-258): diffuse_reflectance(query_irf(...), l, n, albedo, type)/N
    == diffuse_irradiance(...) * albedo / pi.  points (already offset) / normals [P,3], shift [P,2] -> [P,3]"""
    dev = scene.device
    P = points.reshape(-1, 3).shape[0]
    f = lambda t, s: t.to(device=dev, dtype=torch.float32).reshape(*s).contiguous()
    out = torch.empty((P, 3), device=dev, dtype=torch.float32)
    if P == 0:
        return out
    _lib.check(_lib.lib().texir_diffuse_irradiance(scene.h, _lib.ptr(f(points, (P, 3))), _lib.ptr(f(normals, (P, 3))), _lib.ptr(f(shift, (P, 2))), P,
                                                   int(num_samples), MODES[sample_type], _lib.ptr(out), _lib.stream_ptr()))
    return out
