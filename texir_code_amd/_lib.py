"""ctypes loader for libtexir_hip.so (C-ABI declared in include/texir_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, a TexirError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TEXIR_HIP_LIB") or os.path.join(_HERE, "libtexir_hip.so")   # (override: A/B builds of the kernels)
_LIB = None

class TexirError(RuntimeError):
    pass


def build():
    """compile libtexir_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc")])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TexirError("libtexir_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "-- there is no CPU fallback" % LIB_PATH)
        # Device pointers and streams come from PyTorch-ROCm, so the kernels must be launched through the SAME HIP
        # runtime instance torch uses: torch bundles its own libamdhip64.so (soname libamdhip64.so.7, the soname our
        # library needs), so load torch -- and pin its runtime -- before dlopen()ing ours.
        import torch
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tl):
            C.CDLL(tl, mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        L.texir_last_error.restype = C.c_char_p
        L.texir_version.restype = i32
        sig = {
            "texir_scene_create": [vp, i32, vp, i32, vp, vp, i32, i32, i32, C.POINTER(vp)],
            "texir_scene_destroy": [vp],
            "texir_scene_set_texture": [vp, vp, i32, i32, i32, vp],
            "texir_scene_texture_layout": [vp, vp],
            "texir_texel_pack": [vp, i64, vp, vp],
            "texir_texel_unpack": [vp, i64, vp],
            "texir_scene_info": [vp, vp],
            "texir_scene_scheduler": [vp, vp],
            "texir_scene_tune": [vp, vp, vp, vp, vp, i64, i32, i32, vp],
            "texir_scene_prefetch": [vp, i32, i32, vp],
            "texir_scene_reserve_scratch": [vp, i64, i32],
            "texir_trace_shade": [vp, vp, vp, i64, f32, vp, vp, vp, vp, vp],
            "texir_generate_dir": [vp, vp, vp, i64, i32, i32, vp, vp],
            "texir_irt_generate": [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, vp, vp],
            "texir_spec_forward": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp, vp, vp],
            "texir_spec_backward": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp, vp, vp],
            "texir_spec_forward_train": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp, vp, vp, vp],
            "texir_spec_backward_ws": [vp, vp, vp, vp, i64, i32, vp, vp, vp],
            "texir_diffuse_irradiance": [vp, vp, vp, vp, i64, i32, i32, vp, vp],
        }
        sig["texir_irt_kernel_name"] = [vp, i64, i32, C.c_char_p, i32]
        sig["texir_loss_forward"] = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp]
        sig["texir_scene_set_corner_normals"] = [vp, vp]
        sig["texir_gbuffer_cast"] = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        sig["texir_mip_build"] = [vp, vp, i32, i32, i32, i32, i32, vp]
        sig["texir_tex_fetch_forward"] = [vp, vp, i32, i32, i32, i32, vp, vp, i32, i64, vp, vp]
        sig["texir_tex_fetch_backward"] = [vp, vp, i32, i32, i32, i32, vp, vp, i32, i64, vp, vp]
        sig["texir_adam_step"] = [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, f32, vp]
        sig["texir_tex_fetch_backward_deferred"] = [vp, vp, i32, i32, i32, i32, vp, vp, i64, vp, vp]
        sig["texir_tex_taps"] = [i32, i32, i32, i32, vp, vp, i32, i64, vp, vp, vp]
        sig["texir_tex_gather_backward"] = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp]
        sig["texir_adam_step_tex"] = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, i32, f32, f32, vp]
        sig["texir_adam_tick"] = [vp, vp, i32, C.c_uint64, vp]
        sig["texir_adam_step_dev"] = [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, vp]
        sig["texir_adam_step_tex_dev"] = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, f32, f32, f32, f32, f32, vp]
        L.texir_batch_last_error.restype = C.c_char_p
        sig["texir_grad_add_masked"] = [vp, vp, vp, i64, i32, vp]
        sig["texir_tex_fetch_forward_batch"] = [vp, i32, vp]
        sig["texir_tex_gather_backward_batch"] = [vp, i32, vp]
        sig["texir_adam_step_tex_dev_batch"] = [vp, i32, vp]
        L.texir_reload_env.argtypes = []
        L.texir_reload_env.restype = i32
        L.texir_mip_levels.argtypes = [i32, i32, i32]
        L.texir_mip_levels.restype = i32
        L.texir_mip_elems.argtypes = [i32, i32, i32, i32]
        L.texir_mip_elems.restype = i64
        L.texir_loss_workspace_bytes.argtypes = [i64, i32, i32]
        L.texir_loss_workspace_bytes.restype = i64
        sig["texir_png_unfilter"] = [vp, i32, i32, i32, vp]
        L.texir_hdr_decode_scanlines.argtypes = [vp, i64, i32, i32, vp]
        L.texir_hdr_decode_scanlines.restype = i64
        L.texir_hdr_encode_rle.argtypes = [vp, i32, i32, vp, i64]
        L.texir_hdr_encode_rle.restype = i64
        sig["texir_env_switch"] = [C.c_char_p, C.POINTER(i32)]
        sig["texir_rgbe_encode"] = [vp, i64, vp]
        sig["texir_rgbe_decode"] = [vp, i64, vp]
        sig["texir_obj_parse"] = [vp, i64, C.POINTER(vp), vp]
        sig["texir_obj_take"] = [vp, vp, vp, vp, vp, vp, vp]
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = i32
        _LIB = L
    return _LIB


class TexFetchJob(C.Structure):
    """texir_tex_fetch_job (include/texir_hip.h)"""
    _fields_ = [("tex", C.c_void_p), ("mips_rest", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("levels", C.c_int32),
                ("build_from", C.c_int32), ("filter_mode", C.c_int32), ("uv", C.c_void_p), ("uv_da", C.c_void_p), ("P", C.c_int64), ("out", C.c_void_p)]


class TexGatherJob(C.Structure):
    """texir_tex_gather_job"""
    _fields_ = [("d_tex", C.c_void_p), ("grad_rest", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("levels", C.c_int32),
                ("seg_key", C.c_void_p), ("seg_start", C.c_void_p), ("seg_count", C.c_void_p), ("n_seg", C.c_int32), ("pix", C.c_void_p),
                ("weights", C.c_void_p), ("d_out", C.c_void_p), ("filter_mode", C.c_int32), ("defer_last_fold", C.c_int32), ("rest_mask", C.c_void_p), ("d_out2", C.c_void_p)]


class AdamTexJob(C.Structure):
    """texir_adam_tex_job"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("grad_mask", C.c_void_p), ("grad_level1", C.c_void_p), ("level1_mask", C.c_void_p),
                ("grad_level2", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("mip_level1", C.c_void_p),
                ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("hyper", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("clamp_lo", C.c_float), ("clamp_hi", C.c_float)]


MAX_BATCH = 4


def addr(t):
    """data pointer of a contiguous tensor as a plain int (ctypes Structure fields), or None"""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return t.data_ptr()


def batch_call(name, jobs):
    """one batched launch sequence (texir_*_batch) over a list of job structures"""
    L = lib()
    arr = (type(jobs[0]) * len(jobs))(*jobs)
    rc = getattr(L, name)(arr, len(jobs), stream_ptr())
    if rc != 0:
        raise TexirError("libtexir_hip: %s (code %d)" % (L.texir_batch_last_error().decode(), rc))


def env_switch(name):
    """the library's own (load-time or last texir_reload_env) reading of a TEXIR_* switch"""
    v = C.c_int32()
    check(lib().texir_env_switch(name.encode(), C.byref(v)))
    return int(v.value)


def reload_env():
    """re-read the library's TEXIR_* switches (parsed once at load, csrc/env.h) after os.environ was changed"""
    if _LIB is not None:
        _LIB.texir_reload_env()


def check(rc):
    if rc != 0:
        raise TexirError("libtexir_hip: %s (code %d)" % (lib().texir_last_error().decode(), rc))


def ptr(t):
    """device/host pointer of a contiguous tensor / ndarray, or None"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(t.data_ptr())
    return t.ctypes.data_as(C.c_void_p)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
