"""isolated timing of the fused texture Adam kernels: host-argument entry (texir_adam_step_tex) vs device-record entry (texir_adam_step_tex_dev)
usage: python tools/probes/adam_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from texir_code_amd import _lib

L = _lib.lib()
dev = torch.device("cuda", 0)
for C in (3, 1):
    H = W = 4096
    p = torch.rand(H, W, C, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
    g1 = torch.randn((H // 2) * (W // 2) * C, device=dev); g2 = torch.randn((H // 4) * (W // 4) * C, device=dev)
    mip1 = torch.empty((H // 2) * (W // 2) * C, device=dev)
    hyper = torch.tensor([0.03, 0.9], device=dev)
    st = _lib.stream_ptr()
    def host():
        _lib.check(L.texir_adam_step_tex(_lib.ptr(p), None, None, _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(m), _lib.ptr(v), _lib.ptr(mip1), H, W, C, 0.03, 0.9, 0.999, 1e-8, 3, 0.0, 1.0, st))
    def devf():
        _lib.check(L.texir_adam_step_tex_dev(_lib.ptr(p), None, None, _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(m), _lib.ptr(v), _lib.ptr(mip1), H, W, C, _lib.ptr(hyper), 0.9, 0.999, 1e-8, 0.0, 1.0, st))
    for name, f in (("host-args", host), ("dev-record", devf), ("host-args", host), ("dev-record", devf)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("C=%d %-10s %.1f us  (%.2f TB/s on 28 B/param + level-1 in/out)" % (C, name, us, (28.0 * H * W * C + 2 * 4.0 * H * W * C / 4) / us / 1e6))
