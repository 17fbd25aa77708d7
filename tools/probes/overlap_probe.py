"""Can the (compute-bound) specular trace of the next step overlap the (HBM-bound) albedo Adam of this step?  Times both alone, back to back on
one stream, and concurrently on two streams.   usage: python tools/probes/overlap_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from texir_code_amd import _lib, scene as S

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.set_num_threads(1)
sc0, pos, nrm, valid, shift, res, spp = bench.make_workload("c4")
sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
irr = torch.rand(res * res, 3, device=dev)
model, views, data, loss_fn, opt = bench.mat_setup(sc, sc0, irr, res, dev, n_views=2)
mvp, cam = views[0]
with torch.no_grad():
    gb = model._gbuffer(mvp, 0)
    out = model(mvp, 0, cam, 2)
    albedo, _, rough, irr_px = model._fetch_materials(gb, womipmap=False)
P = 6 * 128 * 128
n, pts = gb["normal"].reshape(P, 3).contiguous(), gb["_points"].reshape(P, 3).contiguous()
a, r, ir = albedo.reshape(P, 3).contiguous(), rough.reshape(P).contiguous(), irr_px.reshape(P, 3).contiguous()
camd = cam.to(dev)
sh = torch.rand(P, 2, device=dev)
rgb = torch.empty(P, 3, device=dev); Ls = torch.empty(P, 16, 3, device=dev)
L = _lib.lib()
H = W = 4096; C = 3
p = model.materials_a.detach(); m = torch.zeros_like(p); v = torch.zeros_like(p)
g1 = torch.randn((H // 2) * (W // 2) * C, device=dev); g2 = torch.randn((H // 4) * (W // 4) * C, device=dev); mip1 = torch.empty_like(g1)
hyper = torch.tensor([0.03, 0.9], device=dev)

def spec(stream):
    _lib.check(L.texir_spec_forward(sc.h, _lib.ptr(n), _lib.ptr(a), _lib.ptr(r), _lib.ptr(pts), _lib.ptr(ir), _lib.ptr(camd), _lib.ptr(sh), P, 16, 1e-14, 0,
                                    _lib.ptr(rgb), _lib.ptr(Ls), __import__("ctypes").c_void_p(stream.cuda_stream)))

def adam(stream):
    _lib.check(L.texir_adam_step_tex_dev(_lib.ptr(p), None, None, _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(m), _lib.ptr(v), _lib.ptr(mip1), H, W, C, _lib.ptr(hyper),
                                         0.9, 0.999, 1e-8, 0.0, 1e9, __import__("ctypes").c_void_p(stream.cuda_stream)))

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

print("spec alone      %.1f us" % timeit(lambda: spec(s1)))
print("adam alone      %.1f us" % timeit(lambda: adam(s1)))
print("serial          %.1f us" % timeit(lambda: (spec(s1), adam(s1))))
print("two streams     %.1f us" % timeit(lambda: (spec(s1), adam(s2))))
print("two streams rev %.1f us" % timeit(lambda: (adam(s2), spec(s1))))
for gy in (1024, 256, 128, 85, 64, 43, 32, 21):
    os.environ["TEXIR_ADAM_GRID_Y"] = str(gy)
    print("adam grid.y %4d (%5d blocks): alone %.1f us, with spec on a second stream %.1f us (adam first), %.1f us (spec first)"
          % (gy, 12 * gy, timeit(lambda: adam(s1)), timeit(lambda: (adam(s2), spec(s1))), timeit(lambda: (spec(s1), adam(s2)))))


