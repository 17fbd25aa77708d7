"""Where does the specular forward of the material step lose its time?  (231 us for 1.57 M rays, bound by nothing measurable.)
Times texir_spec_forward alone on the c4 scene's view 0 (98 304 px x 16 GGX rays):
  warm   back to back (caches hold the BVH from the previous launch)
  cold   after a 2 GB device-to-device copy (what the fused Adam leaves behind in a real step)
  cold + texir_scene_prefetch variants before the launch (serial) -- upper bound of what a prefetch on a parallel graph branch can give
usage: python tools/probes/spec_probe.py [--workload c4]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from texir_code_amd import _lib, scene as S
    sc0, pos, nrm, valid, shift, res, spp = bench.make_workload(a.workload)
    dev = torch.device("cuda", 0)
    sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
    irr = torch.rand(res * res, 3, device=dev) + 0.2
    model, views, data, loss_fn, opt = bench.mat_setup(sc, sc0, irr, res, dev, n_views=2)
    mvp, cam, gt, gmask, seg, fm, room = data[0]
    gb = model._gbuffer(mvp, 0)
    with torch.no_grad():
        preds = model(mvp, 0, cam, 2)
    P = 6 * 128 * 128
    nrm_p = gb["normal"].reshape(P, 3).contiguous()
    pts = (gb["position"] + 1e-2 * gb["normal"]).reshape(P, 3).contiguous()
    alb = preds["albedo"].reshape(P, 3).contiguous()
    rgh = preds["roughness"].reshape(P).contiguous()
    irr_p = torch.rand(P, 3, device=dev)
    sh = torch.rand(P, 2, device=dev)
    rgb = torch.empty(P, 3, device=dev)
    Ls = torch.empty(P, 16, 3, device=dev)
    L = _lib.lib()
    st = _lib.stream_ptr()

    def fwd():
        _lib.check(L.texir_spec_forward(sc.h, _lib.ptr(nrm_p), _lib.ptr(alb), _lib.ptr(rgh), _lib.ptr(pts), _lib.ptr(irr_p), _lib.ptr(cam), _lib.ptr(sh),
                                        P, 16, 1e-14, 0, _lib.ptr(rgb), _lib.ptr(Ls), st))

    big_a = torch.empty(512 << 20, device=dev, dtype=torch.float32)     # 2 GB
    big_b = torch.empty(512 << 20, device=dev, dtype=torch.float32)

    def flush():
        big_b.copy_(big_a)

    def timed(pre):
        ts, tp = [], []
        for _ in range(a.reps):
            flush() if pre is not None else None
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            if pre:
                pre()
            e1.record()
            fwd()
            e2.record()
            torch.cuda.synchronize()
            tp.append(e0.elapsed_time(e1) * 1e3)
            ts.append(e1.elapsed_time(e2) * 1e3)
        return float(np.median(ts)), float(np.median(tp))

    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    print("spec fwd on %s view 0: P = %d, S = 16, roughness mean %.3f" % (a.workload, P, float(rgh.mean())))
    print("  warm (back to back)            %7.1f us" % timed(None)[0])
    print("  cold (after 2 GB copy)         %7.1f us" % timed(lambda: None)[0])
    for what, name in ((1, "quantised nodes"), (3, "both node forms"), (5, "q nodes + triangles"), (7, "nodes + triangles"), (15, "everything incl. uvs")):
        for blocks in (256, 2048):
            t, tpre = timed(lambda: _lib.check(L.texir_scene_prefetch(sc.h, what, blocks, st)))
            print("  cold + prefetch %-22s (%4d blocks) %7.1f us  (+ %5.1f us prefetch)" % (name, blocks, t, tpre))


if __name__ == "__main__":
    main()
