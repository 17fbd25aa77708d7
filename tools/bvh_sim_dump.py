"""dump a bench workload's arrays as raw binaries for tools/bvh_sim.cpp:  python tools/bvh_sim_dump.py <workload> <outdir>"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from texir_code_amd import synth, dist_util

wl, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
sc0, pos, nrm, valid, shift, res, spp = bench.make_workload(wl)
ids = dist_util.morton_order(torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32), res).numpy()
for name, a, dt in (("verts.f32", sc0["verts"], np.float32), ("tris.i32", sc0["tris"], np.int32), ("tri_uvs.f32", sc0["tri_uvs"], np.float32),
                    ("pos.f32", pos, np.float32), ("nrm.f32", nrm, np.float32), ("shift.f32", shift, np.float32), ("ids.i32", ids, np.int32),
                    ("meta.i32", np.array([spp, res]), np.int32)):
    np.ascontiguousarray(a, dt).tofile(os.path.join(out, name))
print("wrote", out)
