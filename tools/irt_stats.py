"""Per-ray traversal statistics of the IrT kernel on a sample of a bench workload (counting build of the product kernel).
usage: python tools/irt_stats.py [workload] [texels]      (env TEXIR_IRT_TEXELS_PER_WAVE=1|64 forces a kernel form)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from texir_code_amd import scene as S, synth, dist_util

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
n_tex = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sc0, pos, nrm, valid, shift, res, spp = bench.make_workload(wl)
dev = torch.device("cuda", 0)
sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
ids = dist_util.morton_order(torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32), res)
start = (ids.numel() // 3) // 4096 * 4096
ids = ids[start:start + n_tex].to(dev)
irr, st = sc.irt_generate(torch.from_numpy(pos).to(dev).reshape(-1, 3), torch.from_numpy(nrm).to(dev).reshape(-1, 3),
                          torch.from_numpy(shift).to(dev), spp, "uniform", texel_ids=ids, stats=True)
rays, nodes, tris, hits, wn, wt = [int(x) for x in st[:6].tolist()]
wr = rays / 64.0
print("workload %s: %d texels x %d spp = %d rays, hit rate %.4f" % (wl, ids.numel(), spp, rays, hits / rays))
print("per ray : %.2f node fetches, %.2f triangle tests" % (nodes / rays, tris / rays))
print("per wave-ray: %.2f node steps, %.2f triangle steps" % (wn / wr, wt / wr))
print("lane utilisation: node step %.3f, triangle step %.3f" % (nodes / (64.0 * wn), tris / (64.0 * max(wt, 1))))
