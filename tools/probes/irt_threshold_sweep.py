"""Which kernel form (1 / 64 texels per wave) is fastest for a given texel-list length?  (tunes the launcher's thresholds)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from texir_code_amd import scene as S, synth, dist_util

T, res, tex_res, spp = bench.WORKLOADS["c2"]
sc0 = synth.make_scene(T, seed=666, tex_res=tex_res)
pos, nrm, valid = synth.make_texel_gbuffer(sc0, res)
shift = synth.make_shifts(res * res)
dev = torch.device("cuda", 0)
sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
ids_all = dist_util.morton_order(torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32), res).to(dev)
P, N, SH = torch.from_numpy(pos).to(dev).reshape(-1, 3), torch.from_numpy(nrm).to(dev).reshape(-1, 3), torch.from_numpy(shift).to(dev)
out = torch.zeros((res * res, 3), device=dev)
for n in (1024, 4096, 16384, 65536, 131072, 262144, 524288):
    ids = ids_all[100000:100000 + n].contiguous()
    row = []
    for form in ("1", "64"):
        os.environ["TEXIR_IRT_TEXELS_PER_WAVE"] = form
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sc.irt_generate(P, N, SH, spp, "uniform", texel_ids=ids, out=out)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row.append(n * spp / dt / 1e6)
    print("n_ids %7d  Mrays/s: 1/wave %8.0f   64/wave %8.0f" % (n, row[0], row[1]))
