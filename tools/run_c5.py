"""BASELINE.json configs[4] on one GPU: the joint NIrF -> IrT -> Mat pipeline on a 2M-triangle synthetic mesh, each stage timed.
(The 8-GPU form shards the IrT texels / the views across ranks exactly as bench.py does; this script is the single-GPU walk-through.)
usage: python tools/run_c5.py [--tris 2000000] [--res 4096] [--spp 2048] [--nirf-steps 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from texir_code_amd import conf as C, dist_util, scene as S, synth, tools  # noqa: E402
from texir_code_amd.nirf import IRFLoss, TracerO3dIrrF  # noqa: E402


def rel_l2(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    return float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30))


def checks(a, sc, sc0, m, P, N, vid, d_pos, d_nrm, d_shift, ids, irr, dev):
    """--check: the oracle (test infrastructure) is the checker here, never on the timed path"""
    from oracle import oracle as O
    from texir_code_amd.graph_step import GraphedMatStep
    O.set_num_threads(os.cpu_count() or 1)
    osc = O.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    c = {}
    # NIrF ground truth (TracerO3dIrrF.trace_gt: cosine-weighted irradiance of 256 random surface points at 8 x 16 = 128 spp) vs the oracle
    rng = np.random.default_rng(5)
    sel = np.sort(rng.choice(vid, 256, replace=False))
    p, n = P[sel].to(dev), N[sel].to(dev)
    shift = torch.rand(256, 2, generator=torch.Generator().manual_seed(11))
    gt = m.trace_gt(p, n, [8, 16], shift=shift).cpu().numpy()
    ref = osc.irt_generate(P[sel].numpy(), N[sel].numpy(), None, shift.numpy(), 128, "uniform", tracer="bvh")
    c["nirf_gt_vs_oracle_rel_l2"] = rel_l2(gt, ref)
    # IrT: a random texel sample of the full-size texture vs the oracle; the union of the 8 block-cyclic rank shards vs the whole, bit for bit
    pick = np.sort(rng.choice(vid, 300, replace=False))
    ref = osc.irt_generate(P[pick].numpy(), N[pick].numpy(), None, d_shift[torch.from_numpy(pick).to(dev)].cpu().numpy(), a.spp, "uniform", tracer="bvh")
    c["irt_sample_vs_oracle_rel_l2"] = rel_l2(irr[torch.from_numpy(pick).to(dev)].cpu().numpy(), ref)
    acc = torch.zeros_like(irr)
    st_total = 0
    for r in range(8):
        part = dist_util.shard_block_cyclic(ids, r, 8, bench.BLOCK)
        _, st = sc.irt_generate(d_pos, d_nrm, d_shift, a.spp, "uniform", texel_ids=part, out=acc, stats=True)
        st_total += int(st[0])
    c["irt_shard_union_equals_whole"] = bool(torch.equal(acc, irr))
    c["irt_rays_traced"] = st_total
    del acc
    # material step at 4k textures on this mesh: three steps through hipGraph replay vs the same three steps in eager call order
    cube = 128
    shifts = [torch.rand(6 * cube * cube, 2, generator=torch.Generator().manual_seed(100 + k)) for k in range(3)]
    res = []
    for graph in (False, True):
        torch.manual_seed(3)           # (mat_setup renders its ground truth with GGX shifts from the global CPU generator)
        model, views, data, loss_fn, opt = bench.mat_setup(sc, sc0, irr, a.res, dev, cube=cube, S=16, tres=4096, n_views=2)
        gs = GraphedMatStep(model, loss_fn, opt, [model.materials_a, model.materials_r]) if graph else None
        if gs is not None:
            for i in range(2):
                mvp, cam, gt, gmask, seg, fm, room = data[i]
                gs.capture(i, mvp, cam, gt, gmask, seg, fm, room, 2)
        for k in range(3):
            i = k % 2
            mvp, cam, gt, gmask, seg, fm, room = data[i]
            if gs is not None:
                gs.step(i, 2, shift=shifts[k])
            else:
                model._static_shift = shifts[k].to(dev)
                opt.zero_grad()
                loss_fn(gt, model(mvp, i, cam, 2), gmask, fm, seg, stage=2, room_seg_mask=room)[0].backward()
                opt.step()
        model._static_shift = None
        res.append((model.materials_a.detach().clone(), model.materials_r.detach().clone()))
        del model, opt, gs, data
    c["mat_graph_vs_eager_max_abs"] = float(max((res[0][0] - res[1][0]).abs().max(), (res[0][1] - res[1][1]).abs().max()))
    c["mat_params_moved"] = float((res[0][0] - 0.5).abs().max())
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=2000000)
    ap.add_argument("--res", type=int, default=4096)
    ap.add_argument("--spp", type=int, default=2048)
    ap.add_argument("--nirf-steps", type=int, default=50)
    ap.add_argument("--mat-steps", type=int, default=50)
    ap.add_argument("--check", action="store_true", help="verify every stage (tests/test_gpu_scan_and_configs.py): NIrF ground truth and an IrT texel "
                    "sample against the CPU oracle, IrT 8-shard union == whole, material step hipGraph replay == eager call order")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.set_num_threads(1)                    # as the runners and bench.py (tiny host torch ops; avoids OpenMP fork/join jitter)
    torch.manual_seed(666)
    np.random.seed(666)
    out = {"config": "C5 joint pipeline: %d-tri synthetic mesh, %dx%d texels, %d spp" % (a.tris, a.res, a.res, a.spp)}
    t0 = time.perf_counter()
    sc0 = synth.make_scene(a.tris, seed=666, tex_res=a.res)
    out["scene_gen_s"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
    out["bvh_build_upload_s"] = round(time.perf_counter() - t0, 2)
    out["scene"] = sc.info()

    # 1. NIrF: GT irradiance at random mesh points (IrT kernel) + MLP fit (stock torch)
    cf = C.parse_string("train{ path_mesh_open3d = none\n std_jit = 5e-2 }\nmodels{ irrf_network{ dims = [512,512,512,512]\n p_input_dim = 3\n p_out_dim = 3 } }")
    m = TracerO3dIrrF(cf, scene=sc).cuda()
    opt = torch.optim.Adam(m.ir_radiance_network.parameters(), lr=5e-4)
    lossf = IRFLoss("L1")
    pos, nrm, valid = synth.make_texel_gbuffer(sc0, a.res)
    vid = np.flatnonzero(valid.reshape(-1) > 0)
    P, N = torch.from_numpy(pos.reshape(-1, 3)), torch.from_numpy(nrm.reshape(-1, 3))
    b, losses, gt_ms = 1024, [], []
    for it in range(a.nirf_steps + 5):
        sel = torch.from_numpy(np.random.choice(vid, b, replace=False))
        p, n = P[sel].to(dev), N[sel].to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gt = m.trace_gt(p, n, [8, 16])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = {"gt": gt, "pred": m.ir_radiance_network(p)}
        loss = lossf(res)
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        if it >= 5:
            gt_ms.append((t1 - t0) * 1e3)
            losses.append((time.perf_counter() - t0) * 1e3)
    out["nirf"] = {"batch_points": b, "spp": 128, "gt_trace_ms": round(float(np.median(gt_ms)), 3), "step_ms": round(float(np.median(losses)), 3),
                   "gt_Mrays_s": round(b * 128 / (np.median(gt_ms) * 1e-3) / 1e6, 1), "final_loss": round(float(loss), 4)}

    # 2. IrT
    d_pos, d_nrm = P.to(dev), N.to(dev)
    d_shift = torch.from_numpy(synth.make_shifts(a.res * a.res)).to(dev)
    ids = dist_util.morton_order(torch.from_numpy(vid).to(torch.int32), a.res).to(dev)
    irr = torch.zeros((a.res * a.res, 3), device=dev)
    sc.irt_generate(d_pos, d_nrm, d_shift, 64, "uniform", texel_ids=ids, out=irr)      # warm-up at low spp
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    irr.zero_()
    sc.irt_generate(d_pos, d_nrm, d_shift, a.spp, "uniform", texel_ids=ids, out=irr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["irt"] = {"valid_texels": int(ids.numel()), "seconds": round(dt, 3), "Mrays_s": round(ids.numel() * a.spp / dt / 1e6, 1), "res": a.res, "spp": a.spp}
    if a.check:
        out["checks"] = checks(a, sc, sc0, m, P, N, vid, d_pos, d_nrm, d_shift, ids, irr, dev)

    # 3. the asset step in between (tools/padding_texture.py) + material step
    t0 = time.perf_counter()
    padded = tools.padding_texture(irr.reshape(a.res, a.res, 3).cpu().numpy())
    out["padding_s"] = round(time.perf_counter() - t0, 2)
    mat = bench.mat_leg(sc, sc0, torch.from_numpy(padded).to(dev).reshape(-1, 3), a.res, dev, 0, 1, steps=a.mat_steps)
    out["material_step"] = mat
    print(json.dumps(out))


if __name__ == "__main__":
    main()
