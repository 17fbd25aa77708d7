"""bench.py -- IrT generation throughput (Mrays/s) on MI355X, BASELINE.json's metric.

A "step" is one full pass of the hot path (fused sample + BVH trace + shade + reduce kernel,
texir_irt_generate) over the workload's valid texels at its spp.  Inputs (BVH, radiance texture, texel
G-buffers, shifts) are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|c4|c1|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU: one process per GPU; the compacted valid-texel list is dealt block-cyclically to the ranks
(strong scaling of ONE texture), each rank writes its texels into a zero-initialised full texture and a single
RCCL all_reduce(SUM) assembles it (disjoint support) -- inside the timed region.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes/ray (canonical-BVH2 visit counts measured
by the CPU oracle on a sample of the same rays) x rays per launch / mean kernel time (HIP events).
`cpu_baseline` = the CPU oracle (a port of the reference algorithm; Open3D/Embree is not installable here) timed
on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (triangles, texel res, radiance-texture res, spp)
    "tiny": (20000, 128, 256, 64),
    "c1": (20000, 512, 512, 64),        # configs[0] sizes (reference's CPU-runnable case)
    "c2": (200000, 2048, 2048, 2048),   # configs[1]: IrT 2048 spp, 2k x 2k, 200k-tri mesh, 1 MI355X
    "c4": (1000000, 4096, 4096, 2048),  # configs[3]: 4k x 4k, 1M-tri
}
HBM_PEAK_GBS = 8000.0
BLOCK = 4096  # texels per block of the block-cyclic rank partition


def algorithmic_bytes_per_ray(counters, spp):
    """32*n + 36*t + p_hit*(24 + 48) + (24 + 8 + 1 + 12)/N   (canonical BVH2 of the oracle)"""
    nodes, tris, rays, hits = (float(x) for x in counters)
    return 32.0 * nodes / rays + 36.0 * tris / rays + (hits / rays) * 72.0 + 45.0 / spp, nodes / rays, tris / rays, hits / rays


def cpu_leg(sc0, pos, nrm, valid, shift, spp, budget_s=15.0, timed=True):
    """oracle timed on host cores on a bounded sample; also yields the algorithmic bytes/ray.  timed=False (N > 1: the CPU baseline is
    reported at N = 1 only): just the small counting sample."""
    from oracle import oracle as O
    osc = O.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    vid = np.argwhere(valid.reshape(-1) > 0)[:, 0]
    rng = np.random.default_rng(0)

    def run(n_tex):
        pick = rng.choice(vid, size=min(n_tex, vid.size), replace=False)
        v = np.zeros(valid.size, np.uint8)
        v[pick] = 1
        # compact so the oracle does not scan the whole texture
        c = O.new_counters()
        t0 = time.perf_counter()
        osc.irt_generate(pos.reshape(-1, 3)[pick], nrm.reshape(-1, 3)[pick], None, shift[pick], spp, "uniform", tracer="bvh", counters=c)
        return time.perf_counter() - t0, c, pick.size

    O.set_num_threads(os.cpu_count() or 1)       # all host cores (main() pinned torch's own host ops to one thread)
    cores = O.num_threads()
    run(cores)                                   # warm the thread pool / page in the BVH
    dt, c, n = run(max(cores * 8, 64))           # calibration sample (dynamic schedule needs >> cores texels)
    if not timed:
        return None, c
    rate = n * spp / dt
    n_big = int(max(n, min(vid.size, 0.6 * budget_s * rate / spp)))
    dt, c, n = run(n_big)
    return {"value": n * spp / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d random valid texels x %d spp (%.1f s) of the same workload, canonical BVH2 oracle, OpenMP" % (n, spp, dt)}, c


def mat_leg(sc, sc0, irr_tex, res, dev, rank, world, steps=50, warmup=5):
    """material-estimation step latency (BASELINE.json: "material-step ms at 4k tex"): stage-2 (joint) optimiser step =
    4 texture fetches (4k albedo x3 / 4k roughness x1 / irradiance, mip stacks rebuilt) + GGX-importance specular trace
    (P = 6*128^2 pixels x 16 spp) + fused RenderLoss/SegLoss + backward + (gradient all-reduce) + fused Adam over 67.1 M texels."""
    from texir_code_amd import cameras, conf as C, dist_util, synth
    from texir_code_amd.loss import RenderLoss
    from texir_code_amd.models import MaterialModel
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.trainer.train_material import build_masks
    cube, S, tres = 128, 16, 4096
    conf = C.parse_string("train{ pano_img_res = [%d,%d]\n sample_light = [2048,%d]\n hdr_exposure = 0 }\nmodels{ render{ sample_type = [uniform, importance] } }"
                          % (2 * cube, 4 * cube, S))
    irrt = torch.flip(irr_tex.reshape(res, res, 3), dims=[0]).contiguous()          # file orientation
    model = MaterialModel.from_arrays(sc, sc0["hdr"], irrt, conf, albedo_res=tres, roughness_res=tres)
    views = [cameras.cube_mvps(E) for E in cameras.grid_cameras(4)]
    alb_gt, rgh_gt = synth.make_gt_materials(sc0, tres, tres)
    tri_class = torch.from_numpy(sc0["tri_class"]).to(dev)
    data = []
    with torch.no_grad():
        a0, r0 = model.materials_a.detach().clone(), model.materials_r.detach().clone()
        model.materials_a.copy_(torch.from_numpy(alb_gt))
        model.materials_r.copy_(torch.from_numpy(rgh_gt))
        for i, (mvp, cam) in enumerate(views):
            gt = model(mvp, i, cam, 2)
            tri = model._gbuffer(mvp, i)["tri_id"].long()
            segs = torch.where(tri > 0, tri_class[(tri - 1).clamp(min=0)].long(), torch.zeros_like(tri)).float().unsqueeze(-1)
            m1 = model(mvp, i, cam, -1)
            seg, fm, _ = build_masks(segs, m1["rgb"])
            room = torch.ones((1,) + tuple(seg.shape[1:]), device=dev)
            data.append((mvp, cam.to(dev), gt["rgb"].clone(), gt["empty_mask"].clone(), seg, fm, room))
        model.materials_a.copy_(a0)
        model.materials_r.copy_(r0)
    loss_fn = RenderLoss("L1", 1, lazy_item=True)
    opt = FusedAdam([model.materials_a, model.materials_r], lr=3e-2, fuse_mip_fold=True)
    opt.set_clamp(model.materials_r, 1e-2, 0.8)
    opt.set_clamp(model.materials_a, 0.0, float("inf"))
    if world > 1:
        import torch.distributed as dist

    def fwd_bwd(v):
        mvp, cam, gt, gmask, seg, fm, room = data[v]
        preds = model(mvp, v, cam, 2)
        loss = loss_fn(gt, preds, gmask, fm, seg, stage=2, room_seg_mask=room)[0]
        opt.zero_grad(set_to_none=False)
        loss.backward()
        return loss

    def eager_step(v):
        fwd_bwd(v)
        if world > 1:
            dist_util.reduce_texture_grads([model.materials_a, model.materials_r])
        opt.step()

    # hipGraph capture of the launch-bound part of the step (texir_code_amd/graph_step.py); TEXIR_MAT_GRAPH=0 runs eagerly
    use_graph = os.environ.get("TEXIR_MAT_GRAPH", "1") == "1"
    for p in (model.materials_a, model.materials_r):
        p.grad = torch.zeros_like(p)
    for v in range(len(views)):
        eager_step(v)                       # warm caches
    gs = None
    if use_graph:
        from texir_code_amd.graph_step import GraphedMatStep
        try:
            gs = GraphedMatStep(model, loss_fn, opt, [model.materials_a, model.materials_r])
            for v in range(len(views)):
                mvp, cam, gt, gmask, seg, fm, room = data[v]
                gs.capture(v, mvp, cam, gt, gmask, seg, fm, room, 2)
        except Exception as e:             # capture is an optimisation, not a requirement
            print("material-step graph capture unavailable (%s); running eagerly" % (str(e).splitlines()[0][:120],), file=sys.stderr)
            gs = None
            model._static_shift = None
    graphs = gs is not None
    times = []
    nxt = gs.draw_shift() if gs is not None else None
    for it in range(warmup + steps):
        v = (it * world + rank) % len(views)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if gs is not None:
            gs.step(v, 2, reduce_grads=dist_util.reduce_texture_grads if world > 1 else None, shift=nxt)
            nxt = gs.draw_shift()           # next step's CPU-generator draw overlaps this step's GPU work (same stream order)
        else:
            eager_step(v)
        torch.cuda.synchronize()
        if it >= warmup:
            times.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(times))
    model._static_shift = None
    if world > 1:
        tt = torch.tensor([med], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        med = float(tt.item())
    # N > 1: throughput mode -- every rank renders a different view per optimiser step and the dense texture gradients (335 MB at 4k^2) are
    # all-reduced, so one step covers `world` views: compare ms_per_view across N, not ms
    return {"ms": round(med, 3), "views_per_step": world, "ms_per_view": round(med / world, 3),
            "stage": 2, "steps": steps, "warmup": warmup, "hipgraph": bool(graphs),
            "config": "stage-2 step: albedo %d^2x3 + roughness %d^2x1 (%.1f M params), %d px x %d spp, %d-tri mesh, %d views%s"
                      % (tres, tres, (tres * tres * 4) / 1e6, 6 * cube * cube, S, sc0["T"], len(views), ", view-sharded + grad all_reduce" if world > 1 else "")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-mat", action="store_true")
    args = ap.parse_args()

    # host torch ops on the path are tiny (the per-step CPU-generator draw of 2P floats): one intra-op thread, like the reference's
    # runners (trainer/train_material.py:34).  With the default (= all cores) the draw's OpenMP fork/join jitters by milliseconds.
    torch.set_num_threads(1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(1, torch.cuda.device_count())            # lets a 2-rank smoke test share one GPU (with TEXIR_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("TEXIR_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from texir_code_amd import scene as S, synth, dist_util
    T, res, tex_res, spp = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    sc0 = synth.make_scene(T, seed=666, tex_res=tex_res)
    pos, nrm, valid = synth.make_texel_gbuffer(sc0, res)
    shift = synth.make_shifts(res * res)
    t0 = time.perf_counter()
    sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=local)
    build_s = time.perf_counter() - t0
    d_pos = torch.from_numpy(pos).to(dev).reshape(-1, 3)
    d_nrm = torch.from_numpy(nrm).to(dev).reshape(-1, 3)
    d_shift = torch.from_numpy(shift).to(dev)
    ids_all = torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32)
    if os.environ.get("TEXIR_TEXEL_ORDER", "morton") == "morton":
        ids_all = dist_util.morton_order(ids_all, res)
    ids = dist_util.shard_block_cyclic(ids_all, rank, world, BLOCK).to(dev)
    n_valid = int(ids_all.numel())
    irr = torch.zeros((res * res, 3), device=dev)

    def step():
        irr.zero_()
        sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr)
        if world > 1:
            dist.all_reduce(irr)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        irr.zero_()
        ev[k][0].record()            # HIP events on the stream the kernel is launched on (torch's current stream)
        sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr)
        ev[k][1].record()
        if world > 1:
            dist.all_reduce(irr)
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    mat = None
    if not args.no_mat and args.workload == "c4":
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):          # (constructors print like the reference's; stdout carries the JSON line only)
            mat = mat_leg(sc, sc0, irr, res, dev, rank, world)

    if rank == 0:
        rays_per_step = n_valid * spp
        value = rays_per_step * args.steps / dt / 1e6
        out = {
            "metric": "IrT generation throughput", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: IrT %d spp, %dx%d texels (%d valid), %d-tri synthetic indoor mesh, %dx%d RGB32F radiance texture"
                       % (args.workload, spp, res, res, n_valid, T, tex_res, tex_res),
                       "parallelism": "texel-sharded x%d (block-cyclic %d) + RCCL all_reduce" % (world, BLOCK) if world > 1 else "single GPU",
                       "bvh_build_s": round(build_s, 2), "scene": sc.info()},
        }
        if mat is not None:
            out["material_step"] = mat
        rays_this_rank = int(ids.numel()) * spp
        if not args.no_cpu:
            cpu, counters = cpu_leg(sc0, pos, nrm, valid, shift, spp, timed=world == 1)
            bpr, nbar, tbar, phit = algorithmic_bytes_per_ray(counters, spp)
            from texir_code_amd import _lib
            launches = int(_lib.lib().texir_irt_launch_count(spp))
            achieved = bpr * rays_this_rank / (kern_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.workload)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_step", 0) / launches if "hbm_bytes_per_step" in tj else tj.get("hbm_bytes_per_launch")
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "kernel": "irt_group_kernel<false, 4, 6>", "launches_per_step": launches, "kernel_ms": round(kern_ms / launches, 4),
                               "kernel_ms_per_step": round(kern_ms, 3), "bytes_per_ray": round(bpr, 1),
                               "nodes_per_ray": round(nbar, 2), "tris_per_ray": round(tbar, 2), "p_hit": round(phit, 4),
                               "rays_per_launch": rays_this_rank // launches,
                               # frac > 1 = the canonical algorithm's bytes are served from cache; the measured fabric-side rate is:
                               "traffic_gbs": round(traffic / (kern_ms / launches * 1e-3) / 1e9, 1) if traffic else None,
                               "limiter": "VALU issue in the traversal loop (DESIGN.md section 4)"}
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
