"""bench.py -- IrT generation throughput (Mrays/s) on MI355X, BASELINE.json's metric.

A "step" is one full pass of the hot path (fused sample + BVH trace + shade + reduce kernel,
texir_irt_generate) over the workload's valid texels at its spp.  Inputs (BVH, radiance texture, texel
G-buffers, shifts) are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W] [--workload c4|c4_scan|c2|c1|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N ranks
(one per GPU, backend nccl = RCCL); it refuses loudly when the box has fewer than N devices.

Multi-GPU: one process per GPU; the compacted valid-texel list is dealt block-cyclically to the ranks
(strong scaling of ONE texture), each rank writes its texels into a zero-initialised full texture and a single
RCCL all_gather of the ranks' compacted values assembles it (disjoint support: the same bits as an all_reduce(SUM) at a twelfth of the bytes) -- inside the timed region.

Prints ONE compact JSON line (rank 0; <= 3 KB by construction, `compact_line`) and writes the whole record -- every limit with its numerator and denominator, the
occupancy sweep, the per-shard projection, the sibling workloads, the stages end to end -- to gpurun_out/bench_full.json ($TEXIR_BENCH_FULL overrides; `"full"` in
the line names it).  In the full record `roofline` carries the bounds that hold (all <= 1), each = a per-launch counter of the
dominant kernel (rocprofv3 --pmc, committed under profiles/pmc_<workload>.json together with a hash of the kernel sources
-- a profile of other sources is refused) divided by the kernel time measured live with HIP events:
  * memory: fabric-side bytes (L2 <-> Infinity Cache/HBM read requests x 128 B + writes) / time / 8 TB/s,
  * VALU issue: SQ_INSTS_VALU / (1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction x time),
  * vector L1: TCP_TOTAL_CACHE_ACCESSES / (the 256 TCPs' clocks over the same time) -- a TCP serves one access per clock (self-calibrated).
The top-level `bound` / `achieved` / `peak` / `frac` / `traffic` are always the memory side (frac = achieved / peak); `binding` names the
tightest of the three, `limits` carries each one's numerator, denominator and fraction.  SURVEY.md 8(d)'s algorithmic bytes (canonical BVH2 visit counts x 32/36 B) are reported under
`algorithmic` -- they are served by L1/L2 hits of a 4x more compact tree and exceed the HBM peak, so they bound nothing.
`cpu_baseline` = the CPU oracle (a port of the reference algorithm; Open3D/Embree is not installable here) timed
on this box's host cores on a bounded sample of the same workload.
In the printed line: roofline.{bound, achieved, peak, frac, traffic} = the memory side; binding / binding_frac = the largest measured limit; algorithmic_frac = SURVEY 8(d)'s
bytes / the live kernel time / 8 TB/s (> 1: cache-served, not a bound); profile_matches_live = the traversal's own counters on a fixed slice of the workload, recomputed by
this run, equal the ones stored beside the PMC counters (the counters come from the profile session's box, the time from this one).
"""
import argparse
import hashlib
import json
import math
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (triangles, texel res, radiance-texture res, spp, scene style)
    "tiny": (20000, 128, 256, 64, "room"),
    "c1": (20000, 512, 512, 64, "room"),        # configs[0] sizes (reference's CPU-runnable case)
    "c2": (200000, 2048, 2048, 2048, "room"),   # configs[1]: IrT 2048 spp, 2k x 2k, 200k-tri mesh, 1 MI355X
    "c4": (1000000, 4096, 4096, 2048, "room"),  # configs[3]: 4k x 4k, 1M-tri (the configuration the metric is quoted on)
    # hostile sibling of c4 (never the headline): rotated clutter, thin slats, two openings (p_hit < 1), noisy "scanned" surfaces
    "c4_scan": (1000000, 4096, 4096, 2048, "scan"),
    "tiny_scan": (20000, 128, 256, 64, "scan"),
    # third family (never the headline): the shape of the reference's data -- a 3 x 3-room house, rooms joined by doors, large untessellated shell triangles
    # next to millimetre-scale clutter, windows (p_hit < 1)  (synth._house; /root/reference README.md:21-34)
    "house": (1000000, 4096, 4096, 2048, "house"),
    "tiny_house": (20000, 128, 256, 64, "house"),
    # experiment: c4 with a 1k^2 radiance texture (12.6 MB: L2/Infinity-Cache resident) -- how much of c4's time is texture traffic
    "c4_tex1k": (1000000, 4096, 1024, 2048, "room"),
}
HBM_PEAK_GBS = 8000.0
SIMDS, CLOCK_HZ = 1024, 2.4e9                  # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles
BLOCK = 4096  # texels per block of the block-cyclic rank partition
KERNEL_SOURCES = ["kernels.hip", "device_common.h", "kernels.h", "bvh_build.cpp", "bvh_build.h", "capi.hip", "env.h", "env.cpp", "Makefile"]


def kernel_src_sha():
    """hash of the sources the dominant kernel is built from: a PMC profile only describes the library it was taken with"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "texir_code_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def make_workload(name, cache=None):
    """(sc0, pos, nrm, valid, shift, res, spp) of a named workload; seed 666 everywhere.  `cache` (or
    $TEXIR_SYNTH_CACHE) names a directory where the generated arrays are kept between processes of one session."""
    from texir_code_amd import synth
    T, res, tex_res, spp, style = WORKLOADS[name]
    cache = cache or os.environ.get("TEXIR_SYNTH_CACHE")
    path = os.path.join(cache, "%s%s.npz" % (name, "_f32tex" if os.environ.get("TEXIR_BENCH_FLOAT_TEX", "0") == "1" else "")) if cache else None
    if path and os.path.exists(path):
        z = np.load(path)
        sc0 = {k: z[k] for k in ("verts", "tris", "tri_uvs", "hdr", "tri_class")}
        sc0.update({"T": T, "style": style, "seed": 666})
        if "patch_rects" in z:               # the charts' atlas rectangles: all the material leg needs of the patch list (synth.make_gt_materials)
            import types
            sc0["patches"] = [types.SimpleNamespace(rect=tuple(float(x) for x in r)) for r in z["patch_rects"]]
        return sc0, z["pos"], z["nrm"], z["valid"], z["shift"], res, spp
    sc0 = synth.make_scene(T, seed=666, tex_res=tex_res, style=style)
    if os.environ.get("TEXIR_BENCH_FLOAT_TEX", "0") != "1":
        # the texture as the reference's pipeline holds it: an RGBE file (hdr_texture.hdr) times 2^hdr_exposure (tracer_o3d_irt.py:77-81; configs/*.conf: 5);
        # TEXIR_BENCH_FLOAT_TEX=1 keeps the generator's float-valued texels (rounds 1-5), which the hit shader then reads as float32 tiles
        sc0["hdr"] = synth.rgbe_born(sc0["hdr"], 5.0)
    pos, nrm, valid = synth.make_texel_gbuffer(sc0, res)
    shift = synth.make_shifts(res * res)
    if path:
        os.makedirs(cache, exist_ok=True)
        tmp = path + ".tmp%d.npz" % os.getpid()
        np.savez(tmp, pos=pos, nrm=nrm, valid=valid, shift=shift, patch_rects=np.array([p.rect for p in sc0["patches"]], np.float64),
                 **{k: sc0[k] for k in ("verts", "tris", "tri_uvs", "hdr", "tri_class")})
        os.replace(tmp, path)
    return sc0, pos, nrm, valid, shift, res, spp


def irt_plan_parts(spp):
    """parts per texel of the 64-texels-per-wave IrT form (kernels.hip irt_plan: a function of N alone; up to 32 parts of at least 8 passes)"""
    if spp & (spp - 1):
        return 1
    lp = 0
    while lp < 5 and (spp >> (lp + 1)) >= 8:
        lp += 1
    return 1 << lp


def algorithmic_bytes_per_ray(counters, spp):
    """32*n + 36*t + p_hit*(24 + 48) + (24 + 8 + 1 + 12)/N   (canonical BVH2 of the oracle)"""
    nodes, tris, rays, hits = (float(x) for x in counters)
    return 32.0 * nodes / rays + 36.0 * tris / rays + (hits / rays) * 72.0 + 45.0 / spp, nodes / rays, tris / rays, hits / rays


def host_cpus():
    """how many host CPUs this process may really use: the scheduler affinity mask, cut by a cgroup CPU quota if one is set"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    return {"os_cpu_count": os.cpu_count(), "affinity": aff, "cgroup_cpus": None if quota is None else round(quota, 2),
            "usable": max(1, min(aff, int(math.ceil(quota)) if quota else aff))}


def open3d_leg(sc0, pos, nrm, valid, shift, spp, budget_s):
    """BASELINE.md 2's PRIMARY baseline: the reference's own tracer -- Open3D's RaycastingScene (Embree) called in the reference's shape,
    rays [512, N, 1, 6] per batch (models/tracer_o3d_irt.py:156-173, 242-244).  Returns None when open3d cannot be imported (it is not in this
    image: the port below is what runs; the attempt is made so that a box that has it reports kind = "reference")."""
    try:
        import open3d as o3d
    except Exception:
        return None
    from oracle import oracle as O
    mesh = o3d.t.geometry.TriangleMesh()
    mesh.vertex.positions = o3d.core.Tensor(np.ascontiguousarray(sc0["verts"], np.float32))
    mesh.triangle.indices = o3d.core.Tensor(np.ascontiguousarray(sc0["tris"], np.uint32))
    scene = o3d.t.geometry.RaycastingScene()
    scene.add_triangles(mesh)
    vid = np.argwhere(valid.reshape(-1) > 0)[:, 0]
    rng = np.random.default_rng(0)
    rays_done, t_used = 0, 0.0
    while t_used < 0.6 * budget_s:
        pick = rng.choice(vid, size=512, replace=False)
        dirs = O.generate_dir(nrm.reshape(-1, 3)[pick], spp, "uniform", shift[pick])                       # [512, N, 3]
        org = np.broadcast_to(pos.reshape(-1, 3)[pick][:, None, :], dirs.shape)
        rays = np.ascontiguousarray(np.concatenate([org, dirs], -1)[:, :, None, :], np.float32)            # [512, N, 1, 6]
        t0 = time.perf_counter()
        scene.cast_rays(o3d.core.Tensor(rays))
        t_used += time.perf_counter() - t0
        rays_done += 512 * spp
    hc = host_cpus()
    return {"value": rays_done / t_used / 1e6, "unit": "Mrays/s", "cores": hc["usable"], "kind": "reference", "host": hc,
            "sample": "%d rays in batches of [512, %d, 1, 6] through open3d.t.geometry.RaycastingScene.cast_rays (%.1f s), intersection only" % (rays_done, spp, t_used)}


def cpu_leg(sc0, pos, nrm, valid, shift, spp, budget_s=15.0, timed=True):
    """CPU baseline on the host cores, on a bounded sample of the same workload; also yields the algorithmic bytes/ray.  timed=False (N > 1: the CPU
    baseline is reported at N = 1 only): just the small counting sample.  The reference's own tracer is tried first (open3d_leg); what runs in
    this image is the port: the C oracle on its canonical BVH2, OpenMP over texels (schedule(dynamic, 1)), with the one-thread rate beside the
    all-thread rate and the CPU count the process may really use (affinity mask and cgroup quota), so that the line can be judged as a baseline."""
    from oracle import oracle as O
    osc = O.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"])
    vid = np.argwhere(valid.reshape(-1) > 0)[:, 0]
    rng = np.random.default_rng(0)

    def run(n_tex):
        pick = rng.choice(vid, size=min(n_tex, vid.size), replace=False)
        # compact so the oracle does not scan the whole texture
        c = O.new_counters()
        t0 = time.perf_counter()
        osc.irt_generate(pos.reshape(-1, 3)[pick], nrm.reshape(-1, 3)[pick], None, shift[pick], spp, "uniform", tracer="bvh", counters=c)
        return time.perf_counter() - t0, c, pick.size

    hc = host_cpus()
    threads = hc["usable"]
    O.set_num_threads(threads)                   # (main() pinned torch's own host ops to one thread, which lowers OpenMP's default)
    run(threads)                                 # warm the thread pool / page in the BVH
    dt, c, n = run(max(threads * 16, 64))        # calibration sample: >= 16 texels per thread for the dynamic schedule
    if not timed:
        return None, c
    ref = open3d_leg(sc0, pos, nrm, valid, shift, spp, budget_s)
    if ref is not None:
        return ref, c
    rate = n * spp / dt
    # one thread, ~2 s: the per-core rate the all-thread figure should be a multiple of
    O.set_num_threads(1)
    n1 = max(4, int(2.0 * (rate / threads) / spp))
    dt1, _, n1 = run(n1)
    one = n1 * spp / dt1 / 1e6
    O.set_num_threads(threads)
    n_big = int(max(n, threads * 16, min(vid.size, 0.6 * budget_s * rate / spp)))
    dt, c, n = run(n_big)
    val = n * spp / dt / 1e6
    return {"value": val, "unit": "Mrays/s", "cores": threads, "kind": "port", "one_thread": round(one, 3), "per_thread": round(val / threads, 3),
            "parallel_efficiency": round(val / (one * threads), 3), "host": hc, "omp_threads": O.num_threads(),
            "sample": "%d random valid texels x %d spp (%.1f s) of the same workload, canonical BVH2 oracle, OpenMP schedule(dynamic, 1) over texels; "
                      "one thread: %d texels (%.1f s)" % (n, spp, dt, n1, dt1)}, c


def mat_setup(sc, sc0, irr_tex, res, dev, cube=128, S=16, tres=4096, n_views=16, fuse=True):
    """the material-step problem of BASELINE.json ("material-step ms at 4k tex") on a synthetic scene: MaterialModel with tres^2 albedo /
    roughness textures, per-view data (ground truth rendered from the synthetic GT materials, segmentation / highlight masks built as the
    trainer builds them), the fused loss and optimiser.  Shared by mat_leg, tools/run_c5.py and the 4k-texture tests."""
    from texir_code_amd import cameras, conf as C, synth
    from texir_code_amd.loss import RenderLoss
    from texir_code_amd.models import MaterialModel
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.trainer.train_material import build_masks
    conf = C.parse_string("train{ pano_img_res = [%d,%d]\n sample_light = [2048,%d]\n hdr_exposure = 0 }\nmodels{ render{ sample_type = [uniform, importance] } }"
                          % (2 * cube, 4 * cube, S))
    irrt = torch.flip(irr_tex.reshape(res, res, 3), dims=[0]).contiguous()          # file orientation
    model = MaterialModel.from_arrays(sc, sc0["hdr"], irrt, conf, albedo_res=tres, roughness_res=tres)
    model.lean_outputs = True               # (as the runner sets it: the un-mipmapped roughness fetch only feeds the stage-1 loss)
    views = [cameras.cube_mvps(E) for E in cameras.grid_cameras(4)][:n_views]
    sc0 = dict(sc0)
    if "patches" not in sc0:                 # (workload came from the array cache: the chart list is needed for the GT materials)
        sc0["patches"] = synth.make_scene(sc0["T"], seed=666, tex_res=8, style=sc0.get("style", "room"))["patches"]
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
    loss_fn = RenderLoss("L1", 1, lazy_item=True, unit_upstream=True)       # (as the runner's hipGraph path sets it)
    opt = FusedAdam([model.materials_a, model.materials_r], lr=3e-2, fuse_mip_fold=fuse)
    opt.set_clamp(model.materials_r, 1e-2, 0.8)
    opt.set_clamp(model.materials_a, 0.0, float("inf"))
    return model, views, data, loss_fn, opt


def mat_leg(sc, sc0, irr_tex, res, dev, rank, world, steps=50, warmup=5, cube=128, tres=4096):
    """material-estimation step latency (BASELINE.json: "material-step ms at 4k tex"): stage-2 (joint) optimiser step =
    4 texture fetches (4k albedo x3 / 4k roughness x1 / irradiance, mip stacks rebuilt) + GGX-importance specular trace
    (P = 6*128^2 pixels x 16 spp) + fused RenderLoss/SegLoss + backward + (gradient all-reduce) + fused Adam over 67.1 M texels."""
    from texir_code_amd import dist_util
    S = 16                 # (cube = 128, tres = 4096: BASELINE.json's "material-step ms at 4k tex"; --mat-res / --mat-cube shrink it for the plumbing tests)
    model, views, data, loss_fn, opt = mat_setup(sc, sc0, irr_tex, res, dev, cube, S, tres)
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
    view_of = lambda it: (it * world + rank) % len(views)
    if gs is not None:
        gs.stage_shift(view_of(0), gs.draw_shift())
    dist_util.comm_reset()
    for it in range(warmup + steps):
        v = view_of(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if gs is not None:
            gs.step(v, 2, reduce_grads=dist_util.reduce_texture_grads if world > 1 else None, staged=True)
            # the next step's CPU-generator draw (same stream order as the reference's) and its placement overlap this step's GPU work
            gs.stage_shift(view_of(it + 1), gs.draw_shift())
        else:
            eager_step(v)
        torch.cuda.synchronize()
        if it >= warmup:
            times.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(times))
    comm_bytes = dist_util.COMM["bytes"] / float(warmup + steps)
    # The same steps queued back to back, as the trainer's loop issues them (no host synchronisation per step: the next step's shifts are drawn and
    # staged while the current one runs): total time / steps.  `ms` above is the LATENCY of one step (launch + run + the host noticing the end), the
    # figure of the earlier rounds; this is the step PERIOD of a running optimisation.
    b2b = host_ms = None
    if gs is not None and world == 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(warmup + steps, warmup + 2 * steps):
            gs.step(view_of(it), 2, staged=True)
            gs.stage_shift(view_of(it + 1), gs.draw_shift())
        host_ms = (time.perf_counter() - t0) * 1e3 / steps           # what the host needs per step (it must stay below the period)
        torch.cuda.synchronize()
        b2b = (time.perf_counter() - t0) * 1e3 / steps
    model._static_shift = None
    if world > 1:
        tt = torch.tensor([med], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        med = float(tt.item())
    # algorithmic HBM bytes of one step: fused Adam 28 B/param (the level-1 gradient it folds in is another 1 B/param),
    # mip build of both trainable textures (read level 0, write 1/3), gradient stacks of the touched mip levels, G-buffer + loss streams
    n_par = tres * tres * 4
    P = 6 * cube * cube
    step_bytes = 28.0 * n_par + 4.0 * n_par * (1.0 + 1.0 / 3.0) + 4.0 * n_par / 4.0 + P * (24 + 8 + 16 + 12 + 2 + 12 + 3 * 16 * 12.0)
    # N > 1: throughput mode -- every rank renders a different view per optimiser step and the dense texture gradients (335 MB at 4k^2) are
    # all-reduced, so one step covers `world` views: compare ms_per_view across N, not ms
    # measured fabric traffic of one replayed step (tools/mat_step_pmc.sh -> profiles/pmc_mat_step.json; refused when taken with other sources)
    traffic = tnote = None
    pm = os.path.join(ROOT, "profiles", "pmc_mat_step.json")
    if os.path.exists(pm):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from mat_step_pmc import mat_src_sha
        tj = json.load(open(pm))
        if tj.get("mat_src_sha") == mat_src_sha():
            traffic = tj["fabric_bytes_per_step"]
            tnote = "measured: %.2f GB read + %.2f GB written per step over the fabric (profiles/pmc_mat_step.json)" % (tj["read_bytes_per_step"] / 1e9, tj["write_bytes_per_step"] / 1e9)
        else:
            tnote = "profiles/pmc_mat_step.json was taken with other sources"
    return {"ms": round(med, 3), "ms_back_to_back": None if b2b is None else round(b2b, 3), "host_ms_per_step": None if host_ms is None else round(host_ms, 3),
            "views_per_step": world, "ms_per_view": round(med / world, 3), "mat_shard": "view" if world > 1 else None,
            "collective_bytes_per_step": int(comm_bytes) if world > 1 else 0,
            "stage": 2, "steps": steps, "warmup": warmup, "hipgraph": bool(graphs),
            "roofline": {"bound": "hbm", "achieved": round(step_bytes / (med * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(step_bytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_step": int(step_bytes),
                         "traffic": traffic, "traffic_note": tnote,
                         "note": "algorithmic bytes: Adam 28 B x %.1f M params + mip build + level-1 gradient + per-pixel streams; the %d x %d specular "
                                 "rays are cache-served BVH traffic and are not counted" % (n_par / 1e6, P, S)},
            "config": "stage-2 step: albedo %d^2x3 + roughness %d^2x1 (%.1f M params), %d px x %d spp, %d-tri mesh, %d views%s"
                      % (tres, tres, (tres * tres * 4) / 1e6, 6 * cube * cube, S, sc0["T"], len(views), ", view-sharded + grad all_reduce" if world > 1 else "")}


def mat_leg_pixel(sc, sc0, irr_tex, res, dev, rank, world, steps=50, warmup=5, cube=128, tres=4096):
    """N > 1, train.mat_shard = pixel (the trainer's default, SURVEY.md 8e(i) parity mode): ONE view per step, its pixels split across the ranks for the
    specular trace and its backward, the texture side replicated (sharded_step.ShardedMatStep): two all_gathers of per-pixel data per step, nothing reduced.
    The trajectory is the single-GPU one bit for bit, so the step's loss must be equal on every rank: checked here on the timed steps."""
    import torch.distributed as dist
    from texir_code_amd import dist_util
    from texir_code_amd.sharded_step import ShardedMatStep
    S = 16
    torch.manual_seed(666)                 # the ground-truth views are rendered with shifts from the CPU generator: the same stream on every rank
    model, views, data, loss_fn, opt = mat_setup(sc, sc0, irr_tex, res, dev, cube, S, tres)
    for p in (model.materials_a, model.materials_r):
        p.grad = None
    ss = ShardedMatStep(model, loss_fn, opt, [model.materials_a, model.materials_r], use_graph=os.environ.get("TEXIR_MAT_GRAPH", "1") == "1")
    for v in range(len(views)):
        mvp, cam, gt, gmask, seg, fm, room = data[v]
        ss.capture(v, mvp, cam, gt, gmask, seg, fm, room, 2)
    torch.manual_seed(666)                 # every rank draws the same full-view shifts from the CPU generator, like the single-GPU run
    dist_util.comm_reset()
    times, losses = [], []
    for it in range(warmup + steps):
        v = it % len(views)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = ss.step(v, 2)
        torch.cuda.synchronize()
        if it >= warmup:
            times.append((time.perf_counter() - t0) * 1e3)
            losses.append(loss.detach().reshape(1).clone())
    comm_bytes = dist_util.COMM["bytes"] / float(warmup + steps)
    med = float(np.median(times))
    tt = torch.tensor([med], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    mine = torch.cat(losses)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    agree = all(bool(torch.equal(every[0], e)) for e in every[1:])
    spread = float(max((every[0] - e).abs().max().item() for e in every[1:])) if world > 1 else 0.0
    # back to back (no host synchronisation per step), as the trainer with train.log_lag > 0 queues them
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for it in range(steps):
        ss.step(it % len(views), 2)
    torch.cuda.synchronize()
    b2b = torch.tensor([(time.perf_counter() - t0) * 1e3 / steps], device=dev, dtype=torch.float64)
    dist.all_reduce(b2b, op=dist.ReduceOp.MAX)
    return {"ms": round(float(tt.item()), 3), "ms_back_to_back": round(float(b2b.item()), 3), "views_per_step": 1, "mat_shard": "pixel",
            "collective_bytes_per_step": int(comm_bytes), "collectives_per_step": 2, "ranks_agree_on_every_loss": agree, "max_loss_spread_over_ranks": spread,
            "first_losses_rank0": [float(x) for x in every[0][:4].tolist()],
            "stage": 2, "steps": steps, "warmup": warmup, "hipgraph": "graphs" in next(iter(ss.views.values())),
            "config": "stage-2 step: albedo %d^2x3 + roughness %d^2x1, %d px x %d spp split over %d ranks (pixel slices), texture side replicated"
                      % (tres, tres, 6 * cube * cube, S, world)}


FINGERPRINT_TEXELS = 262144


def irt_fingerprint(sc, d_pos, d_nrm, d_shift, ids_all, spp, dev):
    """what the traversal DID on a fixed slice of the workload (the counting form of the kernel: rays, per-lane node fetches, triangle tests, hits, wave-level node /
    triangle steps over 262 144 listed texels from the middle third of the Morton list): integers, deterministic for given sources + workload.  The committed PMC profile carries
    the same six numbers from the box it was taken on (tools/pmc_to_json.py); equal numbers = the kernel this run timed walks the scene exactly as the profiled one did
    (the line's counters come from another box -- this is the live cross-check of that hybrid)"""
    n = int(ids_all.numel())
    first = ((n // 3) // 4096) * 4096
    ids = ids_all[first:first + min(FINGERPRINT_TEXELS, n - first)].to(dev)
    _, st = sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, stats=True)
    v = [int(x) for x in st[:6].tolist()]
    return dict(zip(("rays", "node_fetches", "tri_tests", "hits", "wave_node_steps", "wave_tri_steps"), v))


def load_pmc(workload, kernel):
    """per-launch PMC counters of the dominant kernel for this workload (written by tools/profile_round.sh); None -- with the
    reason -- when there is no profile of THESE kernel sources / this kernel form"""
    path = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)
    if not os.path.exists(path):
        return None, "no profiles/pmc_%s.json" % workload
    tj = json.load(open(path))
    if tj.get("kernel_src_sha") != kernel_src_sha():
        return None, "profiles/pmc_%s.json was taken with other kernel sources (%s != %s)" % (workload, tj.get("kernel_src_sha"), kernel_src_sha())
    if kernel not in tj.get("kernel", ""):
        return None, "profiles/pmc_%s.json describes %s, this run launched %s" % (workload, tj.get("kernel"), kernel)
    return tj, None


def load_chain(workload, kernel):
    """the chain probe of this workload (tools/chain_probe.py), refused when taken with other kernel sources"""
    path = os.path.join(ROOT, "profiles", "chain_%s.json" % workload)
    if not os.path.exists(path):
        return None, "no profiles/chain_%s.json" % workload
    tj = json.load(open(path))
    if tj.get("kernel_src_sha") != kernel_src_sha():
        return None, "profiles/chain_%s.json was taken with other kernel sources" % workload
    if kernel not in tj.get("kernel", ""):
        return None, "profiles/chain_%s.json describes %s" % (workload, tj.get("kernel"))
    return tj, None


def roofline(workload, kernel, kern_ms, rays_this_rank, world, alg, fingerprint=None):
    """the two bounds of the module docstring; `alg` = (bytes/ray, nodes/ray, tris/ray, p_hit) of SURVEY 8(d) or None; `fingerprint` = this run's irt_fingerprint"""
    t = kern_ms * 1e-3
    out = {"bound": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
           "kernel": kernel, "kernel_ms": round(kern_ms, 3), "rays_per_launch": rays_this_rank}
    pmc, why = load_pmc(workload, kernel)
    if pmc is None:
        out["note"] = "no measured bound: " + why
    else:
        # counters are per launch of the N = 1 profile; a rank of an N-GPU run traces 1/N of the rays with the same per-ray cost
        scale = rays_this_rank / float(pmc.get("rays_per_launch", rays_this_rank))
        traffic = float(pmc["fabric_bytes_per_launch"]) * scale
        valu = float(pmc["SQ_INSTS_VALU"]) * scale
        if world > 1:
            out["note"] = "per-launch counters of the N = 1 profile scaled by this rank's share of the rays (%.4f)" % scale
        mem_frac = traffic / t / 1e9 / HBM_PEAK_GBS
        valu_frac = valu / (SIMDS * CLOCK_HZ / 2.0 * t)
        # vector-L1 (TCP): the unit serves ONE cache access per clock (tools/probes/tcp_rate.hip under the same counters: 1.00 per clock for
        # contiguous and for scattered wave-wide dwordx4 loads alike); the kernel's accesses over the TCP clocks of this run's duration
        l1 = None
        if pmc.get("tcp_cache_accesses") and pmc.get("tcp_clocks") and pmc.get("kernel_ms_under_pmc"):
            tcp_hz = float(pmc["tcp_clocks"]) / (float(np.mean(pmc["kernel_ms_under_pmc"])) * 1e-3)          # all 256 TCPs together
            l1 = float(pmc["tcp_cache_accesses"]) * scale / (tcp_hz * t)
        fr = {"hbm": mem_frac, "valu": valu_frac}
        if l1 is not None:
            fr["l1"] = l1
        # The top-level triple is ALWAYS the memory side (achieved / peak = frac, recomputable from the line itself); which of the three
        # measured limits is the tightest is named under `binding`, each limit with its own numerator and denominator under `limits`.
        out.update({"bound": "hbm", "achieved": round(traffic / t / 1e9, 1), "frac": round(mem_frac, 4), "traffic": traffic,
                    "binding": max(fr, key=fr.get),
                    "limits": {
                        "hbm": {"frac": round(mem_frac, 4), "achieved": round(traffic / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "fabric_bytes_per_launch": traffic, "bytes_per_ray": round(traffic / rays_this_rank, 1), "l2_hit_rate": pmc.get("l2_hit_rate"),
                                "note": "L2 <-> Infinity Cache / HBM: TCC_EA0_RDREQ x request size + WRITE_SIZE (MI355X_MICROARCH.md HBM section), / kernel time / 8 TB/s"},
                        "valu": {"frac": round(valu_frac, 4), "achieved": round(valu / t / 1e9, 2), "peak": round(SIMDS * CLOCK_HZ / 2.0 / 1e9, 2), "unit": "G wave-instructions/s",
                                 "insts_per_launch": valu, "insts_per_64_rays": round(valu / (rays_this_rank / 64.0), 1), "lane_utilisation": pmc.get("valu_lane_utilisation"),
                                 "note": "SQ_INSTS_VALU / t against 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction (raw count: conversions, compares, selects, min / max and shifts "
                                         "-- two thirds of the node step -- take ~1.65 of an fma's issue time at 8 waves per SIMD, profiles/r02/issue_rate.txt)"},
                        "l1": None if l1 is None else {"frac": round(l1, 4), "achieved": round(float(pmc["tcp_cache_accesses"]) * scale / t / 1e9, 2),
                                                       "peak": round(tcp_hz / 1e9, 2), "unit": "G cache accesses/s", "accesses_per_launch": float(pmc["tcp_cache_accesses"]) * scale,
                                                       "note": "vector-L1 (TCP) tag lookups; the peak -- one access per clock per TCP, 256 TCPs at the clock TCP_GATE_EN1 reports -- is "
                                                               "SELF-CALIBRATED (tools/probes/tcp_rate.hip on this hardware, profiles/r02/tcp_rate.txt), not a documented figure; a build with a quarter fewer "
                                                               "node-fetch loads (48-byte nodes, profiles/r03/ab_node48_s18.txt) was not faster: read the fraction as how busy the L1 is, not as proof that it binds"}},
                    "waves": {"wait_any_frac": pmc.get("sq_wait_any_frac"), "active_inst_any_frac": pmc.get("sq_active_inst_any_frac"),
                              "note": "share of the resident waves' cycles spent waiting / issuing (SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES)"},
                    "scalar_path": None if pmc.get("smem_insts") is None else {"smem_insts_per_launch": float(pmc["smem_insts"]) * scale,
                                                                               "scalar_cache_hit_rate": pmc.get("scalar_cache_hit_rate")},
                    "profile": "profiles/pmc_%s.json (%s)" % (workload, pmc.get("source", ""))})
        if fingerprint is not None:
            # the counters above were collected on another box: do the two runs walk the scene identically?
            out["live_fingerprint"] = fingerprint
            out["profile_matches_live"] = (pmc["fingerprint"] == fingerprint) if pmc.get("fingerprint") else None
    chain, why_c = load_chain(workload, kernel)
    if chain is not None and out.get("limits") is not None:
        # the dependent-chain bound (tools/chain_probe.py -> profiles/chain_<workload>.json): every wave-level step of the launch at the latency it has with ONE wave
        # per SIMD, all resident waves overlapping perfectly; and what the rate does when waves are taken away
        pts = chain["occupancy_sweep"]["points"]
        t_chain = float(chain["chain_bound"]["seconds"]) * (rays_this_rank / float(chain["rays_per_launch"]))
        out["limits"]["chain"] = {"frac": round(t_chain / t, 4), "achieved": round(t_chain, 7), "peak": round(t, 7), "unit": "s (chain time over kernel time)",
                                  "cycles_per_step_at_1_wave": pts[0]["cycles_per_step"], "cycles_per_step_at_8_waves": pts[-1]["cycles_per_step"],
                                  "steps_per_pass": chain["full_occupancy"]["per_step"]["steps_per_pass"], "probe_overhead": chain["full_occupancy"]["probe_overhead"],
                                  "note": "sum over step kinds of (wave-level steps of the launch) x (cycles per step measured with ONE wave per SIMD) / 8192 resident waves / 2.4 GHz, over the "
                                          "live kernel time: the share of the launch that the dependent fetch -> test -> sort -> push chain explains if nothing were shared; the rest is "
                                          "waves queueing for their SIMD's issue slots and L1"}
        out["occupancy"] = {"waves_per_simd": [p["waves_per_simd"] for p in pts], "grays_per_s": [p["grays_per_s"] for p in pts], "spp": chain["occupancy_sweep"]["spp"],
                            "rate_8_over_1": chain["reading"]["rate_8_waves_over_1_wave"],
                            "note": "shipped kernel, persistent grid capped (TEXIR_IRT_GRID_CAP): 8x the waves buy %.2fx the rate -- the SIMD-shared resources (issue slots, L1) saturate; "
                                    "a pure latency chain would scale 8x" % chain["reading"]["rate_8_waves_over_1_wave"]}
        fr = {k: v["frac"] for k, v in out["limits"].items() if v is not None and v.get("frac") is not None}
        out["binding"] = max(fr, key=fr.get)
        out["profile_chain"] = "profiles/chain_%s.json" % workload
    elif out.get("limits") is not None:
        out["limits"]["chain"] = None
        out["chain_note"] = why_c
    if alg is not None:
        bpr, nbar, tbar, phit = alg
        out["algorithmic"] = {"bytes_per_ray": round(bpr, 1), "nodes_per_ray": round(nbar, 2), "tris_per_ray": round(tbar, 2), "p_hit": round(phit, 4),
                              "gbs": round(bpr * rays_this_rank / t / 1e9, 1),
                              "note": "SURVEY 8(d) canonical-BVH2 visit bytes; cache-served, exceeds the HBM peak by construction -- not a bound"}
    return out


LINE_CAP = 3072               # the driver keeps ~8 KB of stdout: the printed line stays under 3 KB, the full record goes to a file


def _g(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


def _rnd(x, n):
    return None if x is None else round(float(x), n)


def _mat_stage_ms(m):
    """wall time of the three training stages of the e2e Mat run over its step count (the GPU period is material_step.ms_back_to_back)"""
    ph = _g(m, "phases_s")
    if not ph or not m.get("steps"):
        return None
    return round(1e3 * sum(ph.get("stage%d" % k, 0.0) for k in range(3)) / m["steps"], 4)


def compact_line(full, full_path):
    """the ONE stdout line (<= LINE_CAP bytes): the contract keys + one number per side measurement; everything else lives in the file `full` names
    (the 20 KB line of round 5 no longer fitted the driver's capture)"""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full.get(k) for k in keep}
    out["config"] = {"workload": (_g(full, "config", "workload") or "")[:320], "parallelism": (_g(full, "config", "parallelism") or "")[:200]}
    if _g(full, "config", "workload_make_or_load_s") is not None:
        out["setup_s"] = round(float(_g(full, "config", "workload_make_or_load_s")) + float(_g(full, "config", "bvh_build_s") or 0.0), 2)
    rf = full.get("roofline")
    if rf is not None:
        bind = rf.get("binding")
        o = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms")}
        o["traffic_kind"] = "fabric side of L2 (Infinity Cache + HBM), PMC per launch" if rf.get("traffic") is not None else None
        o["binding"] = bind
        o["binding_frac"] = _g(rf, "limits", bind, "frac") if bind else None
        # SURVEY 8(d) algorithmic bytes / live kernel time over the 8 TB/s peak: > 1 because L1 / L2 / scalar cache serve a 4x more compact tree -- not a bound
        o["algorithmic_frac"] = _rnd(_g(rf, "algorithmic", "gbs") / HBM_PEAK_GBS, 3) if _g(rf, "algorithmic", "gbs") is not None else None
        o["algorithmic_bytes_per_ray"] = _g(rf, "algorithmic", "bytes_per_ray")
        o["fabric_bytes_per_ray"] = _g(rf, "limits", "hbm", "bytes_per_ray")
        o["l2_hit"] = _g(rf, "limits", "hbm", "l2_hit_rate")
        o["valu_frac"], o["l1_frac"], o["chain_frac"] = _g(rf, "limits", "valu", "frac"), _g(rf, "limits", "l1", "frac"), _g(rf, "limits", "chain", "frac")
        o["profile"] = (rf.get("profile") or "").split(" ")[0] or None
        o["profile_matches_live"] = rf.get("profile_matches_live")
        if rf.get("note"):
            o["note"] = rf["note"][:160]
        out["roofline"] = o
    cb = full.get("cpu_baseline")
    if cb is not None:
        out["cpu_baseline"] = {"value": _rnd(cb.get("value"), 3), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": (cb.get("sample") or "")[:120]}
    for key in ("material_step", "material_step_view_mode"):
        m = full.get(key)
        if m is not None:
            out[key] = {"ms": m.get("ms"), "ms_back_to_back": m.get("ms_back_to_back"), "frac": _g(m, "roofline", "frac"), "mat_shard": m.get("mat_shard"),
                        "collective_bytes_per_step": m.get("collective_bytes_per_step")}
    rk = full.get("ranks")
    if rk is not None:
        if "projected" in rk:
            out["projected_speedup"] = {n: _g(rk, "projected", n, "projected_speedup") for n in ("2", "4", "8") if _g(rk, "projected", n) is not None}
        else:
            out["ranks"] = {"kernel_ms": rk.get("kernel_ms"), "assembled_ok": rk.get("assembled_ok"),
                            "collective_bytes_per_step": rk.get("collective_bytes_per_step"), "rccl_ranks": rk.get("rccl_ranks"), "backend": rk.get("backend")}
    ex = full.get("extra_workloads")
    if ex:
        out["extra_mrays_s"] = {w: e.get("value") for w, e in ex.items()}
    e2 = full.get("e2e")
    if e2 is not None:
        out["e2e_s"] = {"error": e2["error"][:120]} if "error" in e2 else {
            "irrt": _g(e2, "irrt", "total_s"), "mat": _g(e2, "mat", "total_s"), "mat_steps": _g(e2, "mat", "steps"), "mat_stage_ms_per_step": _mat_stage_ms(e2.get("mat"))}
    out["full"] = full_path
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_CAP:            # never let an unexpected string push the line past the cap: drop the optional blocks, last first
        for k in ("e2e_s", "extra_mrays_s", "projected_speedup", "material_step_view_mode", "ranks"):
            if k in out and len(line) > LINE_CAP:
                out.pop(k)
                line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= LINE_CAP, len(line)
    return line


def write_full(full):
    """the whole record (what round 5 printed) -> $TEXIR_BENCH_FULL or gpurun_out/bench_full.json; returns the path written (repo-relative where possible)"""
    path = os.environ.get("TEXIR_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        return "unwritten (%s)" % e
    return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path


XGMI_LINK_GBS = 153.0         # per xGMI link and direction (7 links x ~153 GB/s per GPU, point to point)


def project_scaling(sc, d_pos, d_nrm, d_shift, ids_all, spp, res, dev, t1_ms, worlds=(2, 4, 8)):
    """The 1 -> N curve of ONE texture as far as one GPU can measure it: every block-cyclic shard of the N-rank partition (dist_util.shard_block_cyclic,
    the lists the N ranks would trace) is launched ALONE on this GPU and timed with HIP events; a rank's step is its kernel + the all_gather of the
    compacted texel values (modelled: ring over one xGMI link per neighbour, (N-1)/N x 12 B x valid texels / 153 GB/s) + the scatter of the other ranks'
    rows into its texture (measured here).  projected_speedup = T(1) / (max_r T(shard r) + t_all_gather + t_scatter).  What this cannot see: RCCL launch
    latency, clock differences between GPUs, host jitter of 8 processes."""
    from texir_code_amd import dist_util
    out = {}
    irr = torch.zeros((res * res, 3), device=dev)
    n_all = int(ids_all.numel())
    for w in worlds:
        plan = dist_util.shard_plan(ids_all, w, BLOCK, dev)
        ms = []
        for r in range(w):
            ids = plan[r].to(torch.int32)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        # the receiving side of assemble_shards on rank 0: scatter of the other ranks' rows
        mx = max(int(p_.numel()) for p_ in plan)
        fake = torch.zeros((w * mx, 3), device=dev)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        mine = torch.zeros((mx, 3), device=dev)
        mine[: plan[0].numel()] = irr[plan[0]]
        for r in range(1, w):
            irr[plan[r]] = fake[r * mx: r * mx + plan[r].numel()]
        b.record()
        torch.cuda.synchronize()
        scatter_ms = a.elapsed_time(b)
        ag_bytes = 12.0 * mx * w
        ag_ms = (w - 1) / w * ag_bytes / (XGMI_LINK_GBS * 1e9) * 1e3
        step = max(ms) + ag_ms + scatter_ms
        out[str(w)] = {"shard_kernel_ms": [round(x, 2) for x in ms], "shard_kernel_ms_max": round(max(ms), 2), "shard_kernel_ms_mean": round(float(np.mean(ms)), 2),
                       "imbalance": round(max(ms) / float(np.mean(ms)), 4), "sum_over_shards_vs_one_launch": round(sum(ms) / t1_ms, 4),
                       "all_gather_bytes": int(ag_bytes), "all_gather_ms_model": round(ag_ms, 3), "scatter_ms": round(scatter_ms, 3),
                       "projected_step_ms": round(step, 2), "projected_speedup": round(t1_ms / step, 3), "projected_mrays_s": round(n_all * spp / step / 1e3, 1)}
    out["note"] = ("each shard of the N-rank block-cyclic partition (%d-texel blocks of the Morton-ordered valid list) traced alone on ONE GPU; all_gather modelled as a ring over one "
                   "xGMI link (%.0f GB/s); T(1) = %.2f ms (this run's kernel average).  sum_over_shards_vs_one_launch > 1 = what splitting costs (fewer texels per launch: tail + cold caches)"
                   % (BLOCK, XGMI_LINK_GBS, t1_ms))
    return out


def spawn(args):
    """--gpus N without a launcher: re-execute under torch.distributed.run, one rank per GPU"""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        sys.exit("bench.py: %d ranks requested (--gpus %d) but this box has %d GPU(s); refusing to report a %d-GPU number"
                 % (args.gpus, args.gpus, n_dev, args.gpus))
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-mat", action="store_true")
    ap.add_argument("--mat", action="store_true", help="run the material-step leg on workloads other than c4 too (plumbing tests)")
    ap.add_argument("--mat-steps", type=int, default=50)
    ap.add_argument("--mat-res", type=int, default=4096, help="albedo / roughness texture size of the material leg (4096 = BASELINE.json's)")
    ap.add_argument("--mat-cube", type=int, default=128)
    ap.add_argument("--no-project", dest="project", action="store_false", help="skip the per-shard timing of the 2/4/8-rank partitions (N = 1 only; `ranks.projected`)")
    ap.add_argument("--e2e", dest="e2e", action="store_true", default=None, help="time the stages end to end from files on disk (tools/stage_time.py): `e2e` in the line; "
                    "default: on with the full c4 headline line, off otherwise")
    ap.add_argument("--no-e2e", dest="e2e", action="store_false")
    ap.add_argument("--extra", default=None, help="comma-separated extra workloads timed after the headline (IrT only), reported under extra_workloads; "
                    "default: c4_scan (the hostile sibling) and house (the multi-room family) next to the full c4 headline line, nothing otherwise or with --no-cpu / --no-mat; `none` switches it off")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn(args)

    # host torch ops on the path are tiny (the per-step CPU-generator draw of 2P floats): one intra-op thread, like the reference's
    # runners (trainer/train_material.py:34).  With the default (= all cores) the draw's OpenMP fork/join jitters by milliseconds.
    torch.set_num_threads(1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    backend = os.environ.get("TEXIR_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py: no GPU visible (the hot path has no CPU fallback)")
    if local >= n_dev:
        if backend == "nccl":
            sys.exit("bench.py: local rank %d has no GPU of its own (%d visible); one process per GPU" % (local, n_dev))
        local = local % n_dev                # gloo smoke test: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            sys.exit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    from texir_code_amd import scene as S, dist_util

    # N > 1: rank 0 generates the synthetic scene + texel G-buffers once, the other ranks load them (8 ranks each building the 1 M-triangle
    # scene and the 4096^2 G-buffer would cost 8x the host RAM and time before the first barrier).  Every rank derives the same directory
    # from the rendezvous port; rank 0 removes it at the end if this run created it.
    made_cache = None
    if world > 1 and not os.environ.get("TEXIR_SYNTH_CACHE"):
        import tempfile
        made_cache = os.path.join(tempfile.gettempdir(), "texir_synth_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getuid()))
        os.environ["TEXIR_SYNTH_CACHE"] = made_cache

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_irt(name, steps, warmup):
        """time `steps` passes of the hot path over workload `name`; returns the measurements + what the later legs need"""
        if world > 1 and rank != 0 and os.environ.get("TEXIR_SYNTH_CACHE"):
            dist.barrier()                  # rank 0 generates into the cache, the others load it
        t_make = time.perf_counter()
        sc0, pos, nrm, valid, shift, res, spp = make_workload(name)
        t_make = time.perf_counter() - t_make         # rank 0: generate (+ write the cache at N > 1); other ranks: load the cache -- outside the timed region, inside the driver's wall clock
        if world > 1 and rank == 0 and os.environ.get("TEXIR_SYNTH_CACHE"):
            dist.barrier()
        if args.spp:
            spp = args.spp
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
        plan = dist_util.shard_plan(ids_all, world, BLOCK, dev) if world > 1 else None
        irr = torch.zeros((res * res, 3), device=dev)
        for _ in range(warmup):
            irr.zero_()
            sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr)
            if world > 1:
                dist_util.assemble_shards(irr, ids_all, BLOCK, plan=plan)     # one all_gather of every rank's own texel values (12 B per valid texel)
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            irr.zero_()
            ev[k][0].record()            # HIP events on the stream the kernel is launched on (torch's current stream)
            sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr)
            ev[k][1].record()
            if world > 1:
                dist_util.assemble_shards(irr, ids_all, BLOCK, plan=plan)     # one all_gather of every rank's own texel values (12 B per valid texel)
        barrier()
        dt = time.perf_counter() - t0
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        ranks = None
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            # per-rank kernel time (load imbalance of the block-cyclic shards shows here) and a check of the assembly: rank 0 re-traces a 1 %
            # sample of the texel blocks -- its own and the other ranks' -- alone and compares with the all-reduced texture, bit for bit
            km = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(km, torch.tensor([kern_ms], device=dev, dtype=torch.float64))
            km = [float(x.item()) for x in km]
            ok = None
            if rank == 0:
                nb = (int(ids_all.numel()) + BLOCK - 1) // BLOCK
                pick = torch.arange(0, nb, 100)
                sample = torch.cat([ids_all[int(b) * BLOCK:(int(b) + 1) * BLOCK] for b in pick]).to(dev)
                alone = torch.zeros_like(irr)
                sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=sample, out=alone)
                ok = bool(torch.equal(alone[sample.long()], irr[sample.long()]))
                del alone
            info = sc.info()
            scene_bytes = info["node_bytes"] + info["tri_bytes"] + info["uv_bytes"] + info["tex_bytes"] + sc0["hdr"].nbytes
            plan_i = irt_plan_parts(spp)
            ranks = {"footprint_per_rank": {"replicated_scene_bytes": int(scene_bytes), "texel_gbuffers_bytes": int(pos.nbytes + nrm.nbytes + shift.nbytes),
                                            "irt_scratch_bytes": int(12 * plan_i * ids.numel()), "irradiance_bytes": int(irr.numel() * 4),
                                            "note": "everything read-only is replicated; the IrT partial-sum scratch (12 B x %d parts per LISTED texel) scales with the rank's own shard" % plan_i},
                     "kernel_ms_min": round(min(km), 3), "kernel_ms_max": round(max(km), 3), "kernel_ms": [round(x, 3) for x in km],
                     "rccl_ranks": dist.get_world_size() if backend == "nccl" else None, "backend": dist.get_backend(),
                     "collective_bytes_per_step": int(12 * ids_all.numel()),
                     "assembled_ok": ok, "assembled_check": "rank 0 alone re-traced every 100th %d-texel block of the list; bit-equal to the assembled (all-gathered) texture" % BLOCK}
        T, _, tex_res, _, style = WORKLOADS[name]
        n_valid = int(ids_all.numel())
        fp = irt_fingerprint(sc, d_pos, d_nrm, d_shift, ids_all, spp, dev) if world == 1 else None
        if world == 1 and args.project:
            ranks = {"projected": project_scaling(sc, d_pos, d_nrm, d_shift, ids_all, spp, res, dev, kern_ms)}
        return {"sc": sc, "sc0": sc0, "pos": pos, "nrm": nrm, "valid": valid, "shift": shift, "res": res, "spp": spp, "irr": irr, "ids": ids,
                "dt": dt, "kern_ms": kern_ms, "ranks": ranks, "n_valid": n_valid, "build_s": build_s, "make_s": t_make, "fingerprint": fp, "kernel": sc.irt_kernel_name(int(ids.numel()), spp),
                "value": n_valid * spp * steps / dt / 1e6,
                "desc": "%s: IrT %d spp, %dx%d texels (%d valid), %d-tri synthetic %s mesh, %dx%d radiance texture (%s)"
                        % (name, spp, res, res, n_valid, T, {"room": "indoor", "scan": "scan-like (rotated clutter, slats, openings)", "house": "3x3-room house (doors, untessellated shell + dense clutter, windows)"}[style], tex_res, tex_res,
                           {3: "RGBE x 2^5, read as 4-byte shared-exponent texels: 5x5-texel lines", 4: "RGBE x 2^5, read as 4-byte shared-exponent texels: 8x4-texel lines"}.get(
                               sc.texture_layout(), "RGB32F tiles, layout %d" % sc.texture_layout()))}

    r = run_irt(args.workload, args.steps, args.warmup)
    mat = mat_view = None
    if not args.no_mat and (args.workload == "c4" or args.mat):
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):          # (constructors print like the reference's; stdout carries the JSON line only)
            if world > 1:
                # both shardings of the material step: pixel = the trainer's default and the parity mode (one view per step, small all_gathers),
                # view = throughput mode (one view per rank per step, texture-gradient all_reduce).  `material_step` is the default mode's.
                mat = mat_leg_pixel(r["sc"], r["sc0"], r["irr"], r["res"], dev, rank, world, steps=args.mat_steps, warmup=min(5, args.mat_steps),
                                    cube=args.mat_cube, tres=args.mat_res)
                torch.cuda.empty_cache()
                mat_view = mat_leg(r["sc"], r["sc0"], r["irr"], r["res"], dev, rank, world, steps=args.mat_steps, warmup=min(5, args.mat_steps),
                                   cube=args.mat_cube, tres=args.mat_res)
            else:
                mat = mat_leg(r["sc"], r["sc0"], r["irr"], r["res"], dev, rank, world, steps=args.mat_steps, warmup=min(5, args.mat_steps),
                              cube=args.mat_cube, tres=args.mat_res)

    out = None
    if rank == 0:
        out = {
            "metric": "IrT generation throughput", "value": round(r["value"], 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(r["dt"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": r["desc"],
                       "parallelism": "texel-sharded x%d (block-cyclic %d) + RCCL all_gather of the compacted texel values" % (world, BLOCK) if world > 1 else "single GPU",
                       "bvh_build_s": round(r["build_s"], 2), "workload_make_or_load_s": round(r["make_s"], 2), "scene": r["sc"].info()},
        }
        if mat is not None:
            out["material_step"] = mat
        if mat_view is not None:
            out["material_step_view_mode"] = mat_view
        if r["ranks"] is not None:
            out["ranks"] = r["ranks"]
            if "assembled_ok" in r["ranks"]:
                out["assembled_ok"] = r["ranks"]["assembled_ok"]
        rays_this_rank = int(r["ids"].numel()) * r["spp"]
        alg = cpu = None
        if not args.no_cpu:
            cpu, counters = cpu_leg(r["sc0"], r["pos"], r["nrm"], r["valid"], r["shift"], r["spp"], timed=world == 1)
            alg = algorithmic_bytes_per_ray(counters, r["spp"])
        out["roofline"] = roofline(args.workload, r["kernel"], r["kern_ms"], rays_this_rank, world, alg, r["fingerprint"])
        # what one launch MUST move through HBM at least once: the scene it reads (tree in both forms, triangles, corner uvs, the hit shader's texture copy),
        # the texel G-buffers + id list, and the texture it writes -- everything beyond this in `traffic` is re-reading (cache misses), everything in
        # `algorithmic` beyond `traffic` was served by L1 / L2 / LDS / the scalar cache
        info = r["sc"].info()
        comp = (info["node_bytes"] + info["tri_bytes"] + info["uv_bytes"] + info["tex_bytes"]
                + int(r["ids"].numel()) * (12 + 12 + 8 + 4) + int(r["ids"].numel()) * 12)
        out["roofline"]["compulsory_bytes"] = int(comp)
        out["roofline"]["reading"] = ("top-level achieved / peak / frac / traffic = the FABRIC side of L2 (requests to the Infinity Cache and HBM together: Infinity-Cache hits are "
                                      "included, so HBM proper carries less) -- measured, and NOT what bounds this kernel: see `binding` and `waves` (a dependent fetch -> box test -> sort "
                                      "-> push chain at 8 waves per SIMD); compulsory_bytes / traffic = %.4f of the fabric traffic is first-touch, the rest is re-read"
                                      % (comp / out["roofline"]["traffic"] if out["roofline"].get("traffic") else float("nan")))
        if cpu is not None:
            out["cpu_baseline"] = cpu
    # further workloads (IrT only), never the headline
    if args.extra is None:          # the full default line (what the driver runs) carries the hostile sibling; tool invocations (--no-cpu / --no-mat) stay lean
        # (N > 1 stays lean: every extra workload costs rank 0 another ~25 s of generation + cache hand-off inside the driver's wall clock, and the siblings are never the headline)
        args.extra = "c4_scan,house" if (args.workload == "c4" and not args.no_cpu and not args.no_mat and world == 1) else ""
    extras = [w for w in args.extra.split(",") if w and w != "none"]
    if extras:
        del r
        torch.cuda.empty_cache()
        ex = {}
        for w in extras:
            xs, xw = min(args.steps, 5), min(args.warmup, 1)       # (the kernel's time is stable after one pass: the siblings need not repeat the headline's step count)
            e = run_irt(w, xs, xw)
            if rank == 0:
                ex[w] = {"value": round(e["value"], 2), "unit": "Mrays/s", "ms_per_step": round(e["dt"] / xs * 1e3, 3), "steps": xs, "warmup": xw, "workload": e["desc"],
                         "kernel": e["kernel"], "scene": e["sc"].info()}
                if e["ranks"] is not None:
                    ex[w]["ranks"] = e["ranks"]
                if load_pmc(w, e["kernel"])[0] is not None:         # (measured bounds where tools/profile_round.sh has profiled this workload too)
                    ex[w]["roofline"] = roofline(w, e["kernel"], e["kern_ms"], int(e["ids"].numel()) * e["spp"], world, None, e["fingerprint"])
            del e
            torch.cuda.empty_cache()
        if rank == 0:
            out["extra_workloads"] = ex
    # the stages end to end, files on disk to files on disk (tools/stage_time.py): default with the full headline line, --e2e / --no-e2e force it
    if args.e2e is None:
        args.e2e = args.workload == "c4" and not args.no_cpu and not args.no_mat and world == 1
    if args.e2e and rank == 0 and world == 1:
        r = None
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import contextlib
            import stage_time
            with contextlib.redirect_stdout(sys.stderr):
                out["e2e"] = stage_time.run(args.workload if args.workload in stage_time.SIZES else "c4")
        except Exception as e:                # the headline line must survive a failing stage run; the failure is in the line
            out["e2e"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank == 0:
        path = write_full(out)
        print(compact_line(out, path), flush=True)
    if world > 1:
        dist.barrier()
        if rank == 0 and made_cache:
            import shutil
            shutil.rmtree(made_cache, ignore_errors=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
