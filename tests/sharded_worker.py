"""Worker of tests/test_gpu_scale.py::test_pixel_sharded_material_step_*: runs the trainer's stage-1 and stage-2 steps on a fixed synthetic problem
(20 k-triangle room, 1024^2 albedo / roughness textures, two views, c = 32) and writes the textures after every stage.
    python tests/sharded_worker.py single  out.npz            one process, graph_step.GraphedMatStep (the single-GPU step)
    torchrun --nproc-per-node N tests/sharded_worker.py sharded out.npz [eager]   N ranks, sharded_step.ShardedMatStep (rank 0 writes)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    eager = len(sys.argv) > 3 and sys.argv[3] == "eager"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)                                   # (the ranks of the test share one GPU)
    if mode == "sharded" and world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("TEXIR_DIST_BACKEND", "gloo"))
    from texir_code_amd import cameras, conf as C
    from texir_code_amd.graph_step import GraphedMatStep
    from texir_code_amd.loss import RenderLoss
    from texir_code_amd.models import MaterialModel
    from texir_code_amd.optim import FusedAdam
    from texir_code_amd.scene import Scene
    from texir_code_amd.sharded_step import ShardedMatStep
    from texir_code_amd.trainer.train_material import build_masks
    g = np.load(os.path.join(ROOT, "tests", "golden", "irt_room.npz"))
    c, res = 32, 1024
    cf = C.parse_string("train{ pano_img_res = [%d,%d]\n sample_light = [64,16]\n hdr_exposure = 0 }\nmodels{ render{ sample_type = [uniform, importance] } }" % (2 * c, 4 * c))
    gen = torch.Generator().manual_seed(21)
    sc = Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"], device=0)
    m = MaterialModel.from_arrays(sc, g["hdr"], torch.rand(128, 128, 3, generator=gen) + 0.3, cf, albedo_res=res, roughness_res=res)
    m.lean_outputs = True
    with torch.no_grad():
        m.materials_a.copy_((0.2 + 0.6 * torch.rand(res, res, 3, generator=gen)).cuda())
        m.materials_r.copy_((0.1 + 0.5 * torch.rand(res, res, 1, generator=gen)).cuda())
    views = {}
    for key, E in (("v0", cameras.grid_cameras(2)[0]), ("v1", cameras.grid_cameras(2)[3])):
        mvp, cam = cameras.cube_mvps(E)
        gt = (torch.rand(6, c, c, 3, generator=gen) * 1.5).cuda()
        gmask = (torch.rand(6, c, c, 1, generator=gen) > 0.1).float().cuda()
        segs = torch.randint(40, 49, (6, c, c, 1), generator=gen).float().cuda()
        seg, fm, _ = build_masks(segs, (torch.rand(6, c, c, 3, generator=gen) - 0.5).cuda())
        rooms = torch.randint(0, 2, (6, c, c, 1), generator=gen).float().cuda()
        room = ((torch.arange(2.0, device="cuda").reshape(2, 1, 1, 1, 1) - rooms.unsqueeze(0)) == 0).float()
        views[key] = (mvp, cam.cuda(), gt, gmask, seg, fm, room)
    loss_fn = RenderLoss("L1", 1, lazy_item=True, unit_upstream=True)
    snaps, losses, graphs_used = {}, [], True
    for stage in tuple(int(x) for x in os.environ.get("SW_STAGES", "1,2").split(",")):
        m.materials_a.data = torch.clamp(m.materials_a.data, 0.0)
        m.materials_a.requires_grad = stage == 2
        m.materials_r.requires_grad = True
        opt = FusedAdam(m.parameters(), lr=3e-2, fuse_mip_fold=True)
        opt.set_clamp(m.materials_r, 1e-2, 0.8)
        if stage == 2:
            opt.set_clamp(m.materials_a, 0.0, float("inf"))
        params = [m.materials_a, m.materials_r]
        step = GraphedMatStep(m, loss_fn, opt, params) if mode == "single" else ShardedMatStep(m, loss_fn, opt, params, use_graph=not eager)
        for key, (mvp, cam, gt, gmask, seg, fm, room) in views.items():
            step.capture(key, mvp, cam, gt, gmask, seg, fm, room if stage == 2 else None, stage)
        if mode == "sharded":
            graphs_used = graphs_used and all(("graphs" in st) == (not eager) for st in step.views.values())
        torch.manual_seed(100 + stage)                             # the CPU-generator stream of the steps (identical on every rank)
        for key in os.environ.get("SW_ORDER", "v0,v1,v0,v1,v0").split(","):
            P = 6 * c * c
            shift = torch.rand(P, 1, 2).reshape(P, 2)
            loss = step.step(key, stage, shift=shift)
            losses.append(float(loss))
        torch.cuda.synchronize()
        snaps["a%d" % stage] = m.materials_a.detach().cpu().numpy().copy()
        snaps["r%d" % stage] = m.materials_r.detach().cpu().numpy().copy()
    if rank == 0:
        np.savez(out, losses=np.array(losses), graphs=np.array([1 if graphs_used else 0]), world=np.array([world]), **snaps)
    if mode == "sharded" and world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
