"""Which mip levels do the tap lists of the bench's 16 material views write?  Per view: (texture size, fetch mode, touches level 0, touches level 1, level-1 keys, keys) per
tap list, and whether FusedAdam may skip the level-1 stack (`_texir_l1_zero`).  Round 4: every view of the c4 bench touches levels 0 AND 1 (16 ... 76 k level-1 texels of ~90 ... 220 k
keys), so the level-1 skip never applies there."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from texir_code_amd import scene as S
sc0, pos, nrm, valid, shift, res, spp = bench.make_workload("c4")
dev = torch.device("cuda", 0)
sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
irr = torch.rand(res * res, 3, device=dev) + 0.2
model, views, data, loss_fn, opt = bench.mat_setup(sc, sc0, irr, res, dev)
for v in range(len(views)):
    mvp, cam, gt, gmask, seg, fm, room = data[v]
    preds = model(mvp, v, cam, 2)
    loss = loss_fn(gt, preds, gmask, fm, seg, stage=2, room_seg_mask=room)[0]
    opt.zero_grad(); loss.backward()
    gb = model._gbuffer(mvp, v)
    out = []
    for k, t in gb.items():
        if isinstance(k, tuple) and k[0] == "_taps" and t is not None:
            key = t[0]; n0 = k[1] * k[2]; n1 = n0 // 4
            out.append((k[1], k[4], bool(t[5]), bool(t[7]), int(((key >= n0) & (key < n0 + n1)).sum()), int(key.numel())))
    print(v, out, [bool(getattr(p, "_texir_l1_zero", False)) for p in (model.materials_a, model.materials_r)], flush=True)
    # lengths of the per-texel tap lists the gather walks (one thread per list, taps summed in list order): segments, taps, mean / 99 % / 99.9 % / longest list
    for k, t in gb.items():
        if isinstance(k, tuple) and k[0] == "_taps" and t is not None and k[4] == 1:
            c = t[2].to(torch.float32)
            q = torch.quantile(c, torch.tensor([0.5, 0.99, 0.999], device=c.device))
            big = t[2] > 64
            print("   lists %d taps %d  median %.0f  99%% %.0f  99.9%% %.0f  longest %d   lists > 64 taps: %d holding %.1f %% of the taps" % (
                c.numel(), int(c.sum()), q[0], q[1], q[2], int(c.max()), int(big.sum()), 100.0 * float(t[2][big].sum()) / float(c.sum())), flush=True)
            break
    opt.step()
