"""How much of the 4k atlas ever receives a gradient?  (probe for a dense-equivalent Adam that skips texel blocks whose moments are still
exactly zero: p -= lr * 0 / (sqrt(0) + eps) is the identity, so torch.optim.Adam's dense update leaves such texels unchanged bit for bit.)
Runs the bench's material problem for a number of epochs over its views and reports, per block granularity, the share of blocks in
which any first or second moment is non-zero.  usage: python tools/probes/adam_touched_probe.py [epochs]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from texir_code_amd import scene as S  # noqa: E402


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    sc0, pos, nrm, valid, shift, res, spp = bench.make_workload("c4")
    sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
    ids = torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32).to(dev)
    irr = torch.zeros((res * res, 3), device=dev)
    sc.irt_generate(torch.from_numpy(pos).to(dev).reshape(-1, 3), torch.from_numpy(nrm).to(dev).reshape(-1, 3), torch.from_numpy(shift).to(dev), 16, "uniform",
                    texel_ids=ids, out=irr)
    model, views, data, loss_fn, opt = bench.mat_setup(sc, sc0, irr, res, dev)
    for p in (model.materials_a, model.materials_r):
        p.grad = torch.zeros_like(p)
    print("atlas texels referenced by the IrT list (valid): %.3f" % (float(valid.reshape(-1).astype(bool).mean()),))
    for ep in range(epochs):
        for v in range(len(views)):
            mvp, cam, gt, gmask, seg, fm, room = data[v]
            preds = model(mvp, v, cam, 2)
            loss = loss_fn(gt, preds, gmask, fm, seg, stage=2, room_seg_mask=room)[0]
            opt.zero_grad(set_to_none=False)
            loss.backward()
            opt.step()
            if ep == 0 and v in (0, 3, 7, 15):
                report(opt, model, "after %d views" % (v + 1))
        report(opt, model, "after epoch %d" % (ep + 1))


def report(opt, model, label):
    for name, p in (("albedo", model.materials_a), ("roughness", model.materials_r)):
        st = opt.state[p]
        nz = ((st["exp_avg"] != 0) | (st["exp_avg_sq"] != 0)).reshape(p.shape[0], p.shape[1], -1).any(-1)      # [H, W] texels
        H, W = nz.shape
        out = ["%s %s: texels %.3f" % (label, name, float(nz.float().mean()))]
        for bw in (64, 256, 1024, 4096):                       # row segments of bw texels
            out.append("row x%d %.3f" % (bw, float(nz.reshape(H, W // bw, bw).any(-1).float().mean())))
        for t in (8, 16, 32, 64):                              # square tiles
            out.append("tile %dx%d %.3f" % (t, t, float(nz.reshape(H // t, t, W // t, t).any(3).any(1).float().mean())))
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
