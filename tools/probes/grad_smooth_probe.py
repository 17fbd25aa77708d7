"""probe (tool, round 6): the one untrimmed gradient of tests/test_gpu_mat_step_oracle.py that sits at 8.5e-4 (smooth radiance, 64^2 / 128^2 textures, stage 2, roughness) while every
other case is at 1e-6 ... 5e-5 -- and did not move when the texture's contrast went from +-20 % to +-5 %.  Feeds the product's specular forward and the C oracle's with the SAME per-pixel
inputs (the oracle's rasterised G-buffer, its fetched materials) and compares the traced radiance sample by sample; then lists the texels that carry the gradient difference.
usage: python tools/probes/grad_smooth_probe.py   (GPU box)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_mat_step_oracle as T
from texir_code_amd import scene as S
from texir_code_amd.loss import RenderLoss
from oracle import mat_step as MS, ref_torch as RT

golden = lambda name: np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)
ra, rr, c, SPP = 64, 128, 32, 16
m, oracle, views = T._world(golden, c, ra, rr, smooth=True)
loss_fn = RenderLoss("L1", 1, lazy_item=True)
gen = torch.Generator().manual_seed(3)
for stage in (0, 1, 2):
    opt = T._fresh_optimizer(m, stage)
    oracle.make_optimizer(stage, T.LR)
    for key, v in views.items():
        shift = torch.rand(6 * c * c, 2, generator=gen)
        d = T._cu(v)
        m._static_shift = shift.cuda()
        preds = m(v["mvp"], key, d["cam"], stage)
        m._static_shift = None
        loss = loss_fn(d["gt"], preds, d["gmask"], d["fm"], d["seg"], stage=stage, room_seg_mask=d["room"] if stage == 2 else None)[0]
        opt.zero_grad(); loss.backward()
        gr = opt.dense_grad(m.materials_r).cpu().numpy() if m.materials_r.requires_grad else None
        opt.zero_grad()
        lo, oa, orr = oracle.grads(key, v["mvp"].numpy(), v["cam"], stage, shift.numpy(), v["gt"], v["gmask"], v["fm"], v["seg"], v["room"])
        if stage != 2:
            continue
        ref = orr.numpy()
        dev = np.abs(gr - ref).reshape(-1)
        print("stage 2 view %s: roughness gradient rel-L2 %.2e; |ref| max %.3e; texels with |dev| > 1e-3 max|ref|: %d; top deviations (texel, dev / max|ref|): %s"
              % (key, np.linalg.norm(gr - ref) / np.linalg.norm(ref), np.abs(ref).max(), int((dev > 1e-3 * np.abs(ref).max()).sum()),
                 [(int(i), round(float(dev[i] / np.abs(ref).max()), 4)) for i in np.argsort(dev)[::-1][:6]]))
        # the same per-pixel inputs through both tracers
        gb = oracle.gbuffer(key, v["mvp"].numpy())
        uv, da = gb["uv"], gb["uv_da"]
        with torch.no_grad():
            alb = RT.texture(oracle.a, uv, da, "linear-mipmap-linear", oracle.max_mip)
            rough = RT.texture(oracle.r, uv, da, "linear-mipmap-linear", oracle.max_mip)
        nrm, pos, irr = gb["normal"], gb["position"], gb["irr"]
        pts = pos + 1e-2 * nrm
        cam = torch.as_tensor(v["cam"], dtype=torch.float32)
        _, Lo = oracle.osc.spec_forward(nrm.numpy(), alb.numpy(), rough.numpy().reshape(-1), pts.numpy(), irr.numpy(), cam.numpy(), shift.numpy().astype(np.float32), SPP,
                                        tracer=oracle.tracer, return_ls=True)
        f = lambda t, k: t.reshape(-1, k).float().contiguous().cuda()
        _, Lp, _ = S.spec_forward_raw(m.scene, f(nrm, 3), f(alb, 3), rough.reshape(-1).float().contiguous().cuda(), f(pts, 3), f(irr, 3), cam.cuda(), shift.cuda().contiguous(), SPP)
        Lp = Lp.cpu().numpy().reshape(-1, SPP, 3); Lo = np.asarray(Lo).reshape(-1, SPP, 3)
        dL = np.abs(Lp - Lo).max(-1)
        big = np.argwhere(dL > 0.02)
        print("   traced radiance, %d pixels x %d samples: %d samples differ by > 0.02 (texture values are 0.5 +- 0.025); of those %d are a MISS (radiance 0) on one side only" % (
            Lp.shape[0], SPP, len(big), int(sum((Lp[p, s].max() == 0) != (Lo[p, s].max() == 0) for p, s in big))))
        for p, s in big[:8]:
            print("      pixel %d sample %d: product %s oracle %s" % (p, s, np.round(Lp[p, s], 4).tolist(), np.round(Lo[p, s], 4).tolist()))
