"""probe (tool): would the specular forward be faster as  sample directions -> one-ray-per-thread trace_shade -> weighted reduce ?
Times the fused spec_kernel against texir_trace_shade on the SAME reflection rays (c4 scene, 98 304 pixels x 16 GGX samples)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from texir_code_amd import cameras, gbuffer as GB, scene as S

sc0, pos, nrm, valid, shift, res, spp = bench.make_workload("c4")
sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
mvp, cam = cameras.cube_mvps(cameras.grid_cameras(4)[5])
gb = GB.cast_gbuffer(sc, mvp, 128, flip_v=True)
P, Sn = 6 * 128 * 128, 16
n = gb["normal"].reshape(P, 3).contiguous()
pts = (gb["position"].reshape(P, 3) + 1e-2 * n).contiguous()
cam = cam.cuda().reshape(3)
r = torch.full((P,), 0.1, device="cuda")
alb = torch.full((P, 3), 0.5, device="cuda")
irr = torch.ones((P, 3), device="cuda")
sh = torch.rand(P, 2)

def timed(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3

shd = sh.cuda()
t_fused = timed(lambda: S.spec_render(sc, n, alb, r, pts, irr, cam, shd, Sn))
h = S.generate_dir(n, Sn, sh, "importance", r)                        # [P,S,3] half vectors
v = torch.nn.functional.normalize(cam[None] - pts, dim=-1, eps=1e-4)
vdh = (v[:, None] * h).sum(-1, keepdim=True).clamp(0, 1)
l = (2 * vdh * h - v[:, None]).reshape(P * Sn, 3).contiguous()
org = pts[:, None].expand(P, Sn, 3).reshape(P * Sn, 3).contiguous()
t_trace = timed(lambda: sc.trace_shade(org, l))
# the same rays, pixel-major vs sample-major order
l2 = l.reshape(P, Sn, 3).transpose(0, 1).reshape(P * Sn, 3).contiguous(); org2 = org.reshape(P, Sn, 3).transpose(0, 1).reshape(P * Sn, 3).contiguous()
t_trace2 = timed(lambda: sc.trace_shade(org2, l2))
print("fused spec_kernel %.1f us; trace_shade on the same %d rays: pixel-major %.1f us, sample-major %.1f us" % (t_fused, P * Sn, t_trace, t_trace2))

# ---- would binning pay?  The same rays in other orders (round 4): 8x8-pixel tiles, and inside a tile sorted by direction cell (8x8 octahedral grid, Morton order);
# per roughness (the GGX lobe of r = 0.1 is a needle -- the 16 rays of a pixel are nearly one direction -- and widens with r) ----
def morton(a, b):
    k = torch.zeros_like(a)
    for bit in range(3):
        k |= ((a >> bit) & 1) << (2 * bit) | ((b >> bit) & 1) << (2 * bit + 1)
    return k


def orders(rough):
    r = torch.full((P,), rough, device="cuda")
    h = S.generate_dir(n, Sn, sh, "importance", r)
    vdh = (v[:, None] * h).sum(-1, keepdim=True).clamp(0, 1)
    l = (2 * vdh * h - v[:, None]).reshape(P * Sn, 3).contiguous()
    org = pts[:, None].expand(P, Sn, 3).reshape(P * Sn, 3).contiguous()
    p = torch.arange(P, device="cuda")
    c = 128
    face, rem = p // (c * c), p % (c * c)
    i, j = rem // c, rem % c
    tile = face * 256 + (i // 8) * 16 + (j // 8)
    tile_r = tile[:, None].expand(P, Sn).reshape(-1)
    pix_in_tile = ((i % 8) * 8 + (j % 8))[:, None].expand(P, Sn).reshape(-1)
    smp = torch.arange(Sn, device="cuda")[None].expand(P, Sn).reshape(-1)
    # octahedral cell of the direction
    d = l / l.abs().sum(-1, keepdim=True)
    ox = torch.where(d[:, 2] >= 0, d[:, 0], (1 - d[:, 1].abs()) * torch.sign(d[:, 0]))
    oy = torch.where(d[:, 2] >= 0, d[:, 1], (1 - d[:, 0].abs()) * torch.sign(d[:, 1]))
    cx = ((ox * 0.5 + 0.5) * 8).clamp(0, 7).long()
    cy = ((oy * 0.5 + 0.5) * 8).clamp(0, 7).long()
    cell = morton(cx, cy)
    out = {}
    out["pixel-major (as shipped)"] = torch.arange(P * Sn, device="cuda")
    out["8x8 tiles, pixel-major inside"] = torch.argsort(tile_r * 1024 + pix_in_tile * 16 + smp)
    out["8x8 tiles, direction cell inside"] = torch.argsort((tile_r * 64 + cell) * 1024 + pix_in_tile * 16 + smp)
    out["direction cell, then tile (global)"] = torch.argsort((cell * 4096 * 6 + tile_r) * 1024 + pix_in_tile * 16 + smp)
    res = []
    for name, perm in out.items():
        o2, l2 = org[perm].contiguous(), l[perm].contiguous()
        res.append((name, timed(lambda: sc.trace_shade(o2, l2))))
    cells_per_tile = torch.unique(tile_r * 64 + cell).numel() / torch.unique(tile_r).numel()
    return res, cells_per_tile


for rough in (0.1, 0.3, 0.6):
    res, cpt = orders(rough)
    print("roughness %.1f: %.1f occupied direction cells per 8x8 tile (of 64);  trace_shade of the %d rays: " % (rough, cpt, P * Sn) + ";  ".join("%s %.1f us" % kv for kv in res), flush=True)
