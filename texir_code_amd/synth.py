"""Seeded procedural indoor scenes: the reference's dataset is private
(README.md:21-34), so parity fixtures and the bench run on this generator instead.

scene(T, seed): axis-aligned 8 x 3 x 6 m room with inward faces + ~40 floor-standing boxes,
every planar patch tessellated into a displaced quad grid so that the mesh has EXACTLY T
triangles; one UV chart per patch, shelf-packed into [0,1]^2; HDR radiance texture
(log-normal base x smooth noise + emissive rectangles) stored in the layout the tracer
consumes (already "flipped + exposed", models/tracer_o3d_irt.py:77-81: texel (row r, col c)
covers uv = ((c+.5)/W, (r+.5)/H)); texel G-buffers (pos + 1e-2*n, n, valid) produced by
evaluating the chart parametrisation at texel centres (replaces generate_positions +
calcute_position_normal_texture, tracer_o3d_irt.py:99-142, for synthetic data).
"""
import math

import numpy as np

ROOM = (8.0, 3.0, 6.0)
GUTTER = 3.0 / 1024.0


class Patch:
    __slots__ = ("o", "eu", "ev", "n", "lu", "lv", "gu", "gv", "rect", "cls", "vbase", "tbase", "ntri", "amp", "phase", "fixed", "flat")

    def __init__(self, o, eu, ev, cls):
        self.o = np.asarray(o, np.float64)
        self.eu = np.asarray(eu, np.float64)
        self.ev = np.asarray(ev, np.float64)
        n = np.cross(self.eu, self.ev)
        self.lu = float(np.linalg.norm(self.eu))
        self.lv = float(np.linalg.norm(self.ev))
        self.n = n / np.linalg.norm(n)
        self.cls = cls
        self.fixed = None               # (gu, gv) when the generator fixes the tessellation of this patch (house style: the untessellated shell)
        self.flat = False               # no displacement (large planar triangles)


def _room_and_boxes(rng, n_boxes):
    X, Y, Z = ROOM
    P = []
    # room, normals pointing inward
    P.append(Patch((0, 0, 0), (0, 0, Z), (X, 0, 0), 46))          # floor   n=+y
    P.append(Patch((0, Y, 0), (X, 0, 0), (0, 0, Z), 44))          # ceiling n=-y
    P.append(Patch((0, 0, 0), (X, 0, 0), (0, Y, 0), 45))          # z=0 wall n=+z
    P.append(Patch((0, 0, Z), (0, Y, 0), (X, 0, 0), 45))          # z=Z wall n=-z
    P.append(Patch((0, 0, 0), (0, Y, 0), (0, 0, Z), 45))          # x=0 wall n=+x
    P.append(Patch((X, 0, 0), (0, 0, Z), (0, Y, 0), 45))          # x=X wall n=-x
    if n_boxes <= 0:
        return P
    gx, gz = 8, 5
    cells = [(i, j) for i in range(gx) for j in range(gz)]
    order = rng.permutation(len(cells))[:n_boxes]
    cw, cd = (X - 0.4) / gx, (Z - 0.4) / gz
    for k, ci in enumerate(order):
        i, j = cells[ci]
        w = cw * rng.uniform(0.35, 0.8)
        d = cd * rng.uniform(0.35, 0.8)
        h = rng.uniform(0.3, 2.2)
        x0 = 0.2 + i * cw + rng.uniform(0.05, cw - w - 0.05)
        z0 = 0.2 + j * cd + rng.uniform(0.05, cd - d - 0.05)
        cls = int(rng.integers(0, 43))
        x1, z1 = x0 + w, z0 + d
        # 5 faces, outward normals (no bottom: it would coincide with the floor)
        P.append(Patch((x0, h, z0), (0, 0, d), (w, 0, 0), cls))       # top  +y
        P.append(Patch((x0, 0, z0), (0, h, 0), (w, 0, 0), cls))       # z=z0 -z
        P.append(Patch((x0, 0, z1), (w, 0, 0), (0, h, 0), cls))       # z=z1 +z
        P.append(Patch((x0, 0, z0), (0, 0, d), (0, h, 0), cls))       # x=x0 -x
        P.append(Patch((x1, 0, z0), (0, h, 0), (0, 0, d), cls))       # x=x1 +x
    return P


def _rot(rng):
    """uniformly random rotation matrix (random unit quaternion)"""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _box_faces(c, R, half, cls):
    """six outward-facing parallelogram patches of the box centre c, axes R[:,k], half extents half[k]"""
    P = []
    for ax in range(3):
        a1, a2 = (ax + 1) % 3, (ax + 2) % 3
        for sgn in (+1.0, -1.0):
            n = sgn * R[:, ax]
            eu, ev = 2 * half[a1] * R[:, a1], 2 * half[a2] * R[:, a2]
            if sgn < 0:
                eu, ev = ev, eu                       # keeps eu x ev = outward normal
            o = c + half[ax] * n - 0.5 * eu - 0.5 * ev
            P.append(Patch(o, eu, ev, cls))
    return P


def _scan_clutter(rng):
    """`scan` style (bench workload c4_scan): the same 8 x 3 x 6 m shell, but with two openings (a window and a door: rays escape,
    p_hit < 1), ~150 randomly ROTATED boxes anywhere in the volume (nothing axis-aligned, some floating like shelves), a
    venetian blind of thin slats in front of the window, and strongly displaced, jittered surfaces -- what a scanned real-world
    indoor mesh does to a BVH (README.md:21-34 describes the reference's data as captured scenes)."""
    X, Y, Z = ROOM
    P = []
    P.append(Patch((0, 0, 0), (0, 0, Z), (X, 0, 0), 46))          # floor
    P.append(Patch((0, Y, 0), (X, 0, 0), (0, 0, Z), 44))          # ceiling
    # z=0 wall (n=+z) with a window x in [2.5,5.5], y in [0.9,2.2]: four patches around the hole
    P.append(Patch((0, 0, 0), (2.5, 0, 0), (0, Y, 0), 45))
    P.append(Patch((5.5, 0, 0), (X - 5.5, 0, 0), (0, Y, 0), 45))
    P.append(Patch((2.5, 0, 0), (3.0, 0, 0), (0, 0.9, 0), 45))
    P.append(Patch((2.5, 2.2, 0), (3.0, 0, 0), (0, Y - 2.2, 0), 45))
    P.append(Patch((0, 0, Z), (0, Y, 0), (X, 0, 0), 45))          # z=Z wall
    P.append(Patch((0, 0, 0), (0, Y, 0), (0, 0, Z), 45))          # x=0 wall
    # x=X wall (n=-x) with a door z in [2.0,3.0], y in [0,2.1]
    P.append(Patch((X, 0, 0), (0, 0, 2.0), (0, Y, 0), 45))
    P.append(Patch((X, 0, 3.0), (0, 0, Z - 3.0), (0, Y, 0), 45))
    P.append(Patch((X, 2.1, 2.0), (0, 0, 1.0), (0, Y - 2.1, 0), 45))
    for _ in range(150):
        half = rng.uniform(0.05, 0.45, 3) * rng.uniform(0.3, 1.0)
        c = np.array([rng.uniform(0.5, X - 0.5), rng.uniform(0.1, Y - 0.3), rng.uniform(0.5, Z - 0.5)])
        P += _box_faces(c, _rot(rng), half, int(rng.integers(0, 43)))
    # blind: 40 slats 3 m x 3 cm x 2 mm, tilted ~35 degrees about x, 8 cm in front of the window
    tilt = np.deg2rad(35.0)
    Rs = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]], np.float64)
    for k in range(40):
        c = np.array([4.0, 0.92 + k * (1.26 / 40), 0.08])
        P += _box_faces(c, Rs, np.array([1.5, 0.015, 0.001]), 42)
    return P


HOUSE = (16.0, 3.0, 12.0)            # footprint of the `house` style: 3 x 3 rooms
HOUSE_WALL = 0.12


def _wall(o, eu_dir, length, height, cls, hole=None):
    """vertical wall rectangle from corner o along the horizontal unit vector eu_dir (normal = eu_dir x +y), optionally with a rectangular hole
    (a0, a1, b0, b1) in wall coordinates: the patches around the hole"""
    o, e = np.asarray(o, np.float64), np.asarray(eu_dir, np.float64)
    up = np.array([0.0, 1.0, 0.0])
    if hole is None:
        return [Patch(o, e * length, up * height, cls)]
    a0, a1, b0, b1 = hole
    out = []
    if a0 > 1e-6:
        out.append(Patch(o, e * a0, up * height, cls))
    if length - a1 > 1e-6:
        out.append(Patch(o + e * a1, e * (length - a1), up * height, cls))
    if b0 > 1e-6:
        out.append(Patch(o + e * a0, e * (a1 - a0), up * b0, cls))
    if height - b1 > 1e-6:
        out.append(Patch(o + e * a0 + up * b1, e * (a1 - a0), up * (height - b1), cls))
    return out


def _house(rng):
    """`house` style (bench workload `house`): what the reference's data looks like (README.md:21-34: whole multi-room houses reconstructed from scans) --
    a 16 x 3 x 12 m footprint cut into 3 x 3 rooms by 12 cm walls, nine door openings between rooms (rays cross rooms through them; most of the
    house is occluded from any one texel), six windows and an entrance to the outside (p_hit < 1), every room's shell left as a handful of LARGE flat
    triangles (fixed coarse grids: what mesh simplification leaves of planar walls), and ~25 rotated / stacked clutter boxes plus furniture per room that
    receive all the remaining triangles (millimetre-scale, displaced like scanned surfaces).  Returns (patches, emissive rectangles)."""
    X, Y, Z = HOUSE
    t = HOUSE_WALL
    nx, nz = 3, 3
    xs, zs = np.linspace(0, X, nx + 1), np.linspace(0, Z, nz + 1)
    P, emissive, clutter = [], [], []
    door_w, door_h = 1.0, 2.1

    def shell(ps):
        for p in ps:
            p.fixed = (max(1, int(round(p.lu / 2.5))), max(1, int(round(p.lv / 2.5))))
            p.flat = True
        P.extend(ps)

    for i in range(nx):
        for j in range(nz):
            x0 = xs[i] + (t / 2 if i > 0 else 0.0)
            x1 = xs[i + 1] - (t / 2 if i < nx - 1 else 0.0)
            z0 = zs[j] + (t / 2 if j > 0 else 0.0)
            z1 = zs[j + 1] - (t / 2 if j < nz - 1 else 0.0)
            dx, dz = x1 - x0, z1 - z0
            zc, xc = 0.5 * (z0 + z1), 0.5 * (x0 + x1)
            shell([Patch((x0, 0, z0), (0, 0, dz), (dx, 0, 0), 46)])                           # floor   n = +y
            ceil_index = len(P)
            shell([Patch((x0, Y, z0), (dx, 0, 0), (0, 0, dz), 44)])                           # ceiling n = -y
            emissive.append((ceil_index, 0.4, 0.4, 0.2, 0.2))
            door_x = (zc - door_w / 2, zc + door_w / 2, 0.0, door_h)                          # door in a wall running along z (world z range)
            door_z = (xc - door_w / 2, xc + door_w / 2, 0.0, door_h)
            win = (0.9, 2.2)
            # z = z0 wall (n = +z), eu = +x from (x0, 0, z0)
            if j == 0:
                hole = (dx / 2 - 0.8, dx / 2 + 0.8, win[0], win[1])                            # window to the outside
            else:
                hole = (door_z[0] - x0, door_z[1] - x0, 0.0, door_h) if (i + (j - 1)) % 2 == 0 else None
            shell(_wall((x0, 0, z0), (1, 0, 0), dx, Y, 45, hole))
            # z = z1 wall (n = -z), eu = -x from (x1, 0, z1)
            if j == nz - 1:
                hole = (x1 - door_z[1], x1 - door_z[0], 0.0, door_h) if i == 1 else None      # the entrance
            else:
                hole = (x1 - door_z[1], x1 - door_z[0], 0.0, door_h) if (i + j) % 2 == 0 else None
            shell(_wall((x1, 0, z1), (-1, 0, 0), dx, Y, 45, hole))
            # x = x0 wall (n = +x), eu = -z from (x0, 0, z1)
            hole = (z1 - door_x[1], z1 - door_x[0], 0.0, door_h) if i > 0 else None
            w_first = len(P)
            shell(_wall((x0, 0, z1), (0, 0, -1), dz, Y, 45, hole))
            if i == 0 and j == 1:
                emissive.append((w_first, 0.35, 0.4, 0.3, 0.4))                                # a bright wall panel
            # x = x1 wall (n = -x), eu = +z from (x1, 0, z0)
            if i == nx - 1:
                hole = (dz / 2 - 0.8, dz / 2 + 0.8, win[0], win[1])                            # window
            else:
                hole = (door_x[0] - z0, door_x[1] - z0, 0.0, door_h)
            shell(_wall((x1, 0, z0), (0, 0, 1), dz, Y, 45, hole))
            # door frames through the wall's thickness (each door once: from the room on its low side): two jambs, lintel underside, threshold
            if i < nx - 1:
                xa, xb = x1, x1 + t
                shell([Patch((xa, 0, door_x[0]), (t, 0, 0), (0, door_h, 0), 45),                # jamb at z = door low, n = +z
                       Patch((xb, 0, door_x[1]), (-t, 0, 0), (0, door_h, 0), 45),               # jamb at z = door high, n = -z
                       Patch((xa, door_h, door_x[0]), (t, 0, 0), (0, 0, door_w), 45),           # lintel underside, n = -y
                       Patch((xa, 0, door_x[0]), (0, 0, door_w), (t, 0, 0), 46)])               # threshold, n = +y
            if j < nz - 1 and (i + j) % 2 == 0:
                za, zb = z1, z1 + t
                shell([Patch((door_z[0], 0, zb), (0, 0, -t), (0, door_h, 0), 45),               # jamb at x = door low, n = +x
                       Patch((door_z[1], 0, za), (0, 0, t), (0, door_h, 0), 45),                # jamb at x = door high, n = -x
                       Patch((door_z[0], door_h, za), (door_w, 0, 0), (0, 0, t), 45),           # lintel underside, n = -y
                       Patch((door_z[0], 0, za), (0, 0, t), (door_w, 0, 0), 46)])               # threshold, n = +y
            # clutter: rotated boxes anywhere in the room's volume, clear of the door axes' first 60 cm
            for _ in range(22):
                half = rng.uniform(0.05, 0.45, 3) * rng.uniform(0.3, 1.0)
                c = np.array([rng.uniform(x0 + 0.7, x1 - 0.7), rng.uniform(0.1, Y - 0.3), rng.uniform(z0 + 0.7, z1 - 0.7)])
                clutter += _box_faces(c, _rot(rng), half, int(rng.integers(0, 43)))
            # furniture: three floor-standing axis-aligned boxes (no bottom face)
            for _ in range(3):
                w, d, h = rng.uniform(0.5, 1.6), rng.uniform(0.4, 0.9), rng.uniform(0.4, 2.0)
                bx, bz = rng.uniform(x0 + 0.1, x1 - w - 0.1), rng.uniform(z0 + 0.1, z1 - d - 0.1)
                cls = int(rng.integers(0, 43))
                clutter += [Patch((bx, h, bz), (0, 0, d), (w, 0, 0), cls), Patch((bx, 0, bz), (0, h, 0), (w, 0, 0), cls),
                            Patch((bx, 0, bz + d), (w, 0, 0), (0, h, 0), cls), Patch((bx, 0, bz), (0, 0, d), (0, h, 0), cls),
                            Patch((bx + w, 0, bz), (0, h, 0), (0, 0, d), cls)]
            # a venetian blind in front of the windows of the two corner rooms on the z = 0 side
            if j == 0 and i in (0, nx - 1):
                tilt = np.deg2rad(35.0)
                Rs = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]], np.float64)
                for k in range(40):
                    c = np.array([xc, 0.92 + k * (1.26 / 40), z0 + 0.08])
                    clutter += _box_faces(c, Rs, np.array([0.85, 0.015, 0.001]), 42)
    return P + clutter, emissive


def _allocate_grids(P, T):
    """choose (gu, gv) per patch, quads ~ area, 2*sum(gu*gv) <= T (remainder fixed by edge splits).  Patches with a `fixed` grid keep it; the
    others share what is left of T."""
    fixed = [p for p in P if p.fixed is not None]
    if fixed:
        for p in fixed:
            p.gu, p.gv = p.fixed
        T = T - sum(2 * p.gu * p.gv for p in fixed)
        P = [p for p in P if p.fixed is None]
        if T < 2 * len(P):
            raise ValueError("not enough triangles for the fixed shell plus one quad per remaining patch")
    if T <= 2 * len(P):
        for p in P:
            p.gu = p.gv = 1
        return
    area = np.array([p.lu * p.lv for p in P])
    lo, hi = 1e-3, 1e7

    def total(dens):
        t = 0
        for p in P:
            gu = max(1, int(round(p.lu * math.sqrt(dens))))
            gv = max(1, int(round(p.lv * math.sqrt(dens))))
            t += 2 * gu * gv
        return t

    for _ in range(80):
        mid = math.sqrt(lo * hi)
        if total(mid) <= T:
            lo = mid
        else:
            hi = mid
    for p in P:
        p.gu = max(1, int(round(p.lu * math.sqrt(lo))))
        p.gv = max(1, int(round(p.lv * math.sqrt(lo))))
    del area


def _pack(P):
    """shelf-pack chart rectangles (size ~ physical size) into [0,1]^2 with gutters."""
    order = sorted(range(len(P)), key=lambda i: -P[i].lv)
    lo, hi = 1e-4, 1.0

    def try_pack(s, assign):
        x = y = GUTTER
        row_h = 0.0
        for i in order:
            w, h = P[i].lu * s, P[i].lv * s
            if x + w + GUTTER > 1.0:
                x = GUTTER
                y += row_h + GUTTER
                row_h = 0.0
            if x + w + GUTTER > 1.0 or y + h + GUTTER > 1.0:
                return False
            if assign:
                P[i].rect = (x, y, w, h)
            x += w + GUTTER
            row_h = max(row_h, h)
        return True

    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if try_pack(mid, False):
            lo = mid
        else:
            hi = mid
    assert try_pack(lo, True)


def _patch_heights(p, rng_phase, gu, gv):
    """smooth displacement along the normal, zero on the patch border (keeps seams closed)."""
    a = np.linspace(0.0, 1.0, gu + 1)[:, None]
    b = np.linspace(0.0, 1.0, gv + 1)[None, :]
    ph = rng_phase
    f = (np.sin(2 * np.pi * (3.0 * a * p.lu / 2.0 + ph[0])) * np.cos(2 * np.pi * (2.0 * b * p.lv / 2.0 + ph[1]))
         + 0.5 * np.sin(2 * np.pi * (7.0 * a * p.lu / 2.0 + 5.0 * b * p.lv / 2.0 + ph[2])))
    win = np.minimum(1.0, 8.0 * np.minimum(np.minimum(a, 1 - a), np.minimum(b, 1 - b)) * np.ones_like(f))
    return p.amp * f * win


def _scan_heights(p, rng, gu, gv):
    """scan style: three octaves of smooth noise (wavelengths 60 / 15 / 4 cm) plus per-vertex white jitter (sensor noise), zero on
    the patch border"""
    a = np.linspace(0.0, 1.0, gu + 1)[:, None]
    b = np.linspace(0.0, 1.0, gv + 1)[None, :]
    f = np.zeros((gu + 1, gv + 1))
    for wl, wgt in ((0.6, 1.0), (0.15, 0.45), (0.04, 0.2)):
        f = f + wgt * (2.0 * _smooth_noise(rng, gu + 1, gv + 1, max(1.0, wl / max(p.lu / max(gu, 1), 1e-9))) - 1.0)
    f = f + 0.08 * rng.normal(size=f.shape)
    win = np.minimum(1.0, 8.0 * np.minimum(np.minimum(a, 1 - a), np.minimum(b, 1 - b)) * np.ones_like(f))
    return p.amp * f * win


def make_scene(T, seed=666, tex_res=1024, n_boxes=40, amp=3e-3, style="room"):
    """returns dict: verts [V,3] f32, tris [T,3] i32, tri_uvs [3T,2] f32, hdr [tex_res,tex_res,3] f32,
    patches (list of Patch), plus per-chart GT material colours.  style = "room" (axis-aligned room + boxes)
    or "scan" (_scan_clutter: rotated clutter, thin slats, openings, noisy surfaces; T >= 20000) or "house" (_house: 3 x 3 rooms joined by doors,
    large untessellated shell triangles next to dense clutter, windows; T >= 10000)."""
    rng = np.random.default_rng(seed)
    if T < 12 or T % 2:
        raise ValueError("T must be even and >= 12")
    emissive = None
    if style == "scan":
        if T < 4000:
            raise ValueError("scan style needs T >= 4000")
        P = _scan_clutter(rng)
        amp = 1.2e-2
    elif style == "house":
        if T < 10000:
            raise ValueError("house style needs T >= 10000")
        P, emissive = _house(rng)
        amp = 6e-3
    else:
        P = _room_and_boxes(rng, 0 if T < 12 + 40 * 10 else n_boxes)
    _allocate_grids(P, T)
    _pack(P)
    verts, tris, tuvs, tcls = [], [], [], []
    vbase = 0
    tcount = 0
    for p in P:
        p.amp = amp if (p.gu > 1 and p.gv > 1 and not p.flat) else 0.0
        p.phase = rng.uniform(0, 1, 3)
        gu, gv = p.gu, p.gv
        a = np.linspace(0.0, 1.0, gu + 1)
        b = np.linspace(0.0, 1.0, gv + 1)
        hgt = _scan_heights(p, rng, gu, gv) if (style in ("scan", "house") and p.amp > 0.0) else _patch_heights(p, p.phase, gu, gv)
        pos = (p.o[None, None, :] + a[:, None, None] * p.eu[None, None, :] + b[None, :, None] * p.ev[None, None, :]
               + hgt[:, :, None] * p.n[None, None, :])
        verts.append(pos.reshape(-1, 3))
        ii, jj = np.meshgrid(np.arange(gu), np.arange(gv), indexing="ij")
        v00 = (ii * (gv + 1) + jj).reshape(-1) + vbase
        v10 = ((ii + 1) * (gv + 1) + jj).reshape(-1) + vbase
        v01 = (ii * (gv + 1) + jj + 1).reshape(-1) + vbase
        v11 = ((ii + 1) * (gv + 1) + jj + 1).reshape(-1) + vbase
        t = np.empty((gu * gv, 2, 3), np.int64)
        t[:, 0] = np.stack([v00, v10, v01], -1)
        t[:, 1] = np.stack([v11, v01, v10], -1)
        tris.append(t.reshape(-1, 3))
        x, y, w, h = p.rect
        ua = x + w * a
        vb = y + h * b
        uvgrid = np.stack(np.meshgrid(ua, vb, indexing="ij"), -1).reshape(-1, 2)
        tuvs.append(uvgrid[(t.reshape(-1, 3) - vbase).reshape(-1)].reshape(-1, 2))
        tcls.append(np.full(2 * gu * gv, p.cls, np.int32))
        p.vbase, p.tbase, p.ntri = vbase, tcount, 2 * gu * gv
        vbase += (gu + 1) * (gv + 1)
        tcount += 2 * gu * gv
    verts = np.concatenate(verts).astype(np.float64)
    tris = np.concatenate(tris)
    tuvs = np.concatenate(tuvs).reshape(-1, 3, 2)
    tcls = np.concatenate(tcls)
    # fix-up to exactly T: split triangles at the midpoint of their first edge (stays on the edge -> no crack)
    rem = T - tris.shape[0]
    assert rem >= 0
    if rem:
        pick = rng.choice(tris.shape[0], size=rem, replace=False)
        a_, b_, c_ = tris[pick, 0], tris[pick, 1], tris[pick, 2]
        m = np.arange(rem) + verts.shape[0]
        verts = np.concatenate([verts, 0.5 * (verts[a_] + verts[b_])])
        muv = 0.5 * (tuvs[pick, 0] + tuvs[pick, 1])
        new_t = np.stack([m, b_, c_], -1)
        new_uv = np.stack([muv, tuvs[pick, 1], tuvs[pick, 2]], 1)
        tris[pick, 1] = m
        tuvs[pick, 1] = muv
        tris = np.concatenate([tris, new_t])
        tuvs = np.concatenate([tuvs, new_uv])
        tcls = np.concatenate([tcls, tcls[pick]])
    assert tris.shape[0] == T
    sc = {
        "verts": verts.astype(np.float32), "tris": tris.astype(np.int32),
        "tri_uvs": tuvs.reshape(-1, 2).astype(np.float32), "patches": P, "seed": seed, "T": T, "tri_class": tcls, "style": style,
    }
    if emissive is not None:
        sc["emissive"] = emissive
    sc["hdr"] = make_hdr_texture(sc, tex_res, seed)
    return sc


def _chart_texels(p, res):
    """texel centres strictly inside the chart rect -> (rows, cols, a, b) with a,b in (0,1) chart coords."""
    x, y, w, h = p.rect
    c0 = int(math.ceil(x * res - 0.5)); c1 = int(math.floor((x + w) * res - 0.5))
    r0 = int(math.ceil(y * res - 0.5)); r1 = int(math.floor((y + h) * res - 0.5))
    if c1 < c0 or r1 < r0:
        return None
    cols = np.arange(c0, c1 + 1)
    rows = np.arange(r0, r1 + 1)
    a = np.clip(((cols + 0.5) / res - x) / w, 0.0, 1.0)
    b = np.clip(((rows + 0.5) / res - y) / h, 0.0, 1.0)
    return rows, cols, a, b


def _smooth_noise(rng, n_r, n_c, scale):
    """cheap band-limited noise in [0,1] on an n_r x n_c grid."""
    gr = max(2, int(n_r / scale) + 2)
    gc = max(2, int(n_c / scale) + 2)
    g = rng.uniform(0, 1, (gr, gc))
    yr = np.linspace(0, gr - 1.001, n_r)
    xc = np.linspace(0, gc - 1.001, n_c)
    y0 = yr.astype(int); x0 = xc.astype(int)
    fy = (yr - y0)[:, None]; fx = (xc - x0)[None, :]
    return (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0 + 1][:, x0] * fy * (1 - fx)
            + g[y0][:, x0 + 1] * (1 - fy) * fx + g[y0 + 1][:, x0 + 1] * fy * fx)


def make_hdr_texture(sc, res, seed=666):
    rng = np.random.default_rng(seed + 1)
    hdr = np.zeros((res, res, 3), np.float32)
    P = sc["patches"]
    for k, p in enumerate(P):
        # fill the rect plus its gutter so bilinear taps at chart borders read sane values
        x, y, w, h = p.rect
        c0 = max(0, int(math.floor((x - GUTTER / 2) * res))); c1 = min(res, int(math.ceil((x + w + GUTTER / 2) * res)))
        r0 = max(0, int(math.floor((y - GUTTER / 2) * res))); r1 = min(res, int(math.ceil((y + h + GUTTER / 2) * res)))
        base = np.exp(rng.normal(-2.0, 0.5)) * rng.uniform(0.6, 1.0, 3)
        nz = 0.6 + 0.8 * _smooth_noise(rng, r1 - r0, c1 - c0, max(4.0, res / 64.0))
        hdr[r0:r1, c0:c1, :] = (base[None, None, :] * nz[:, :, None]).astype(np.float32)
    # emissive rectangles: 4 ceiling lamps + 2 wall windows
    lamps = [(1, 0.15, 0.2, 0.1, 0.15), (1, 0.55, 0.2, 0.1, 0.15), (1, 0.15, 0.65, 0.1, 0.15), (1, 0.55, 0.65, 0.1, 0.15),
             (2, 0.3, 0.45, 0.25, 0.35), (5, 0.35, 0.4, 0.3, 0.4)]
    if sc.get("style") == "scan":          # patch 1 is the ceiling in both styles; the wall lights sit on the z=Z and x=0 walls
        lamps = lamps[:4] + [(6, 0.3, 0.45, 0.25, 0.35), (7, 0.35, 0.4, 0.3, 0.4)]
    if sc.get("emissive") is not None:     # (house: one ceiling lamp per room + a wall panel, listed by the generator)
        lamps = sc["emissive"]
    for (pi, a0, b0, da, db) in lamps:
        p = P[pi]
        x, y, w, h = p.rect
        c0 = int((x + a0 * w) * res); c1 = max(c0 + 1, int((x + (a0 + da) * w) * res))
        r0 = int((y + b0 * h) * res); r1 = max(r0 + 1, int((y + (b0 + db) * h) * res))
        hdr[r0:r1, c0:c1, :] = (rng.uniform(50.0, 500.0) * rng.uniform(0.8, 1.0, 3)).astype(np.float32)
    return hdr


def rgbe_born(hdr, exposure=5.0):
    """the texture as the reference's pipeline would hold it: `hdr * 2^-exposure` written to hdr_texture.hdr (RGBE), read back and multiplied by
    2^exposure (tracer_o3d_irt.py:77-81) -- every texel three 8-bit integers times one power of two"""
    from .io_formats import rgbe_decode_py, rgbe_encode_py
    s = np.float32(2.0 ** exposure)
    return (rgbe_decode_py(rgbe_encode_py(np.asarray(hdr, np.float32) / s)) * s).astype(np.float32)


def make_texel_gbuffer(sc, res):
    """pos [res,res,3] (= surface + 1e-2*n_geo, tracer_o3d_irt.py:110), nrm [res,res,3] (= n_geo),
    valid [res,res] u8.  Invalid texels are all-zero (seams, tracer_o3d_irt.py:137-139)."""
    pos = np.zeros((res, res, 3), np.float32)
    nrm = np.zeros((res, res, 3), np.float32)
    valid = np.zeros((res, res), np.uint8)
    V = sc["verts"].astype(np.float64)
    for p in sc["patches"]:
        ct = _chart_texels(p, res)
        if ct is None:
            continue
        rows, cols, a, b = ct
        gu, gv = p.gu, p.gv
        fa = a * gu; fb = b * gv
        ia = np.minimum(fa.astype(int), gu - 1); ib = np.minimum(fb.astype(int), gv - 1)
        la = (fa - ia)[None, :]; lb = (fb - ib)[:, None]      # [1,C], [R,1]
        IA = np.broadcast_to(ia[None, :], (rows.size, cols.size))
        IB = np.broadcast_to(ib[:, None], (rows.size, cols.size))
        grid = V[p.vbase:p.vbase + (gu + 1) * (gv + 1)].reshape(gu + 1, gv + 1, 3)
        v00 = grid[IA, IB]; v10 = grid[IA + 1, IB]; v01 = grid[IA, IB + 1]; v11 = grid[IA + 1, IB + 1]
        LA = np.broadcast_to(la, IA.shape)[..., None]; LB = np.broadcast_to(lb, IA.shape)[..., None]
        lower = (LA + LB) <= 1.0
        # tri0 = (v00, v10, v01): P = v00 + la*(v10-v00) + lb*(v01-v00)
        p0 = v00 + LA * (v10 - v00) + LB * (v01 - v00)
        n0 = np.cross(v10 - v00, v01 - v00)
        # tri1 = (v11, v01, v10): P = v11 + (1-la)*(v01-v11) + (1-lb)*(v10-v11)
        p1 = v11 + (1 - LA) * (v01 - v11) + (1 - LB) * (v10 - v11)
        n1 = np.cross(v01 - v11, v10 - v11)
        pp = np.where(lower, p0, p1)
        nn = np.where(lower, n0, n1)
        nn = nn / np.linalg.norm(nn, axis=-1, keepdims=True)
        rr = rows[:, None]; cc = cols[None, :]
        pos[rr, cc] = (pp + 1e-2 * nn).astype(np.float32)
        nrm[rr, cc] = nn.astype(np.float32)
        valid[rr, cc] = 1
    return pos, nrm, valid


def make_shifts(nt, seed=666):
    """Per-texel Cranley-Patterson shifts drawn as the reference does: CPU generator, seed 666
    (generate_ir_texture.py:45-47), torch.rand(512,1,2) per 512-texel batch in batch order
    (sample_util.py:102 called from tracer_o3d_irt.py:165-168).  Drawing [nt,2] at once from the
    same generator yields the same stream (row-major fill)."""
    import torch
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.rand(nt, 1, 2, generator=g).reshape(nt, 2).numpy()


def make_gt_materials(sc, albedo_res, rough_res, seed=666):
    """ground-truth material textures in FILE orientation (row 0 = top; sampled with (u, 1-v) by the nvdiffrast-side
    convention): albedo = per-chart colour x checker, roughness per chart in [0.1, 0.7]  """
    rng = np.random.default_rng(seed + 2)
    alb = np.full((albedo_res, albedo_res, 3), 0.5, np.float32)
    rgh = np.full((rough_res, rough_res, 1), 0.3, np.float32)
    for p in sc["patches"]:
        x, y, w, h = p.rect
        col = rng.uniform(0.2, 0.9, 3).astype(np.float32)
        r = np.float32(rng.uniform(0.1, 0.7))
        for tex, res, val in ((alb, albedo_res, col), (rgh, rough_res, r)):
            c0 = max(0, int(math.floor((x - GUTTER / 2) * res))); c1 = min(res, int(math.ceil((x + w + GUTTER / 2) * res)))
            r0 = max(0, int(math.floor((y - GUTTER / 2) * res))); r1 = min(res, int(math.ceil((y + h + GUTTER / 2) * res)))
            if tex is alb:
                yy, xx = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing="ij")
                chk = (((xx * 16 // res) + (yy * 16 // res)) % 2).astype(np.float32)[..., None]
                tex[r0:r1, c0:c1] = val[None, None, :] * (0.75 + 0.25 * chk)
            else:
                tex[r0:r1, c0:c1] = val
    # arrays above are in the tracer ("flipped") layout; files / nvdiffrast-side parameters are un-flipped
    return np.ascontiguousarray(alb[::-1]), np.ascontiguousarray(rgh[::-1])
