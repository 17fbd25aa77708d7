import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from texir_code_amd import cameras, gbuffer as GB
    from texir_code_amd.scene import Scene
    g = np.load(os.path.join(ROOT, "tests", "golden", "irt_room.npz"))
    sc = Scene(g["verts"], g["tris"], g["tri_uvs"], g["hdr"], device=0)
    out = {}
    for key, E in (("v0", cameras.grid_cameras(2)[0]), ("v1", cameras.grid_cameras(2)[3])):
        gb = GB.cast_gbuffer(sc, cameras.cube_mvps(E)[0], 32, flip_v=True)
        for k, v in gb.items():
            if torch.is_tensor(v):
                out[key + "_" + k] = v.cpu().numpy()
    # trace_shade radiance on random rays
    rng = np.random.default_rng(1)
    org = np.tile(np.array([[4.0, 1.5, 3.0]], np.float32), (200000, 1))
    d = rng.normal(size=(200000, 3)).astype(np.float32)
    rad, t, pid, uv = sc.trace_shade(torch.from_numpy(org), torch.from_numpy(d), return_hits=True)
    out["rad"] = rad.cpu().numpy(); out["pid"] = pid.cpu().numpy(); out["t"] = t.cpu().numpy()
    np.savez(sys.argv[2], **out)
    sys.exit(0)
res = {}
for name, lib in (("quad", None), ("slot", os.path.join(ROOT, "build_ab", "libtexir_uvslot.so"))):
    env = dict(os.environ)
    if lib: env["TEXIR_HIP_LIB"] = lib
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/dbg_%s.npz" % name], env=env)
    res[name] = np.load("/tmp/dbg_%s.npz" % name)
a, b = res["quad"], res["slot"]
for k in a.files:
    x, y = a[k], b[k]
    if x.dtype.kind == "f":
        fin = np.isfinite(x) & np.isfinite(y)
        dm = np.abs(x - y)[fin].max() if fin.any() else 0
        print(k, x.shape, "max abs diff", dm, "n differing", int((x != y).sum()), "nan mismatch", int((np.isfinite(x) != np.isfinite(y)).sum()))
        if (x != y).any() and x.ndim >= 2:
            idx = np.argwhere((x != y).reshape(-1, x.shape[-1]).any(-1))[:5, 0]
            print("   first rows", idx, x.reshape(-1, x.shape[-1])[idx], y.reshape(-1, x.shape[-1])[idx])
    else:
        print(k, "n differing", int((x != y).sum()))
