"""Fold the counter CSVs of tools/mat_step_pmc.sh into profiles/pmc_mat_step.json: fabric-side bytes of one replayed material step, per kernel.
usage: python tools/mat_step_pmc.py <dir with mat_rd.csv / mat_wr.csv / mat_tcc.csv> <out.json>"""
import collections, csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAT_SOURCES = ["material.hip", "loss.hip", "kernels.hip", "device_common.h", "kernels.h", "capi.hip", "Makefile"]


def mat_src_sha():
    h = hashlib.sha256()
    for f in MAT_SOURCES:
        h.update(open(os.path.join(ROOT, "texir_code_amd", "csrc", f), "rb").read())
    for f in ("graph_step.py", "texture.py", "optim.py", "models.py", "loss.py"):
        h.update(open(os.path.join(ROOT, "texir_code_amd", f), "rb").read())
    return h.hexdigest()[:16]


def steps_of(path, n_steps=20):
    """rows of the last n_steps replayed steps: a step ends with the roughness texture's Adam launch"""
    rows = list(csv.DictReader(open(path)))
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
    ids = sorted(disp)
    names = [disp[i][0]["Kernel_Name"] for i in ids]
    # (the step's last launch: the batched Adam over both textures, or -- TEXIR_TEX_BATCH=0 -- the one-channel texture's own)
    ends = [k for k, nm in enumerate(names) if "adam_tex" in nm and ("kernel<1>" in nm or "batch_kernel" in nm)]
    assert len(ends) > n_steps + 1, "not enough replayed steps in %s" % path
    per_step = ends[-1] - ends[-2]
    assert all(ends[-k] - ends[-k - 1] == per_step for k in range(1, n_steps + 1)), "steps of different length"
    first, last = ends[-n_steps - 1] + 1, ends[-1]
    out = collections.OrderedDict()
    for k in range(first, last + 1):
        pos = (k - first) % per_step
        nm = names[k].split("(")[0]
        key = "%02d %s" % (pos, nm[:70])
        for r in disp[ids[k]]:
            out.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"]) / n_steps
    return out, per_step


if __name__ == "__main__":
    d, outp = sys.argv[1], sys.argv[2]
    rd, per_step = steps_of(os.path.join(d, "mat_rd.csv"))
    wr, _ = steps_of(os.path.join(d, "mat_wr.csv"))
    tcc = steps_of(os.path.join(d, "mat_tcc.csv"))[0] if os.path.exists(os.path.join(d, "mat_tcc.csv")) else {}
    kernels = []
    tot_r = tot_w = 0.0
    for key in rd:
        c = rd[key]
        # request sizes as counted (32 / 64 / 128 B sub-counters; what they do not cover is taken as 128 B, the size every request of the
        # streaming and tracing kernels has: MI355X_MICROARCH.md, HBM section)
        n = c.get("TCC_EA0_RDREQ_sum", 0.0)
        n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        rbytes = 32.0 * n32 + 64.0 * n64 + 128.0 * max(n - n32 - n64, n128)
        wbytes = wr.get(key, {}).get("WRITE_SIZE", 0.0) * 1024.0
        t = tcc.get(key, {})
        hit = t.get("TCC_HIT_sum", 0.0) / max(t.get("TCC_HIT_sum", 0.0) + t.get("TCC_MISS_sum", 0.0), 1.0)
        kernels.append({"kernel": key, "read_bytes": rbytes, "write_bytes": wbytes, "l2_hit_rate": round(hit, 4)})
        tot_r += rbytes
        tot_w += wbytes
    # calibration on a kernel whose bytes are known: the 3-channel Adam reads p, m, v (12 B / parameter) + the level-1 gradient and writes p, m, v + level 1
    adam = [k for k in kernels if "adam_tex_vec_kernel<3>" in k["kernel"]]
    cal = None
    if adam:
        n_par = 4096 * 4096 * 3
        cal = {"adam3_read_expected": 12.0 * n_par + 4.0 * n_par / 4 * (1 + 0.25), "adam3_read_counted": adam[0]["read_bytes"],
               "adam3_write_expected": 12.0 * n_par + 4.0 * n_par / 4, "adam3_write_counted": adam[0]["write_bytes"]}
    res = {"mat_src_sha": mat_src_sha(), "kernels_per_step": per_step, "read_bytes_per_step": tot_r, "write_bytes_per_step": tot_w,
           "fabric_bytes_per_step": tot_r + tot_w, "calibration": cal, "kernels": kernels,
           "source": "rocprofv3 --pmc over bench.py's material leg, three passes, mean of the last 20 replayed steps (tools/mat_step_pmc.sh)"}
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
    for k in kernels:
        print("%-75s read %8.1f MB  write %8.1f MB  L2 hit %.3f" % (k["kernel"], k["read_bytes"] / 1e6, k["write_bytes"] / 1e6, k["l2_hit_rate"]))
