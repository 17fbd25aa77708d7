"""Print the numbers DESIGN.md section 5 / README quote from a final-session directory (gpurun_out/<tag> or profiles/<round>):
usage: python tools/summarise_final.py gpurun_out/r04_final"""
import csv, json, os, sys

d = sys.argv[1]


def line(name):
    p = os.path.join(d, name)
    full = p.replace(".json", "_full.json")           # round 6: the printed line is compact, the whole record sits beside it
    if os.path.exists(full):
        return json.load(open(full))
    if not os.path.exists(p):
        return None
    rows = [x for x in open(p).read().strip().splitlines() if x.startswith("{")]
    return json.loads(rows[-1]) if rows else None


def rl(r):
    L = r.get("limits") or {}
    g = lambda k, f: (L.get(k) or {}).get(f)
    return "mem %s (%.2f TB/s, %s B/ray, L2 hit %s) | binding %s | valu %s (lanes %s) | l1 %s | wait %s issue %s | compulsory %s" % (
        r.get("frac"), (r.get("achieved") or 0) / 1e3, g("hbm", "bytes_per_ray"), g("hbm", "l2_hit_rate"), r.get("binding"), g("valu", "frac"), g("valu", "lane_utilisation"),
        g("l1", "frac"), (r.get("waves") or {}).get("wait_any_frac"), (r.get("waves") or {}).get("active_inst_any_frac"), r.get("compulsory_bytes"))


for name in ("bench_default.json", "bench_c2.json", "bench_c1.json", "bench_2rank_gloo_one_gpu.json"):
    b = line(name)
    if not b:
        print(name, "missing")
        continue
    print("==", name, b["config"]["workload"][:40], "|", b["value"], "Mrays/s", b["ms_per_step"], "ms | n_gpus", b["n_gpus"])
    if "roofline" in b:
        print("   ", rl(b["roofline"]), "| kernel_ms", b["roofline"].get("kernel_ms"))
    if "cpu_baseline" in b:
        c = b["cpu_baseline"]
        print("    cpu", round(c["value"], 2), c["unit"], "cores", c["cores"], "one_thread", c.get("one_thread"), "eff", c.get("parallel_efficiency"), c.get("host"))
    if "material_step" in b:
        m = b["material_step"]
        print("    mat", m["ms"], "ms, views/step", m.get("views_per_step"), "roofline frac", (m.get("roofline") or {}).get("frac"), "traffic", (m.get("roofline") or {}).get("traffic"), (m.get("roofline") or {}).get("traffic_note"))
    for w, e in (b.get("extra_workloads") or {}).items():
        print("    extra", w, e["value"], e["ms_per_step"], rl(e["roofline"]) if "roofline" in e else "")
    if "ranks" in b:
        print("    ranks", {k: v for k, v in b["ranks"].items() if k != "assembled_check"})
for sub in ("", "prof"):
    p = os.path.join(d, sub, "c4_kernel_stats.csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            if "irt_group_kernel<false" in r["Name"]:
                print("rocprof stats:", r["Name"][:50], "calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6)
    p = os.path.join(d, sub, "bench_default_under_rocprof.json")
    if os.path.exists(p):
        b = json.loads([x for x in open(p).read().strip().splitlines() if x.startswith("{")][-1])
        print("under rocprof: value", b["value"], "kernel_ms", b["roofline"]["kernel_ms"])
p = os.path.join(d, "mat_step_trace.txt")
if os.path.exists(p):
    print(open(p).read())
p = os.path.join(d, "pytest_gpu.txt")
if os.path.exists(p):
    print(open(p).read().strip().splitlines()[-1])
