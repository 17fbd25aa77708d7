"""Where the cycles of the dominant kernel go, and what its rate does when waves are taken away.

    python tools/chain_probe.py --workload c4 [--spp 2048] [--sweep-spp 256] [--out profiles/chain_c4.json]

Two libraries are used, each in its own process: the shipped one (texir_code_amd/libtexir_hip.so) and the same sources compiled with
-DTEXIR_CHAIN_PROBE=1 (build_ab/libtexir_probe.so; `make -C texir_code_amd/csrc OUT=../../build_ab/libtexir_probe.so EXTRA=-DTEXIR_CHAIN_PROBE=1`),
whose waves read the shader clock (s_memtime) around every wave-level step of the traversal, around the trace, the hit shader, the pass and the chunk.

 1. attribution at full occupancy (8 waves per SIMD, the shipped configuration): cycles per step kind x steps = the wave's time; summed over the
    resident waves and divided by their number this is the kernel's duration -- `covered` says how much of it the probes account for;
 2. occupancy sweep: the persistent grid is capped (TEXIR_IRT_GRID_CAP) at 1, 2, 4 and 8 waves per SIMD; per cap the shipped kernel's time and the
    probe's cycles per step.  With n waves per SIMD and a step latency c(n), the SIMD completes n / c(n) steps per cycle (Little's law): if c(n) stays
    at c(1) the kernel is bound by the dependent chain's latency and more waves would buy rate linearly; if c(n) grows like n the SIMD's shared
    resources (issue slots, L1 lookups) are saturated and the chain's latency is hidden.

The chain bound of bench.py's roofline (`limits.chain`) = sum over kinds of steps x c(1) / resident waves: what the launch would take if every step ran at its
unloaded latency with all resident waves overlapping perfectly.  Its fraction of the measured time is how much of the kernel is explained by latency alone."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE_LIB = os.path.join(ROOT, "build_ab", "libtexir_probe.so")
NAMES = ["node_vector_steps", "node_vector_timed", "node_vector_cycles", "node_scalar_steps", "node_scalar_timed", "node_scalar_cycles",
         "leaf_steps", "leaf_timed", "leaf_cycles", "null_timed", "null_cycles", "passes", "passes_timed", "trace_cycles", "shade_cycles", "pass_cycles",
         "chunk_cycles", "chunks"]


def child(workload, spp, probe):
    """one measurement in this process: returns {"kernel_ms": ..., "probe": {...}}"""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from texir_code_amd import scene as S, dist_util
    sc0, pos, nrm, valid, shift, res, spp0 = bench.make_workload(workload)
    spp = spp or spp0
    dev = torch.device("cuda", 0)
    sc = S.Scene(sc0["verts"], sc0["tris"], sc0["tri_uvs"], sc0["hdr"], device=0)
    d_pos, d_nrm, d_shift = (torch.from_numpy(a).to(dev) for a in (pos.reshape(-1, 3), nrm.reshape(-1, 3), shift))
    ids = dist_util.morton_order(torch.nonzero(torch.from_numpy(valid.reshape(-1)) > 0)[:, 0].to(torch.int32), res).to(dev)
    irr = torch.zeros((res * res, 3), device=dev)
    # the scene's scheduler weight is measured at full occupancy, whatever cap the timed launches run under
    cap = os.environ.pop("TEXIR_IRT_GRID_CAP", None)
    from texir_code_amd import _lib
    _lib.reload_env()
    sc.irt_generate(d_pos, d_nrm, d_shift, 64, "uniform", texel_ids=ids, out=irr)          # tune + warm
    if cap is not None:
        os.environ["TEXIR_IRT_GRID_CAP"] = cap
        _lib.reload_env()
    sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if probe:
        _, st = sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr, stats=True)
    else:
        sc.irt_generate(d_pos, d_nrm, d_shift, spp, "uniform", texel_ids=ids, out=irr)
    b.record()
    torch.cuda.synchronize()
    out = {"kernel_ms": a.elapsed_time(b), "rays": int(ids.numel()) * spp, "n_ids": int(ids.numel()), "spp": spp, "kernel": sc.irt_kernel_name(int(ids.numel()), spp)}
    if probe:
        v = st.cpu().tolist()
        out["probe"] = dict(zip(NAMES, v[8:8 + len(NAMES)]))
        assert v[0] == 0, "the probe library must launch the un-counted kernel form"
    print("CHAIN_PROBE_RESULT " + json.dumps(out))


def run_child(workload, spp, probe, cap):
    env = dict(os.environ)
    env["TEXIR_SYNTH_CACHE"] = env.get("TEXIR_SYNTH_CACHE", "/tmp/texir_synth")
    if probe:
        env["TEXIR_HIP_LIB"] = PROBE_LIB
    else:
        env.pop("TEXIR_HIP_LIB", None)
    if cap:
        env["TEXIR_IRT_GRID_CAP"] = str(cap)
    else:
        env.pop("TEXIR_IRT_GRID_CAP", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--workload", workload, "--spp", str(spp)] + (["--probe"] if probe else []),
                       env=env, capture_output=True, text=True, timeout=1500)
    for line in r.stdout.splitlines():
        if line.startswith("CHAIN_PROBE_RESULT "):
            return json.loads(line[len("CHAIN_PROBE_RESULT "):])
    raise RuntimeError("child failed (probe=%s cap=%s): %s" % (probe, cap, r.stderr[-1500:]))


def per_step(p):
    """cycles per wave-level step of each kind (mean of the timed steps minus the mean empty region) and the split of a pass, from one probe dict"""
    d = lambda a, b: (a / b) if b else None
    null = d(p["null_cycles"], p["null_timed"]) or 0.0
    c = {k: (None if not p[k + "_timed"] else max(0.0, p[k + "_cycles"] / p[k + "_timed"] - null)) for k in ("node_vector", "node_scalar", "leaf")}
    spp = {k: d(p[k + "_steps"], p["passes"]) for k in ("node_vector", "node_scalar", "leaf")}
    trace = d(p["trace_cycles"], p["passes_timed"])
    shade = d(p["shade_cycles"], p["passes_timed"])
    pas = d(p["pass_cycles"], p["passes_timed"])
    if trace is not None:
        trace, shade, pas = trace - null, shade - null, pas - null
    in_steps = sum((c[k] or 0.0) * (spp[k] or 0.0) for k in c)
    chunk_per_pass = d(p["chunk_cycles"], p["passes"])
    share = None
    if pas:
        share = {"node_vector": (c["node_vector"] or 0) * (spp["node_vector"] or 0) / pas, "node_scalar": (c["node_scalar"] or 0) * (spp["node_scalar"] or 0) / pas,
                 "leaf": (c["leaf"] or 0) * (spp["leaf"] or 0) / pas, "scheduler_between_steps": (trace - in_steps) / pas, "shade": shade / pas,
                 "sampling_and_pass_overhead": (pas - trace - shade) / pas}
    return {"node_vector": c["node_vector"], "node_scalar": c["node_scalar"], "leaf": c["leaf"], "null_region": null, "shade_per_pass": shade, "trace_per_pass": trace,
            "steps_per_pass": spp, "pass_cycles": pas, "chunk_cycles_per_pass": chunk_per_pass, "share_of_pass_cycles": share,
            "wave_level_steps": p["node_vector_steps"] + p["node_scalar_steps"] + p["leaf_steps"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--sweep-spp", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--probe", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.workload, a.spp, a.probe)
    sys.path.insert(0, ROOT)
    import bench
    WAVES_PER_BLOCK, BLOCKS_FULL = 4, 2048                       # 256 CUs x 8 blocks of 4 waves = 8 waves per SIMD
    full = {"shipped": run_child(a.workload, a.spp, False, 0), "probe": run_child(a.workload, a.spp, True, 0)}
    p = full["probe"]["probe"]
    W = BLOCKS_FULL * WAVES_PER_BLOCK
    t_probe = full["probe"]["kernel_ms"] * 1e-3
    clock = p["chunk_cycles"] / W / t_probe                       # if the probes cover the waves' lifetime this is the shader clock
    att = per_step(p)
    out = {"workload": a.workload, "kernel": full["shipped"]["kernel"], "kernel_src_sha": bench.kernel_src_sha(), "rays_per_launch": full["shipped"]["rays"],
           "resident_waves": W, "full_occupancy": {"shipped_kernel_ms": round(full["shipped"]["kernel_ms"], 3), "probe_kernel_ms": round(full["probe"]["kernel_ms"], 3),
                                                   "probe_overhead": round(full["probe"]["kernel_ms"] / full["shipped"]["kernel_ms"] - 1.0, 4),
                                                   "implied_clock_ghz": round(clock / 1e9, 3), "raw": p, "per_step": att}}
    # occupancy sweep at fewer samples per texel (same rays per wave step, an eighth of the launch)
    sweep = []
    for waves_per_simd in (1, 2, 4, 8):
        cap = 256 * waves_per_simd if waves_per_simd < 8 else 0
        sh = run_child(a.workload, a.sweep_spp, False, cap)
        pr = run_child(a.workload, a.sweep_spp, True, cap)
        ps = per_step(pr["probe"])
        sweep.append({"waves_per_simd": waves_per_simd, "grid_blocks": cap or BLOCKS_FULL, "shipped_kernel_ms": round(sh["kernel_ms"], 3), "probe_kernel_ms": round(pr["kernel_ms"], 3),
                      "grays_per_s": round(sh["rays"] / sh["kernel_ms"] / 1e6, 3), "cycles_per_step": {k: (None if ps[k] is None else round(ps[k], 1)) for k in ("node_vector", "node_scalar", "leaf", "shade_per_pass")},
                      "pass_cycles": round(ps["pass_cycles"], 1), "null_region": round(ps["null_region"], 1), "steps_per_pass": ps["steps_per_pass"]})
    out["occupancy_sweep"] = {"spp": a.sweep_spp, "points": sweep}
    # chain bound: every wave-level step at its UNLOADED latency (1 wave per SIMD), all resident waves overlapping perfectly
    c1 = sweep[0]["cycles_per_step"]
    spp_full = full["shipped"]["spp"]
    chain_cycles = (p["node_vector_steps"] * (c1["node_vector"] or 0) + p["node_scalar_steps"] * (c1["node_scalar"] or 0) + p["leaf_steps"] * (c1["leaf"] or 0)
                    + p["passes"] * (c1["shade_per_pass"] or 0))
    nominal = 2.4e9
    t_chain = chain_cycles / W / nominal
    out["chain_bound"] = {"seconds": t_chain, "frac_of_shipped_kernel_time": round(t_chain / (full["shipped"]["kernel_ms"] * 1e-3), 4),
                          "note": "sum over step kinds of (wave-level steps of the full launch) x (cycles per step at ONE wave per SIMD) / %d resident waves / 2.4 GHz" % W}
    s8, s1 = sweep[-1], sweep[0]
    out["reading"] = {"rate_8_waves_over_1_wave": round(s8["grays_per_s"] / s1["grays_per_s"], 3),
                      "step_latency_8_waves_over_1_wave": {k: (None if not s1["cycles_per_step"][k] else round(s8["cycles_per_step"][k] / s1["cycles_per_step"][k], 3)) for k in s1["cycles_per_step"]}}
    txt = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
