"""Fold the per-pass counter CSVs of tools/profile_round.sh into profiles/pmc_<workload>.json (read by bench.py's roofline):
usage: python tools/pmc_to_json.py <dir with pmc_*.csv> <workload> <out.json>"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d, wl, out = sys.argv[1], sys.argv[2], sys.argv[3]
c = collections.defaultdict(float)
kernel, dur = None, []
for f in sorted(glob.glob(os.path.join(d, "pmc_*.csv"))):
    for r in csv.DictReader(open(f)):
        kernel = r["Kernel_Name"].split("(")[0]
        c[r["Counter_Name"]] += float(r["Counter_Value"])
    rows = list(csv.DictReader(open(f)))
    if rows:
        dur.append((int(rows[0]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6)
res = {"workload": wl, "kernel": kernel, "kernel_src_sha": bench.kernel_src_sha(),
       "source": "rocprofv3 --pmc, separate passes, one launch each (tools/profile_round.sh -> %s)" % os.path.relpath(d, ROOT),
       "counters": dict(c), "kernel_ms_under_pmc": dur}
rd = c.get("TCC_EA0_RDREQ_sum", 0.0)
if rd:
    # read requests priced at their own size where the split counters were collected (98.4 % are 128-byte lines on c4; round 5 priced the 64-byte
    # ones at 128 B too: +0.8 %); without the split every request is taken as a 128-byte line.  WRITE_SIZE is in KB
    r128, r64, r32 = c.get("TCC_EA0_RDREQ_128B_sum"), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    if r128:
        rd_bytes = r128 * 128.0 + r64 * 64.0 + r32 * 32.0 + max(0.0, rd - r128 - r64 - r32) * 128.0
    else:
        rd_bytes = rd * 128.0
    res["fabric_read_bytes_per_launch"] = rd_bytes
    res["fabric_bytes_per_launch"] = rd_bytes + c.get("WRITE_SIZE", 0.0) * 1024.0
if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
    res["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
if c.get("SQ_INSTS_VALU", 0):
    res["SQ_INSTS_VALU"] = c["SQ_INSTS_VALU"]
    if c.get("SQ_THREAD_CYCLES_VALU", 0):
        res["valu_lane_utilisation"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]), 4)
if c.get("GRBM_GUI_ACTIVE", 0) and dur:
    res["effective_clock_ghz"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / (dur[0] * 1e-3) / 1e9, 3)      # (the counter sums the 8 XCDs)
if c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) and c.get("TCP_GATE_EN1_sum", 0):
    # vector-L1 tag lookups per TCP clock (the 256 TCPs' busy clocks summed): the traversal's other limiter
    res["tcp_cache_accesses"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    res["tcp_clocks"] = c["TCP_GATE_EN1_sum"]                      # summed over the 256 CUs' L1s
    res["tcp_accesses_per_clk"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["TCP_GATE_EN1_sum"], 4)
if c.get("SQ_WAVE_CYCLES", 0):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in c:
            res[k.lower() + "_frac"] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
if c.get("SQC_DCACHE_REQ", 0):
    # scalar path of the traversal (wave-uniform node steps fetch through the scalar cache)
    res["smem_insts"] = c.get("SQ_INSTS_SMEM", 0.0)
    res["scalar_cache_hit_rate"] = round(c.get("SQC_DCACHE_HITS", 0.0) / c["SQC_DCACHE_REQ"], 4)
# rays of the profiled launch = valid texels x spp of the workload (every PMC pass runs `bench.py --workload <wl> --steps 1`)
try:
    T, r_, tex_, spp_, style_ = bench.WORKLOADS[wl]
    _, _, _, valid_, _, _, _ = bench.make_workload(wl)
    res["rays_per_launch"] = int((valid_.reshape(-1) > 0).sum()) * spp_
except Exception as e:      # (the json is still usable at N = 1)
    res["rays_per_launch_error"] = str(e)
# the traversal's own counters on a fixed slice of the workload (bench.irt_fingerprint): bench.py recomputes them live and compares -- the cross-check between
# the box these PMC counters come from and the box a bench line is timed on
try:
    import torch
    from texir_code_amd import scene as S, dist_util
    if torch.cuda.is_available():
        sc0_, pos_, nrm_, valid_, shift_, res_, spp_ = bench.make_workload(wl)
        dev = torch.device("cuda", 0)
        sc_ = S.Scene(sc0_["verts"], sc0_["tris"], sc0_["tri_uvs"], sc0_["hdr"], device=0)
        ids_ = dist_util.morton_order(torch.nonzero(torch.from_numpy(valid_.reshape(-1)) > 0)[:, 0].to(torch.int32), res_)
        d_pos, d_nrm, d_shift = (torch.from_numpy(a).to(dev) for a in (pos_.reshape(-1, 3), nrm_.reshape(-1, 3), shift_))
        sc_.irt_generate(d_pos, d_nrm, d_shift, spp_, "uniform", texel_ids=ids_.to(dev))        # (the full-list launch that tunes the scene's scheduler weight, exactly as bench.py's warm-up does)
        res["fingerprint"] = bench.irt_fingerprint(sc_, d_pos, d_nrm, d_shift, ids_, spp_, dev)
        res["tex_layout"] = sc_.texture_layout()
except Exception as e:
    res["fingerprint_error"] = "%s: %s" % (type(e).__name__, e)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "counters"}))
