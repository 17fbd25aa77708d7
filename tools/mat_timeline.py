"""GPU timeline of the Mat stage from a rocprofv3 kernel trace: per training step (delimited by the fused Adam launch) the span of its kernels, their
busy time, and the idle gap to the next step -- where a stage whose wall time exceeds steps x GPU period loses it.
usage: python tools/mat_timeline.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys

import numpy as np


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_tex" in r[2]]
    steps = []
    prev = None
    for a in adam:
        first = (prev + 1) if prev is not None else max(0, a - 14)
        # a step's kernels: everything after the previous Adam whose start is within 5 ms of this Adam (validation forwards / plots sit further away)
        ks = [rows[i] for i in range(first, a + 1) if rows[a][0] - rows[i][0] < 5_000_000]
        steps.append({"start": ks[0][0], "end": rows[a][1], "busy": sum(k[1] - k[0] for k in ks), "n": len(ks), "has_spec": any("spec_kernel" in k[2] for k in ks)})
        prev = a
    out = {"adam_launches": len(adam)}
    for name, sel in (("stage0", [s for s in steps if not s["has_spec"]]), ("stage12", [s for s in steps if s["has_spec"]])):
        if len(sel) < 3:
            continue
        span = np.array([s["end"] - s["start"] for s in sel]) / 1e3
        busy = np.array([s["busy"] for s in sel]) / 1e3
        period = np.diff(np.array([s["start"] for s in sel])) / 1e3
        gap = (np.array([b["start"] for b in sel[1:]]) - np.array([a["end"] for a in sel[:-1]])) / 1e3
        q = lambda x: [round(float(np.percentile(x, p)), 1) for p in (10, 50, 90, 99)]
        out[name] = {"steps": len(sel), "kernels_per_step_median": int(np.median([s["n"] for s in sel])), "span_us_p10_50_90_99": q(span), "busy_us_p10_50_90_99": q(busy),
                     "period_us_p10_50_90_99": q(period), "gap_us_p10_50_90_99": q(gap), "gap_over_100us_frac": round(float((gap > 100).mean()), 3),
                     "sum_span_s": round(float(span.sum() / 1e6), 3), "sum_gap_s": round(float(gap.sum() / 1e6), 3)}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
