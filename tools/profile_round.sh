#!/bin/bash
# usage: tools/profile_round.sh <tag> [workloads...]   (default: c4 c2)
# kernel-trace stats of the default bench + PMC passes of the IrT kernel per workload; results -> gpurun_out/<tag>/ and
# profiles/pmc_<workload>.json (the counters bench.py's roofline divides by its live kernel time; stamped with the kernel-source hash)
tag=${1:-prof}; shift
WLS=("$@"); [ ${#WLS[@]} -eq 0 ] && WLS=(c4 c2)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export TEXIR_SYNTH_CACHE=${TEXIR_SYNTH_CACHE:-/tmp/texir_synth}
run() { wl=$1; name=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 400 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu --no-mat --extra none --no-project > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$out/$wl/pmc_$name.csv" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel<false' in r['Kernel_Name'] or 'irt_kernel<false' in r['Kernel_Name'] or 'irt_stream_kernel<false' in r['Kernel_Name']]
w=csv.DictWriter(open(sys.argv[2],'w'),fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
print(rows[0]['Kernel_Name'][:60], dict(d))
PY
}
for wl in "${WLS[@]}"; do
  mkdir -p $out/$wl
  run $wl valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS
  run $wl waves SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run $wl tcc TCC_HIT_sum TCC_MISS_sum
  run $wl rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  run $wl write WRITE_SIZE
  run $wl grbm GRBM_GUI_ACTIVE
  run $wl tccbusy TCC_BUSY_sum TCC_CYCLE_sum
  run $wl tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
  run $wl tcpgate TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum
  run $wl sqc SQ_INSTS_SMEM SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VALU_CVT
  python $R/tools/pmc_to_json.py $out/$wl $wl $R/gpurun_out/$tag/pmc_$wl.json
done
# kernel-trace stats of the default bench run (the headline), with the PMC json in place so that the line carries the measured bounds
mkdir -p $R/profiles; cp $out/pmc_*.json $R/profiles/ 2>/dev/null
rm -rf /tmp/kt
TEXIR_BENCH_FULL=$out/bench_default_under_rocprof_full.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 2 --warmup 1 --extra none --no-project --no-e2e > $out/bench_default_under_rocprof.json 2> $out/bench_default.err
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/c4_kernel_stats.csv
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && (head -1 "$f"; grep "irt_" "$f") > $out/c4_irt_kernel_trace_rows.csv
tail -1 $out/bench_default_under_rocprof.json
