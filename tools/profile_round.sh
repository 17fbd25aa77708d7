#!/bin/bash
# usage: tools/profile_round.sh <tag>    -- kernel-trace stats of the default bench + PMC passes of the IrT kernel (c4); results -> gpurun_out/<tag>/
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 2 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/c4_kernel_stats.csv
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && (head -1 "$f"; grep "irt_" "$f") > $out/c4_irt_kernel_trace_rows.csv
EXTRA=()
run() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload c4 --steps 1 --warmup 0 --no-cpu --no-mat > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$out/pmc_$name.csv" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel' in r['Kernel_Name'] or 'irt_kernel' in r['Kernel_Name']]
w=csv.DictWriter(open(sys.argv[2],'w'),fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
print(rows[0]['Kernel_Name'][:60], dict(d))
PY
}
run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS
run waves SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run tcc TCC_HIT_sum TCC_MISS_sum
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tccbusy TCC_BUSY_sum TCC_CYCLE_sum
tail -1 $out/bench_default.json
