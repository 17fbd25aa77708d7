#!/bin/bash
# usage: tools/pmc_pass.sh <out-dir-under-gpurun_out> <workload> [env...]   -- one rocprofv3 --pmc pass per counter group, each under timeout
out=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$out
run() { name=$1; shift
  env "${EXTRA[@]}" timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu --no-mat --extra none --no-project > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$R/gpurun_out/$out/pmc_$name.csv" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel<false' in r['Kernel_Name'] or 'irt_kernel<false' in r['Kernel_Name'] or 'irt_stream_kernel<false' in r['Kernel_Name']]
w=csv.DictWriter(open(sys.argv[2],'w'),fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
import collections
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
print(rows[0]['Kernel_Name'][:60], rows[0]['VGPR_Count'], dict(d))
PY
}
EXTRA=("$@")
run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS
run waves SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run tcc TCC_HIT_sum TCC_MISS_sum
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
