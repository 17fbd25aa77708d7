#!/bin/bash
# round-4 GPU session 26, one box: counters of the unified-record kernel (libtexir_hip.so) against the pair-sort kernel (build_ab/libtexir_urec0.so) on c2,
# and the wave-level step counts of both (tools/irt_stats.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s26
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
pass() { lib=$1; name=$2; shift 2
  rm -rf /tmp/pmc_$name
  TEXIR_HIP_LIB=$lib timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload c2 --steps 1 --warmup 0 --no-cpu --no-mat --extra none > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel<false' in r['Kernel_Name']]
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(d))
PY
}
for v in urec0 urec1; do
  lib=$R/build_ab/libtexir_urec0.so; [ $v = urec1 ] && lib=$R/texir_code_amd/libtexir_hip.so
  echo "== $v" | tee -a $out/pmc.txt
  pass $lib ${v}_valu SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM | tee -a $out/pmc.txt
  pass $lib ${v}_waves SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU | tee -a $out/pmc.txt
  pass $lib ${v}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum | tee -a $out/pmc.txt
  pass $lib ${v}_rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum | tee -a $out/pmc.txt
  echo "== stats $v" | tee -a $out/stats.txt
  (cd $R && TEXIR_HIP_LIB=$lib timeout 300 python tools/irt_stats.py c2 262144 2>>$out/err.txt | tee -a $out/stats.txt)
done
