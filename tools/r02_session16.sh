#!/bin/bash
# round-2 GPU session 16: counters of the scalar-path variants next to the default kernel (c4, one launch per pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s16
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() { lib=$1; name=$2; shift 2
  rm -rf /tmp/pmc_$name
  TEXIR_HIP_LIB=$lib timeout 400 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload ${WL:-c4} --steps 1 --warmup 0 --no-cpu --no-mat > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$name" <<'PY' | tee -a $R/gpurun_out/r02_s16/pmc.txt
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel' in r['Kernel_Name']]
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
ns=set((r['Dispatch_Id'], int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows)
print(sys.argv[2], "ms", [round(n/1e6,1) for _,n in ns], {k: "%.4g" % v for k,v in d.items()})
PY
}
for v in default usl usf; do
  lib=$R/build_ab/libtexir_hip_$v.so; [ $v = default ] && lib=$R/texir_code_amd/libtexir_hip.so
  run $lib ${v}_sq1 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
  run $lib ${v}_sq2 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_IFETCH SQ_ACTIVE_INST_ANY
  run $lib ${v}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum
  run $lib ${v}_sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_REQ SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL
done
