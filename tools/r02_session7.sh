#!/bin/bash
# round-2 GPU session 7: wave-uniform LDS broadcast A/B, counter list, TA/TCP counters of the default kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s7
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
ab() { label=$1; shift
  for W in "${WLS[@]}"; do
    v=$(env "$@" timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])" 2>&1 | tail -1)
    echo "$label $W $v" | tee -a $out/ab.txt
  done
}
WLS=(c4 c2 c4_scan)
ab default X=1
ab ub TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_ub.so
TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_ub.so timeout 1200 python -m pytest tests/test_gpu_watertight.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_ub.txt 2>&1
tail -n 3 $out/pytest_ub.txt | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $out/counters_full.txt 2>&1 )
grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+" $out/counters_full.txt | awk '{print $NF}' | sort -u | tr '\n' ' ' > $out/counter_names.txt
wc -c $out/counter_names.txt
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 400 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload c4 --steps 1 --warmup 0 --no-cpu --no-mat > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $R/gpurun_out/r02_s7/pmc_ta.txt
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel' in r['Kernel_Name']]
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(d))
PY
  tail -n 2 /tmp/pmc_$name.log | cut -c1-300 >> $R/gpurun_out/r02_s7/pmc_ta.log
}
run ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run sq2 SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run sq3 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY
run grbm GRBM_GUI_ACTIVE
