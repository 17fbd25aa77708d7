#!/bin/bash
# round-4 GPU session 5: (i) counters of the refill (stream) kernel against the lock-step kernel at 256 spp on c4_scan and c4: why it is 4x slower;
# (ii) the specular forward against a plain one-ray-per-thread trace of the same rays in pixel-major and sample-major order (tools/spec_split_probe.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s5
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 600 python -m pytest tests/test_gpu_scan_and_configs.py -m gpu -q -x -k "refill or never_blocks" 2>&1 | tail -15 | tee $out/pytest.txt
timeout 600 python tools/spec_split_probe.py 2>&1 | tail -3 | tee $out/spec_split.txt
cd /tmp && export TMPDIR=/tmp
pmc() { # label env workload counters...
  label=$1; envs=$2; wl=$3; shift 3
  rm -rf /tmp/pmc_x
  env $envs timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_x -- python $R/bench.py --workload $wl --spp 256 --steps 1 --warmup 0 --no-cpu --no-mat --extra none > /tmp/pmc_x.log 2>&1
  f=$(find /tmp/pmc_x -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$label $wl" <<'PY' | tee -a $out/pmc.txt
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel<false' in r['Kernel_Name'] or 'irt_stream_kernel<false' in r['Kernel_Name']]
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
print(sys.argv[2], rows[0]['Kernel_Name'][:40] if rows else '-', {k: round(v) for k, v in d.items()})
PY
}
for wl in c4_scan c4; do
  for cfg in "lockstep|TEXIR_IRT_REFILL=0" "refill32|TEXIR_IRT_REFILL=32"; do
    label=${cfg%%|*}; envs=${cfg#*|}
    v=$(env $envs timeout 300 python $R/bench.py --workload $wl --spp 256 --steps 3 --warmup 1 --no-cpu --no-mat --extra none 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])")
    echo "$label $wl 256spp: $v" | tee -a $out/pmc.txt
    pmc $label "$envs" $wl SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS
    pmc $label "$envs" $wl SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
    pmc $label "$envs" $wl TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
    pmc $label "$envs" $wl TCC_HIT_sum TCC_MISS_sum
    pmc $label "$envs" $wl SQ_INSTS_SMEM SQC_DCACHE_REQ SQC_DCACHE_HITS
  done
done
