#!/bin/bash
# round-3 GPU session 17: the scheduler weight decided per scene from the measured fullness of its node steps (default) vs forced weights
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s17
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scan_and_configs.py tests/test_gpu_scale.py -m gpu -q -k "irt or scan or c1 or c4 or c2 or full_size" 2>&1 | tail -3
for W in c4 c4_scan c2; do
  for cfg in "auto|" "w2|TEXIR_SCHED_WEIGHT=2" "w1|TEXIR_SCHED_WEIGHT=1"; do
    label=${cfg%%|*}; envs=${cfg#*|}
    v=$(env $envs timeout 400 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); sc=d['config']['scene']; print(d['value'], d['ms_per_step'], 'weight', sc.get('sched_weight'), 'fill', sc.get('node_step_fill'))" 2>&1 | tail -1)
    echo "$W $label $v" | tee -a $out/ab.txt
  done
done
