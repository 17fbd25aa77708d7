#!/bin/bash
# round-4 GPU session 33, one box: the phase scheduler's weight re-measured with quad leaves (a leaf visit is one record step now): TEXIR_SCHED_WEIGHT = 1, 2, 3, 4 and unset
# (the scene's own: 2, or 1 where texir_scene_tune finds the node steps less than 60 % full)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s33
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() {  # label, weight, bench args
  v=$(TEXIR_SCHED_WEIGHT=$2 timeout 400 python bench.py $3 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$1 $v" | tee -a $out/ab.txt
}
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c4_scan|--workload c4_scan --steps 2 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1"; do
  label=${cfg%%|*}; args=${cfg#*|}
  for w in 0 1 2 3 4; do run "$label weight$w" $w "$args"; done
done
