#!/bin/bash
# round-4 GPU session 32, one box: leaf sizes re-measured with quad leaves (TEXIR_MAX_LEAF = 2 default, 3, 4: a leaf of four triangles is two 48-byte records)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s32
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() {  # label, max leaf, bench args
  v=$(TEXIR_MAX_LEAF=$2 timeout 400 python bench.py $3 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$1 $v" | tee -a $out/ab.txt
}
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c4_scan|--workload c4_scan --steps 2 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1"; do
  label=${cfg%%|*}; args=${cfg#*|}
  for ml in 2 3 4; do run "$label max_leaf$ml" $ml "$args"; done
done
