#!/bin/bash
# round-4 GPU session 35, one box: what does ONE 16-byte fetch per lane and per-lane node step cost?  build_ab/libtexir_extra.so (-DTEXIR_PROBE_NODE_EXTRA_LOAD=1: a fifth
# fetch of the node's first word, same line, result discarded) against the shipped library -- the exchange rate a 48-byte node would be traded at
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s35
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() {  # label, lib, bench args
  v=$(TEXIR_HIP_LIB=$2 timeout 400 python bench.py $3 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$1 $v" | tee -a $out/ab.txt
}
A=$R/texir_code_amd/libtexir_hip.so; B=$R/build_ab/libtexir_extra.so
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1" "c4_scan|--workload c4_scan --steps 2 --warmup 1"; do
  label=${cfg%%|*}; args=${cfg#*|}
  run "$label shipped" $A "$args"
  run "$label extra_load" $B "$args"
done
