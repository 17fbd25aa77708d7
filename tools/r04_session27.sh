#!/bin/bash
# round-4 GPU session 27, one box: what do the hit shader's two dependent fetches (corner uvs, then the radiance tile) cost a pass?  build_ab/libtexir_noshade.so
# (-DTEXIR_PROBE_NOSHADE=1: the shader returns (u, v, slot) without touching memory) against the shipped library: the difference bounds what overlapping the
# shading of pass p with the traversal of pass p + 1 could buy
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s27
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() {  # label, lib, bench args
  v=$(TEXIR_HIP_LIB=$2 timeout 400 python bench.py $3 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$1 $v" | tee -a $out/ab.txt
}
A=$R/texir_code_amd/libtexir_hip.so; B=$R/build_ab/libtexir_noshade.so
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1" "c4_scan|--workload c4_scan --steps 2 --warmup 1"; do
  label=${cfg%%|*}; args=${cfg#*|}
  run "$label shipped" $A "$args"
  run "$label noshade" $B "$args"
done
