#!/bin/bash
# round-4 GPU session 6 (one box, IrT only): triangles per leaf (TEXIR_MAX_LEAF = 2 default, 3, 4: the leaf step got 18 instructions cheaper per triangle
# in session 4, which moves the node / leaf balance) and hardware sin / cos in the fused sampling (build_ab/libtexir_fastsincos.so) re-measured
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s6
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() {  # label, lib, max_leaf, bench args
  v=$(TEXIR_HIP_LIB=$2 TEXIR_MAX_LEAF=$3 timeout 400 python bench.py $4 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['scene']['inner_nodes'])" 2>&1 | tail -1)
  echo "$1 $v" | tee -a $out/ab.txt
}
A=$R/texir_code_amd/libtexir_hip.so; B=$R/build_ab/libtexir_fastsincos.so
timeout 300 python tools/graph_branch_probe.py 2>&1 | tail -6 | tee $out/graph_branch.txt
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1" "c4_scan|--workload c4_scan --steps 2 --warmup 1"; do
  label=${cfg%%|*}; args=${cfg#*|}
  for ml in 2 3 4; do run "$label leaf$ml" $A $ml "$args"; done
  run "$label leaf2_fastsincos" $B 2 "$args"
  run "$label leaf2" $A 2 "$args"
done
