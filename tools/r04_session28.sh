#!/bin/bash
# round-4 GPU session 28, one box (compare within this block): irt_group_kernel shading the hits of 4 passes together (libtexir_hip.so; build_ab/libtexir_sb2.so:
# 2 passes) against pass-by-pass shading (build_ab/libtexir_head.so = the library of commit b1ee5ed)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s28
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_watertight.py tests/test_gpu_scan_and_configs.py -m gpu -q -x -k "not c5 and not 4k and not full_size" 2>&1 | tail -6 | tee $out/pytest.txt
run() {  # label, lib, bench args
  v=$(TEXIR_HIP_LIB=$2 timeout 400 python bench.py $3 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$1 $v" | tee -a $out/ab.txt
}
A=$R/texir_code_amd/libtexir_hip.so; B=$R/build_ab/libtexir_head.so; C=$R/build_ab/libtexir_sb2.so
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1" "c4_scan|--workload c4_scan --steps 2 --warmup 1" "c1|--workload c1 --steps 20 --warmup 3"; do
  label=${cfg%%|*}; args=${cfg#*|}
  run "$label head" $B "$args"
  run "$label batch4" $A "$args"
  run "$label batch2" $C "$args"
done
