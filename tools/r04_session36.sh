#!/bin/bash
# round-4 GPU session 36, one box (compare within this block): the 48-byte node in 64-byte slots -- three 16-byte fetches per lane and per-lane node step, origin on the
# scene's 16-bit grid, 5-bit cell exponents, the 4-wide traversal in grid units (libtexir_hip.so) -- against the 64-byte node (build_ab/libtexir_head.so = the library of commit 2a65eb2)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s36
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_watertight.py tests/test_gpu_scan_and_configs.py tests/test_gpu_tester.py -m gpu -q -x -k "not c5 and not 4k and not full_size" 2>&1 | tail -6 | tee $out/pytest.txt
run() {  # label, lib, bench args
  v=$(TEXIR_HIP_LIB=$2 timeout 400 python bench.py $3 --no-cpu --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('material_step') or {}).get('ms'))")
  echo "$1 $v" | tee -a $out/ab.txt
}
A=$R/texir_code_amd/libtexir_hip.so; B=$R/build_ab/libtexir_head.so
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1 --no-mat" "c4_scan|--workload c4_scan --steps 2 --warmup 1 --no-mat" "c1|--workload c1 --steps 20 --warmup 3 --no-mat"; do
  label=${cfg%%|*}; args=${cfg#*|}
  run "$label node64" $B "$args"
  run "$label node48" $A "$args"
done
