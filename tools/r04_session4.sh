#!/bin/bash
# round-4 GPU session 4, one box, IrT only (compare within this block):
#  (i)   leaf steps specialised for the wave's common dominant axis + 4-register shear (libtexir_hip.so) against per-lane selects everywhere
#        (build_ab/libtexir_kz0.so = -DTEXIR_LEAF_UNIFORM_KZ=0)
#  (ii)  memory order of the 4-wide tree: TEXIR_BVH_LAYOUT = 0 depth-first, 1 sibling blocks, 2 / 3 treelets
#  (iii) compaction by refill (irt_stream_kernel): TEXIR_IRT_REFILL = lanes that must be idle before the idle lanes take their next ray
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s4
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_watertight.py tests/test_gpu_scan_and_configs.py -m gpu -q -x -k "not c5 and not 4k and not full_size" 2>&1 | tail -4 | tee $out/pytest.txt
run() {  # label, lib, layout, refill, bench args
  v=$(TEXIR_HIP_LIB=$2 TEXIR_BVH_LAYOUT=$3 TEXIR_IRT_REFILL=$4 timeout 400 python bench.py $5 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])" 2>&1 | tail -1)
  echo "$1 $v" | tee -a $out/ab.txt
}
A=$R/texir_code_amd/libtexir_hip.so; B=$R/build_ab/libtexir_kz0.so
for rep in 1; do
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1" "c4_scan|--workload c4_scan --steps 2 --warmup 1"; do
  label=${cfg%%|*}; args=${cfg#*|}
  run "$label kz0_layout0" $B 0 0 "$args"
  for lay in 0 1 2 3; do run "$label kz1_layout$lay" $A $lay 0 "$args"; done
  for rf in 48 32 16; do run "$label kz1_layout0_refill$rf" $A 0 $rf "$args"; done
done
done
for rf in 0 32 16; do
  for W in c4_scan c4; do
    echo "== stats $W refill $rf" | tee -a $out/stats.txt
    TEXIR_IRT_REFILL=$rf timeout 300 python tools/irt_stats.py $W 262144 2>>$out/err.txt | tee -a $out/stats.txt
  done
done
