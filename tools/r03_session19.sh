#!/bin/bash
# round-3 GPU session 19: sort keys through v_min_u32 / v_max_u32 (libtexir_hip_base.so = -DTEXIR_SORT_MINMAX=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s19
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_watertight.py tests/test_gpu_scan_and_configs.py -m gpu -q -x -k "not c5 and not 4k" 2>&1 | tail -4 | tee $out/pytest.txt
for rep in 1 2; do
for cfg in "c4|--workload c4" "c2|--workload c2" "c4_scan|--workload c4_scan" "c1|--workload c1 --steps 10 --warmup 2"; do
  label=${cfg%%|*}; args=${cfg#*|}
  for lib in new base; do
    L=$R/texir_code_amd/libtexir_hip.so; [ $lib = base ] && L=$R/texir_code_amd/libtexir_hip_base.so
    v=$(TEXIR_HIP_LIB=$L timeout 400 python bench.py $args --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$label $lib $v" | tee -a $out/ab.txt
  done
done
done
for lib in new base; do
  L=$R/texir_code_amd/libtexir_hip.so; [ $lib = base ] && L=$R/texir_code_amd/libtexir_hip_base.so
  v=$(TEXIR_HIP_LIB=$L timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $lib $v" | tee -a $out/ab.txt
done
