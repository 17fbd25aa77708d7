#!/bin/bash
# round-3 GPU session 12: specular forward at forced occupancies (5 / 6 waves per SIMD) vs the shipped 4; the IRRF runner test after the --is_continue fix
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s12
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 900 python -m pytest tests/test_nirf.py -m gpu -q 2>&1 | tail -2
for rep in 1 2; do
for cfg in "default|" "spec5|TEXIR_HIP_LIB=$R/build_ab/spec5.so" "spec6|TEXIR_HIP_LIB=$R/build_ab/spec6.so"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $label $v" | tee -a $out/mat_ab.txt
done
done
