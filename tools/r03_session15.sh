#!/bin/bash
# round-3 GPU session 15: the single-pass specular forward at 8 waves per SIMD (spec_single_kernel) vs the general kernel (TEXIR_SPEC_SINGLE=0): parity + material step
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s15
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py tests/test_gpu_tester.py tests/test_gpu_scan_and_configs.py tests/test_gpu_raster.py -m gpu -q -k "spec or trainer or runner or graph or 4k or tester or raster or render" 2>&1 | tail -4
for rep in 1 2 3; do
for cfg in "single|" "general|TEXIR_SPEC_SINGLE=0"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $label $v" | tee -a $out/mat_ab.txt
done
done
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
grep -E "spec_|kernels " $out/mat_step_trace.txt | cut -c1-120
