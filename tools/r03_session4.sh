#!/bin/bash
# round-3 GPU session 4: failing tests re-run, raster-oracle GPU tests, material-step trace (optimiser step inside the graph) and its A/B against
# the while-while traversal, specular lanes-per-pixel sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s4
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_scan_and_configs.py tests/test_gpu_parity.py -m gpu -q -s -k "raster or 4k or fused_adam or c5_joint_pipeline_full" > $out/pytest_sel.txt 2>&1
grep -E "pixels pick|passed|failed|^E  " $out/pytest_sel.txt | cut -c1-300 | tail -20
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 30 $out/mat_step_trace.txt | cut -c1-150
for cfg in "default|" "sched0|TEXIR_HIP_LIB=$R/build_ab/s0.so" "lpp4|TEXIR_SPEC_LPP=4" "lpp8|TEXIR_SPEC_LPP=8" "lpp1|TEXIR_SPEC_LPP=1"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $label $v" | tee -a $out/mat_ab.txt
done
