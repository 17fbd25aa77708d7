#!/bin/bash
# round-3 GPU session 10: material step with the arena fill + tick on a parallel graph branch (A/B), trainer / graph tests, material-step PMC
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s10
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
for rep in 1 2 3; do
for cfg in "fork|" "nofork|TEXIR_GRAPH_FORK=0"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $label $v" | tee -a $out/mat_ab.txt
done
done
timeout 1500 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_scan_and_configs.py tests/test_gpu_optim_regressions.py -m gpu -q -k "graph or runner or 4k or trajectory or accumulation or edits or dense" 2>&1 | tail -5
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 26 $out/mat_step_trace.txt | cut -c1-130
bash tools/mat_step_pmc.sh r03_s10/matpmc > $out/mat_pmc.log 2>&1
tail -n 30 $out/mat_pmc.log | cut -c1-200
