#!/bin/bash
# round-3 GPU session 6: material step A/B -- shifts read from pinned host memory vs staged copy, optimiser step inside vs after the graph; trainer tests
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s6
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
for rep in 1 2; do
for cfg in "default|" "copy|TEXIR_SHIFT_ZEROCOPY=0" "eagerstep|TEXIR_GRAPH_STEP=0" "copy+eagerstep|TEXIR_SHIFT_ZEROCOPY=0 TEXIR_GRAPH_STEP=0"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $label $v" | tee -a $out/mat_ab.txt
done
done
timeout 1500 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_scan_and_configs.py -m gpu -q -k "graph or runner or 4k or trajectory" 2>&1 | tail -5
