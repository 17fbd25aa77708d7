#!/bin/bash
# (session 13 and tools/trace_mat_pipeline.sh ran on a tree with tools/experiments/pipelined_material_step.patch applied: the pipelined step is not in the product)
# round-3 GPU session 13: the pipelined material step -- 4k-texture trajectory test (eager == graph == split == pipelined), bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s13
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_scan_and_configs.py -m gpu -q -x -k "4k" 2>&1 | tail -15
for rep in 1 2; do
for cfg in "pipelined|" "plain|TEXIR_MAT_PIPELINE=0" "pipe_g32|TEXIR_PIPE_ADAM_GRID_Y=32" "pipe_g64|TEXIR_PIPE_ADAM_GRID_Y=64" "pipe_gfull|TEXIR_PIPE_ADAM_GRID_Y=4096"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['material_step'].get('pipelined'))" 2>&1 | tail -1)
  echo "mat $label $v" | tee -a $out/mat_ab.txt
done
done
tail -5 $out/err.txt | cut -c1-300
