#!/bin/bash
# session 13: Adam body with its own loads ahead of the level-1 staging, fold tile with all loads ahead of the barriers -- parity tests, step trace, Adam grid shapes
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s13
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 900 python -m pytest tests/test_gpu_tex_batch.py tests/test_gpu_optim_regressions.py -x -q > $out/pytest_batch.txt 2>&1
tail -n 5 $out/pytest_batch.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "adam or mip or tex or fold or graph" > $out/pytest_mat.txt 2>&1
tail -n 5 $out/pytest_mat.txt | cut -c1-220
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 16 $out/mat_step_trace.txt | cut -c1-130
bash tools/ab_mat.sh "default|X=1" "gy1024|TEXIR_ADAM_GRID_Y=1024" "gy512|TEXIR_ADAM_GRID_Y=512" "gy256|TEXIR_ADAM_GRID_Y=256" "default|X=1" "gy1024|TEXIR_ADAM_GRID_Y=1024" "gy512|TEXIR_ADAM_GRID_Y=512" "nobatch|TEXIR_TEX_BATCH=0" > $out/ab_adam_grid.txt 2>&1
cat $out/ab_adam_grid.txt
