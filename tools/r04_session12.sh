#!/bin/bash
# session 12: batched texture launches + mask-read gradient stacks -- new parity tests, the suites that exercise the material step, step trace and A/B of the switches
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s12
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 900 python -m pytest tests/test_gpu_tex_batch.py -x -q > $out/pytest_batch.txt 2>&1
tail -n 15 $out/pytest_batch.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_optim_regressions.py tests/test_gpu_mat_step_oracle.py tests/test_gpu_trainer.py tests/test_gpu_parity.py -q -m gpu > $out/pytest_mat.txt 2>&1
tail -n 15 $out/pytest_mat.txt | cut -c1-220
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 20 $out/mat_step_trace.txt | cut -c1-130
bash tools/ab_mat.sh "default|X=1" "nomask|TEXIR_GRAD_MASK=0" "nobatch|TEXIR_TEX_BATCH=0" "default|X=1" "nomask|TEXIR_GRAD_MASK=0" "nobatch|TEXIR_TEX_BATCH=0" > $out/ab_batch_mask.txt 2>&1
cat $out/ab_batch_mask.txt
