#!/bin/bash
# session 21: staged (load-balanced) gather in the batched launch -- bit-equality with the one-thread-per-list form, step trace
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s21
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_tex_batch.py tests/test_gpu_optim_regressions.py tests/test_gpu_parity.py tests/test_gpu_mat_step_oracle.py -q -m gpu > $out/pytest.txt 2>&1
tail -n 4 $out/pytest.txt | cut -c1-220
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 17 $out/mat_step_trace.txt | cut -c1-150
bash tools/ab_mat.sh "default|X=1" "default|X=1" > $out/ab.txt 2>&1
cat $out/ab.txt
