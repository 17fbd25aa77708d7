#!/bin/bash
# session 16: defaults = non-temporal Adam at 5 waves, masked reads by select -- parity tests, step trace, latency / period
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s16
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 900 python -m pytest tests/test_gpu_tex_batch.py tests/test_gpu_optim_regressions.py -x -q > $out/pytest_batch.txt 2>&1
tail -n 3 $out/pytest_batch.txt | cut -c1-220
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 17 $out/mat_step_trace.txt | cut -c1-150
bash tools/ab_mat.sh "default|X=1" "default|X=1" "default|X=1" > $out/ab.txt 2>&1
cat $out/ab.txt
