#!/bin/bash
# round-3 GPU session 20: host-side review fixes of the material step (optimiser groups fixed at construction, betas in the device record, tap-list budget gives back
# dropped views): the suites that cover it, the step's PMC profile and a bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s20
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1800 python -m pytest tests/test_gpu_optim_regressions.py tests/test_gpu_trainer.py tests/test_gpu_scan_and_configs.py tests/test_gpu_parity.py -m gpu -q -k "not c5" 2>&1 | tail -4 | tee $out/pytest.txt
bash tools/mat_step_pmc.sh r03_s20/matpmc > $out/mat_pmc.log 2>&1
head -n 1 $out/mat_pmc.log | cut -c1-300
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 2 $out/mat_step_trace.txt | cut -c1-110
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json | cut -c1-400
