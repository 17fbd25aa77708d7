#!/bin/bash
# round-2 GPU session 31: final profiles and bench lines of the round (kernels of commit adc5667): gpu suite, PMC passes of c4 / c2 / c4_scan / c1,
# bench default (+ c4_scan), c2, c1, the default under rocprofv3 --kernel-trace --stats, material-step trace
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s31
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 2 $out/pytest_gpu.txt | cut -c1-200
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 2 $out/mat_step_trace.txt | cut -c1-110
bash tools/profile_round.sh r02_s31/prof c4 c2 c4_scan c1 > $out/profile_round.log 2>&1
tail -n 2 $out/profile_round.log | cut -c1-300
cp $R/profiles/pmc_c4.json $R/profiles/pmc_c2.json $R/profiles/pmc_c4_scan.json $R/profiles/pmc_c1.json $out/ 2>/dev/null
cp $out/prof/c4_kernel_stats.csv $out/prof/bench_default_under_rocprof.json $out/ 2>/dev/null
timeout 900 python bench.py --extra c4_scan > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json | cut -c1-300
timeout 600 python bench.py --workload c2 > $out/bench_c2.json 2>> $out/bench_default.err
timeout 600 python bench.py --workload c1 --steps 5 --warmup 1 > $out/bench_c1.json 2>> $out/bench_default.err
