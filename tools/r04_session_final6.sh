#!/bin/bash
# sixth final session of round 4 (quad leaves + weight 3 for coherent scenes: new IrT kernel sources): GPU suite, PMC passes of c4 / c2 / c4_scan / c1, kernel-trace stats of
# the default bench, material-step trace + PMC, bench lines (default incl. c4_scan, c2, c1, 2 ranks over gloo on this one GPU)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_final6
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 3000 python -m pytest tests -m gpu -q --durations=6 > $out/pytest_gpu.txt 2>&1
tail -n 12 $out/pytest_gpu.txt | cut -c1-200
bash tools/profile_round.sh r04_final6/prof c4 c2 c4_scan c1 > $out/profile_round.log 2>&1
tail -n 1 $out/profile_round.log | cut -c1-300
cp $R/profiles/pmc_c4.json $R/profiles/pmc_c2.json $R/profiles/pmc_c4_scan.json $R/profiles/pmc_c1.json $out/ 2>/dev/null
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 3 $out/mat_step_trace.txt | cut -c1-110
bash tools/mat_step_pmc.sh r04_final6/matpmc > $out/mat_pmc.log 2>&1
cp $R/profiles/pmc_mat_step.json $out/ 2>/dev/null
head -n 1 $out/mat_pmc.log | cut -c1-300
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json | cut -c1-300
timeout 600 python bench.py --workload c2 > $out/bench_c2.json 2>> $out/bench_default.err
timeout 600 python bench.py --workload c1 --steps 5 --warmup 1 > $out/bench_c1.json 2>> $out/bench_default.err
TEXIR_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --workload c2 --steps 2 --warmup 1 --no-cpu --mat --mat-steps 10 > $out/bench_2rank_gloo_one_gpu.json 2>> $out/bench_default.err
tail -n 1 $out/bench_2rank_gloo_one_gpu.json | cut -c1-300
