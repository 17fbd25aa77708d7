#!/bin/bash
# round-3 GPU session 3: full GPU suite on the new tree (phase scheduling w2 + per-XCD chunk counters + wedge parts, optimiser step inside the
# hipGraph, new parity tests incl. full-size c5 / c1 / c4_scan), then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s3
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 3000 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 15 $out/pytest_gpu.txt | cut -c1-250
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 1500 $out/bench_default.json
tail -n 5 $out/bench_default.err
