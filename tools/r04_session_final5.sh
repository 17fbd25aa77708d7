#!/bin/bash
# fifth final session of round 4: product sources unchanged since the fourth (the experiments of sessions 25 - 29 live in tools/experiments/ as patches); the 4k reference-form
# test now runs its reference per texture: GPU suite on the shipped tree + the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_final5
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 3000 python -m pytest tests -m gpu -q --durations=6 > $out/pytest_gpu.txt 2>&1
tail -n 12 $out/pytest_gpu.txt | cut -c1-200
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
