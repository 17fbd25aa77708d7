#!/bin/bash
# round-2 GPU session 27: pre-flight of the driver's round-end sequence (gpu suite x2, smoke, default bench, --gpus 2 refusal)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s27
mkdir -p $out
cd $R
for i in 1 2; do timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_$i.txt 2>&1; tail -n 1 $out/pytest_$i.txt; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -n 2 $out/smoke.txt
( time timeout 1200 python bench.py ) > $out/bench_noflags.json 2> $out/bench_noflags.err; tail -n 1 $out/bench_noflags.json | cut -c1-300; tail -n 4 $out/bench_noflags.err
timeout 300 python bench.py --gpus 2 > $out/bench_gpus2.txt 2>&1; echo "rc=$?" >> $out/bench_gpus2.txt; tail -n 2 $out/bench_gpus2.txt
