#!/bin/bash
# round-3 GPU session 2: phase-scheduling weights x azimuth-wedge parts x per-XCD chunk counters (A/B on c4, c4_scan, c2) + the new parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s2
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
for W in c4 c4_scan c2; do
  for L in default sched0 w2old w2wedge w2xcd w3xcd s0xcd; do
    if [ $L = default ]; then lib=""; else lib="TEXIR_HIP_LIB=$R/build_ab/$L.so"; fi
    v=$(env $lib timeout 400 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$W $L $v" | tee -a $out/ab.txt
  done
done
timeout 1500 python -m pytest tests/test_gpu_optim_regressions.py tests/test_gpu_scan_and_configs.py -m gpu -q -k "not c5" > $out/pytest_new.txt 2>&1
tail -n 30 $out/pytest_new.txt | cut -c1-250
