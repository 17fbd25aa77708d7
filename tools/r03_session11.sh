#!/bin/bash
# round-3 GPU session 11: two-node scalar path A/B (default = with, nopair = without) on c4, c2, c4_scan + parity of the traversal
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s11
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_watertight.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
for W in c4 c2 c4_scan; do
  for L in default nopair; do
    if [ $L = default ]; then lib=""; else lib="TEXIR_HIP_LIB=$R/build_ab/$L.so"; fi
    v=$(env $lib timeout 400 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$W $L $v" | tee -a $out/ab.txt
  done
done
done
