#!/bin/bash
# round-3 GPU session 1: per-step phase scheduling (TEXIR_SCHED) -- parity suite, then A/B of while-while / majority (node weight 1, 2) on c4, c2, c4_scan
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s1
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1
tail -n 3 $out/pytest_gpu.txt | cut -c1-200
for W in c4 c4_scan c2; do
  for L in default sched0 w2; do
    if [ $L = default ]; then lib=""; else lib="TEXIR_HIP_LIB=$R/build_ab/$L.so"; fi
    v=$(env $lib timeout 400 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$W $L $v" | tee -a $out/ab.txt
  done
done
for L in default sched0; do
  if [ $L = default ]; then lib=""; else lib="TEXIR_HIP_LIB=$R/build_ab/$L.so"; fi
  for W in c4 c4_scan; do
    echo "== $L $W" >> $out/stats.txt
    env $lib timeout 300 python tools/irt_stats.py $W 262144 >> $out/stats.txt 2>>$out/err.txt
  done
done
cat $out/stats.txt
