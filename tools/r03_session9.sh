#!/bin/bash
# round-3 GPU session 9: time-dependent scheduling weight (early weight / late weight / early steps) A/B on c4, c4_scan, c2
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s9
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
for W in c4 c4_scan c2; do
  for L in default w22 w21_24 w31_16 w31_32 w41_24 w11; do
    if [ $L = default ]; then lib=""; else lib="TEXIR_HIP_LIB=$R/build_ab/$L.so"; fi
    v=$(env $lib timeout 400 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$W $L $v" | tee -a $out/ab.txt
  done
done
