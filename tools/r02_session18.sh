#!/bin/bash
# round-2 GPU session 18: new default (scalar float-node path) -- full gpu suite, A/B against the per-lane build, material step, bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s18
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 8 $out/pytest_gpu.txt | cut -c1-200
ab() { label=$1; shift
  for W in "${WLS[@]}"; do
    v=$(env "$@" timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])" 2>&1 | tail -1)
    echo "$label $W $v" | tee -a $out/ab.txt
  done
}
WLS=(c4 c2 c4_scan c1)
ab default X=1
ab nosl TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_nosl.so
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm default X=1
abm nosl TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_nosl.so
