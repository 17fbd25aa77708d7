#!/bin/bash
# round-2 GPU session 13: wave-uniform node steps through the scalar cache (TEXIR_UNIFORM_SLOAD) A/B + parity of the variant
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s13
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
ab() { label=$1; shift
  for W in "${WLS[@]}"; do
    v=$(env "$@" timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])" 2>&1 | tail -1)
    echo "$label $W $v" | tee -a $out/ab.txt
  done
}
WLS=(c4 c2 c4_scan)
ab default X=1
for v in "$@"; do ab $v TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_$v.so; done
for v in "$@"; do
TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_$v.so timeout 1200 python -m pytest tests/test_gpu_watertight.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_$v.txt 2>&1
tail -n 3 $out/pytest_$v.txt | cut -c1-200
done
