#!/bin/bash
# round-2 GPU session 6: 128-byte full-float node A/B, fixed tester tests
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s6
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
ab() { label=$1; shift
  for W in "${WLS[@]}"; do
    v=$(env "$@" timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['config']['scene']['node_bytes'])" 2>&1 | tail -1)
    echo "$label $W $v" | tee -a $out/ab.txt
  done
}
WLS=(c4 c2 c4_scan)
ab default X=1
ab f32 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_f32.so
TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_f32.so timeout 1200 python -m pytest tests/test_gpu_watertight.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_f32.txt 2>&1
tail -n 5 $out/pytest_f32.txt | cut -c1-200
timeout 2400 python -m pytest tests/test_gpu_tester.py tests/test_gpu_trainer.py -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 12 $out/pytest_gpu.txt | cut -c1-220
