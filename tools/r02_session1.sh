#!/bin/bash
# round-2 GPU session 1: instruction issue costs, watertightness (both intersectors), A/B of pop culling x texture layout x intersector,
# how much of c4 is radiance-texture traffic.  Results -> gpurun_out/r02_s1/
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s1
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
( timeout 300 tools/issue_rate ) > $out/issue_rate.txt 2>&1
# watertightness: default library (Pluecker edge functions) and the Moeller-Trumbore build
timeout 900 python -m pytest tests/test_gpu_watertight.py -q -x -m gpu > $out/watertight_default.txt 2>&1
TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_mt.so timeout 900 python -m pytest tests/test_gpu_watertight.py -q -m gpu > $out/watertight_mt.txt 2>&1
tail -3 $out/watertight_default.txt $out/watertight_mt.txt
ab() { label=$1; shift
  for W in "${WLS[@]}"; do
    v=$(env "$@" timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])" 2>&1 | tail -1)
    echo "$label $W $v" | tee -a $out/ab.txt
  done
}
WLS=(c4)
ab default_l2 TEXIR_TEX_LAYOUT=2
ab default_l0 TEXIR_TEX_LAYOUT=0
ab default_l1 TEXIR_TEX_LAYOUT=1
ab mt_l2 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_mt.so TEXIR_TEX_LAYOUT=2
ab mt_l0 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_mt.so TEXIR_TEX_LAYOUT=0
ab nocull_l2 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_nocull.so TEXIR_TEX_LAYOUT=2
ab r1_l0 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_r1.so TEXIR_TEX_LAYOUT=0
ab r1_l2 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_r1.so TEXIR_TEX_LAYOUT=2
WLS=(c4_tex1k)
ab r1_l0 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_r1.so TEXIR_TEX_LAYOUT=0
ab default_l2 TEXIR_TEX_LAYOUT=2
WLS=(c2 c4_scan)
ab default_l2 TEXIR_TEX_LAYOUT=2
ab r1_l0 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_r1.so TEXIR_TEX_LAYOUT=0
# the whole gpu suite on the default library
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
tail -5 $out/pytest_gpu.txt
cat $out/issue_rate.txt
