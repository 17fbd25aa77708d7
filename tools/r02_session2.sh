#!/bin/bash
# round-2 GPU session 2: instruction issue costs, watertightness (Woop + box slack vs Moeller-Trumbore), A/B, full gpu suite, bench, material-step trace
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s2
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
( timeout 120 tools/issue_rate ) > $out/issue_rate.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_watertight.py -q -m gpu > $out/watertight_default.txt 2>&1
TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_mt.so timeout 900 python -m pytest tests/test_gpu_watertight.py -q -m gpu > $out/watertight_mt.txt 2>&1
tail -n 3 $out/watertight_default.txt $out/watertight_mt.txt
ab() { label=$1; shift
  for W in "${WLS[@]}"; do
    v=$(env "$@" timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])" 2>&1 | tail -1)
    echo "$label $W $v" | tee -a $out/ab.txt
  done
}
WLS=(c4 c2 c4_scan)
ab default X=1
ab mt TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_mt.so
ab r1 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_r1.so TEXIR_TEX_LAYOUT=0 TEXIR_BOX_SLACK_LOG2=99
WLS=(c4)
ab default_noslack TEXIR_BOX_SLACK_LOG2=99
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 15 $out/pytest_gpu.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 40 $out/mat_step_trace.txt
cat $out/issue_rate.txt
