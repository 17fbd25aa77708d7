#!/bin/bash
# round-2 GPU session 20: albedo / roughness chains of the material step on two streams (TEXIR_MAT_STREAMS) -- tests, A/B, trace
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s20
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 2400 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_tester.py tests/test_gpu_scale.py -m gpu -q -x > $out/pytest_gpu.txt 2>&1
tail -n 12 $out/pytest_gpu.txt | cut -c1-220
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm streams X=1
abm one_stream TEXIR_MAT_STREAMS=0
abm streams_again X=1
abm one_stream_again TEXIR_MAT_STREAMS=0
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 26 $out/mat_step_trace.txt | cut -c1-110
tail -n 5 $out/abm.err | cut -c1-300
