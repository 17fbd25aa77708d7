#!/bin/bash
# round-2 GPU session 4: mixed-pipe issue costs, full gpu suite, material step A/B (Adam kernel forms, lanes per pixel)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s4
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
( timeout 200 tools/issue_rate ) > $out/issue_rate.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 25 $out/pytest_gpu.txt | cut -c1-200
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm default X=1
abm adam_scalar TEXIR_ADAM_SCALAR=1
abm lpp8 TEXIR_SPEC_LPP=8
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 32 $out/mat_step_trace.txt | cut -c1-110
cat $out/issue_rate.txt
