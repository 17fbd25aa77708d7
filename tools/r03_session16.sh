#!/bin/bash
# round-3 GPU session 16: XCD-contiguous pixel bands in the single-pass specular forward (TEXIR_SPEC_XCD=0: round-robin blocks)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s16
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
for rep in 1 2 3; do
for cfg in "xcd|" "roundrobin|TEXIR_SPEC_XCD=0" "general|TEXIR_SPEC_SINGLE=0"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $label $v" | tee -a $out/mat_ab.txt
done
done
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
grep -E "spec_|kernels " $out/mat_step_trace.txt | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scan_and_configs.py -m gpu -q -k "spec or 4k" 2>&1 | tail -2
