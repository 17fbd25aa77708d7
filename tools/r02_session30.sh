#!/bin/bash
# round-2 GPU session 30: two-level deferred fold (the optimiser step takes level 2 -> 1 -> 0 over) -- full suite, material step A/B, trace
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s30
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 2400 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1
tail -n 12 $out/pytest_gpu.txt | cut -c1-220
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm defer2 X=1
abm defer1 TEXIR_DEFER_LEVELS=1
abm defer2_again X=1
abm defer1_again TEXIR_DEFER_LEVELS=1
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 22 $out/mat_step_trace.txt | cut -c1-110
