#!/bin/bash
# round-2 GPU session 32: is the number of resident waves of the scratch-using kernels capped by the runtime's scratch limit?  (material step / IrT A/B)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s32
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm default X=1
abm limit1g HSA_SCRATCH_SINGLE_LIMIT=1073741824
abm limit4g HSA_SCRATCH_SINGLE_LIMIT=4294967296
abm limit1g_async HSA_SCRATCH_SINGLE_LIMIT=1073741824 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4294967296
abm noasync HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
abm limit64m HSA_SCRATCH_SINGLE_LIMIT=67108864 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=67108864
abm default_again X=1
HSA_SCRATCH_SINGLE_LIMIT=1073741824 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4294967296 bash tools/trace_mat_step.sh > $out/mat_step_trace_limit.txt 2>&1
grep -E "spec_kernel|kernels " $out/mat_step_trace_limit.txt | cut -c1-100
