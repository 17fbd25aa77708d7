#!/bin/bash
# round-2 GPU session 25: occupancy of the specular forward kernel (launch bounds 6 / 7 / 8 waves per SIMD with a matching LDS stack) -- material step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s25
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm default X=1
for w in 6 7 8; do abm spec$w TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec$w.so; done
abm default_again X=1
abm spec8_again TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec8.so
TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec8.so bash tools/trace_mat_step.sh > $out/mat_step_trace_spec8.txt 2>&1
grep -E "spec_kernel|kernels " $out/mat_step_trace_spec8.txt | cut -c1-100
