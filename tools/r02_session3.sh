#!/bin/bash
# round-2 GPU session 3: gpu suite on the new material path, material-step A/B (spec kernel occupancy x lanes per pixel), trace
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s3
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 2400 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1
tail -n 15 $out/pytest_gpu.txt
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm default X=1
abm default_lpp8 TEXIR_SPEC_LPP=8
abm default_lpp4 TEXIR_SPEC_LPP=4
abm spec6 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec6.so
abm spec6_lpp8 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec6.so TEXIR_SPEC_LPP=8
abm spec6_lpp4 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec6.so TEXIR_SPEC_LPP=4
abm spec7 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec7.so
abm spec7_lpp4 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_spec7.so TEXIR_SPEC_LPP=4
abm mt TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_mt.so
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 36 $out/mat_step_trace.txt | cut -c1-120
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -iE "VALU|SQ_INST_CYCLES|SQ_BUSY|GRBM" | head -60 ) > $out/counters_list.txt 2>&1
