#!/bin/bash
# session 20: taps per round of the gather (2 / 4 / 8): kernel time inside the step
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s20
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
for v in hip d1 d2; do
  TEXIR_HIP_LIB=$R/texir_code_amd/libtexir_$v.so bash tools/trace_mat_step.sh > $out/trace_$v.txt 2>&1
  echo "$v: $(grep gather $out/trace_$v.txt | cut -c1-60) | $(grep '^kernels' $out/trace_$v.txt)"
done > $out/ab_gather_diag.txt 2>&1
cat $out/ab_gather_diag.txt
