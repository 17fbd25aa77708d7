#!/bin/bash
# session 15: Adam with non-temporal loads + stores at 4 / 5 / 6 waves per SIMD (probe and in the step); step period of back-to-back replays
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s15
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
L=$R/texir_code_amd
{
for rep in 1 2; do
for lib in libtexir_hip.so libtexir_nt3.so libtexir_nt3w5.so libtexir_nt3w6.so; do
  TEXIR_HIP_LIB=$L/$lib timeout 120 python tools/adam_batch_probe.py 2>&1 | tail -1
done
done
} > $out/adam_batch_probe.txt 2>&1
cat $out/adam_batch_probe.txt
bash tools/ab_mat.sh "default|X=1" "nt3|TEXIR_HIP_LIB=$L/libtexir_nt3.so" "nt3w5|TEXIR_HIP_LIB=$L/libtexir_nt3w5.so" "nt3w6|TEXIR_HIP_LIB=$L/libtexir_nt3w6.so" "default|X=1" "nt3|TEXIR_HIP_LIB=$L/libtexir_nt3.so" "nt3w5|TEXIR_HIP_LIB=$L/libtexir_nt3w5.so" "nt3w6|TEXIR_HIP_LIB=$L/libtexir_nt3w6.so" > $out/ab_adam_nt.txt 2>&1
cat $out/ab_adam_nt.txt
