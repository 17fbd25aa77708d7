#!/bin/bash
# session 14: the batched Adam launch alone -- grid shapes and non-temporal hints; idle time between graph replays
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s14
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
{
for rep in 1 2; do
for lib in libtexir_hip.so libtexir_nt1.so libtexir_nt2.so libtexir_nt3.so; do
  TEXIR_HIP_LIB=$R/texir_code_amd/$lib timeout 120 python tools/adam_batch_probe.py 2>&1 | tail -1
done
for gy in 1024 512 256 128; do
  TEXIR_ADAM_GRID_Y=$gy timeout 120 python tools/adam_batch_probe.py 2>&1 | tail -1
done
done
TEXIR_HIP_LIB=$R/texir_code_amd/libtexir_nt3.so TEXIR_ADAM_GRID_Y=1024 timeout 120 python tools/adam_batch_probe.py 2>&1 | tail -1
TEXIR_HIP_LIB=$R/texir_code_amd/libtexir_nt3.so TEXIR_ADAM_GRID_Y=512 timeout 120 python tools/adam_batch_probe.py 2>&1 | tail -1
} > $out/adam_batch_probe.txt 2>&1
cat $out/adam_batch_probe.txt
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 2 $out/mat_step_trace.txt | cut -c1-200
