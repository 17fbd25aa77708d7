#!/bin/bash
# round-3 GPU session 5: material-step kernel trace with the optimiser inside the graph, Adam entry points in isolation, IrT: hardware sin/cos A/B,
# one-texel-per-wave form on the scan scene
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s5
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 30 $out/mat_step_trace.txt | cut -c1-150
python tools/adam_probe.py 2>&1 | tee $out/adam_probe.txt
for cfg in "c4 default|" "c4 slowsin|TEXIR_HIP_LIB=$R/build_ab/slowsin.so" "c2 default|" "c2 slowsin|TEXIR_HIP_LIB=$R/build_ab/slowsin.so" "c4_scan default|" "c4_scan slowsin|TEXIR_HIP_LIB=$R/build_ab/slowsin.so" "c4_scan onetexel|TEXIR_IRT_TEXELS_PER_WAVE=1"; do
  label=${cfg%%|*}; envs=${cfg#*|}; W=${label%% *}
  v=$(env $envs timeout 400 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
  echo "$label $v" | tee -a $out/ab.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "irt or generate_dir" 2>&1 | tail -3
