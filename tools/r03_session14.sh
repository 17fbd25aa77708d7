#!/bin/bash
# round-3 GPU session 14: passes per part down to 8 (more chunks for small sample counts): c1 and reduced-spp runs of c2, A/B against 64 (round 2); parity of the IrT forms
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s14
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scan_and_configs.py -m gpu -q -k "irt or c1 or scan_scene" 2>&1 | tail -3
for cfg in "c1 64|--workload c1 --steps 10 --warmup 2" "c2 64|--workload c2 --spp 64" "c2 256|--workload c2 --spp 256" "c2 1024|--workload c2 --spp 1024" "c4 128|--workload c4 --spp 128"; do
  label=${cfg%%|*}; args=${cfg#*|}
  for mc in 8 16 64; do
    v=$(TEXIR_IRT_MIN_PART_CELLS=$mc timeout 400 python bench.py $args --no-cpu --no-mat 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$label min_cells=$mc $v" | tee -a $out/ab.txt
  done
done
