#!/bin/bash
# usage: tools/probes/ab_layout.sh [workloads...]  -- IrT rate per hit-shader texture layout (2 = float32 3x3 tiles, 3 = 4-byte texels 5x5, 4 = 4-byte texels 8x4) on the RGBE-born texture
R=${GRAFT_REPO_ROOT:-/root/repo}
export TEXIR_SYNTH_CACHE=${TEXIR_SYNTH_CACHE:-/tmp/texir_synth}
WLS=("$@"); [ ${#WLS[@]} -eq 0 ] && WLS=(c4 c2 c4_scan house)
for W in "${WLS[@]}"; do
  for rep in 1 2; do
    for L in 2 3 4; do
      v=$(TEXIR_TEX_LAYOUT=$L timeout 600 python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu --no-mat --extra none --no-project 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['workload'][-60:])" 2>&1 | tail -1)
      echo "$W layout=$L rep=$rep $v"
    done
  done
done
