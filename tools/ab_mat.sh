#!/bin/bash
# usage: tools/ab_mat.sh "<label>|ENV=.." ...   -- material-step latency (bench c4 material leg) and NIrF-size IrT batch per configuration
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  v=$(env $envs timeout 300 python bench.py --no-cpu --steps 1 --warmup 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "$label material_step_ms $v"
done
