#!/bin/bash
# usage: tools/probes/ab_mat.sh "<label>|ENV=.." ...   -- material-step latency (bench c4 material leg) and NIrF-size IrT batch per configuration
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  v=$(env $envs timeout 300 python bench.py --no-cpu --steps 1 --warmup 0 --extra none --no-project --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); m=d['material_step']; print(m['ms'], 'back_to_back', m.get('ms_back_to_back'), 'host', m.get('host_ms_per_step'))" 2>&1 | tail -1)
  echo "$label material_step_ms $v"
done
