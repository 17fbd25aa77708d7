#!/bin/bash
# usage: tools/probes/ab.sh "<label>|ENV=.. ENV=.." ...   -- runs bench c2 + c4 (IrT only) per configuration and prints Mrays/s
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  for W in c2 c4; do
    v=$(env $envs timeout 300 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat --extra none --no-project 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
    echo "$label $W $v"
  done
done
