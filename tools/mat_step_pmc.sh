#!/bin/bash
# usage: tools/mat_step_pmc.sh <tag>   -- measured fabric traffic of ONE replayed material step (stage 2, 4k textures): two rocprofv3 --pmc passes over
# bench.py's material leg (read requests; written bytes), folded per kernel over the last 20 replayed steps -> gpurun_out/<tag>/pmc_mat_step.json and profiles/
tag=${1:-matpmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export TEXIR_SYNTH_CACHE=${TEXIR_SYNTH_CACHE:-/tmp/texir_synth}
for pass in "rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "wr WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/matpmc_$name
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/matpmc_$name -- python $R/bench.py --no-cpu --steps 1 --warmup 0 --extra none --no-project --no-e2e > /tmp/matpmc_$name.log 2>&1
  f=$(find /tmp/matpmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/mat_$name.csv
done
python $R/tools/mat_step_pmc.py $out $out/pmc_mat_step.json && cp $out/pmc_mat_step.json $R/profiles/pmc_mat_step.json
rm -f $out/mat_rd.csv $out/mat_wr.csv $out/mat_tcc.csv
