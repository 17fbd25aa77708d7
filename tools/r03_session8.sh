#!/bin/bash
# round-3 GPU session 8: PMC passes of c4 again (session 7's c4 passes also counted the c4_scan launch of the default line)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s8
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
bash tools/profile_round.sh r03_s8/prof c4 > $out/profile_round.log 2>&1
tail -n 3 $out/profile_round.log | cut -c1-300
cp $R/profiles/pmc_c4.json $out/ 2>/dev/null
