#!/bin/bash
# round-3 GPU session 7: PMC passes of the IrT kernel (phase scheduling w2 + per-XCD chunk counters + wedge parts) on c4 / c4_scan / c2
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r03_s7
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
bash tools/profile_round.sh r03_s7/prof c4 c4_scan c2 > $out/profile_round.log 2>&1
tail -n 4 $out/profile_round.log | cut -c1-400
cp $R/profiles/pmc_c4.json $R/profiles/pmc_c2.json $R/profiles/pmc_c4_scan.json $out/ 2>/dev/null
