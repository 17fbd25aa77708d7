#!/bin/bash
# round-4 GPU session 8: PMC passes of the IrT kernel on c4 / c2 / c4_scan / c1 with the round's final kernel sources (-> profiles/pmc_*.json), kernel-trace
# stats of the default bench, bench lines of c2 and c1
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s8
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
bash tools/profile_round.sh r04_s8/prof c4 c2 c4_scan c1 > $out/profile_round.log 2>&1
tail -n 2 $out/profile_round.log | cut -c1-400
cp $R/profiles/pmc_c4.json $R/profiles/pmc_c2.json $R/profiles/pmc_c4_scan.json $R/profiles/pmc_c1.json $out/ 2>/dev/null
timeout 600 python bench.py --workload c2 --no-mat > $out/bench_c2.json 2>> $out/bench.err
timeout 600 python bench.py --workload c1 --steps 5 --warmup 1 --no-mat > $out/bench_c1.json 2>> $out/bench.err
tail -c 600 $out/bench_c1.json
