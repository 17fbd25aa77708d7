#!/bin/bash
# round-2 GPU session 28: PMC profiles of the hostile scene (c4_scan) and of the reference-sized case (c1), bench lines carrying them
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s28
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
sed -i 's/^timeout 900 rocprofv3 --kernel-trace/true || timeout 900 rocprofv3 --kernel-trace/' tools/profile_round.sh      # (PMC passes only here)
bash tools/profile_round.sh r02_s28/prof c4_scan c1 > $out/profile_round.log 2>&1
tail -n 3 $out/profile_round.log | cut -c1-300
cp $R/profiles/pmc_c4_scan.json $R/profiles/pmc_c1.json $out/ 2>/dev/null
timeout 900 python bench.py --extra c4_scan --no-mat > $out/bench_default_extra.json 2> $out/bench.err
tail -n 1 $out/bench_default_extra.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['extra_workloads'])[:1500])"
timeout 600 python bench.py --workload c1 --steps 5 --warmup 1 > $out/bench_c1.json 2>> $out/bench.err
tail -n 1 $out/bench_c1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['roofline'])[:1200])"
