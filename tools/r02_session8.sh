#!/bin/bash
# round-2 GPU session 8: L1 (TCP) rate calibration, full gpu suite, material-step A/B, PMC profiles of the final kernels (c4, c2), bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s8
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
( timeout 120 tools/tcp_rate ) > $out/tcp_rate.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_tcprate && timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d /tmp/pmc_tcprate -- $R/tools/tcp_rate > /tmp/pmc_tcprate.log 2>&1
  f=$(find /tmp/pmc_tcprate -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" > $out/tcp_rate_pmc.txt <<'PY'
import csv,sys,collections
d=collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k=(r['Dispatch_Id'], r['Kernel_Name'][:40], r['Grid_Size'])
    d.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
    d[k]['ns']=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in d.items(): print(k, v)
PY
)
cat $out/tcp_rate.txt
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 8 $out/pytest_gpu.txt | cut -c1-200
abm() { label=$1; shift
  v=$(env "$@" timeout 600 python bench.py --no-cpu --steps 1 --warmup 0 2>>$out/abm.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'], d['value'])" 2>&1 | tail -1)
  echo "$label material_step_ms,irt $v" | tee -a $out/abm.txt
}
abm default X=1
abm grid2048 TEXIR_SPEC_GRID_CAP=2048
abm grid1024 TEXIR_SPEC_GRID_CAP=1024
bash tools/trace_mat_step.sh > $out/mat_step_trace.txt 2>&1
tail -n 30 $out/mat_step_trace.txt | cut -c1-110
bash tools/profile_round.sh r02_s8/prof c4 c2 > $out/profile_round.log 2>&1
tail -n 3 $out/profile_round.log | cut -c1-600
cp $R/profiles/pmc_c4.json $R/profiles/pmc_c2.json $out/ 2>/dev/null
timeout 900 python bench.py --extra c4_scan > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json
timeout 600 python bench.py --workload c2 > $out/bench_c2.json 2>> $out/bench_default.err
tail -n 1 $out/bench_c2.json
timeout 600 python bench.py --workload c1 --steps 5 --warmup 1 > $out/bench_c1.json 2>> $out/bench_default.err
tail -n 1 $out/bench_c1.json
