#!/bin/bash
# One parameterised GPU session (replaces the per-session r0N_session*.sh scripts; what each earlier session ran is listed in tools/SESSIONS.md).
#   usage (on the GPU box, from the repo root):  tools/session.sh <tag> <step> [<step> ...]
#   steps:  tests | tests-x | bench | driver | bench-lean | gloo2 | ab:<label>:<lib.so>[:ENV=..,ENV=..] | chain:<workload> | profile:<wl>[,<wl>...] | matpmc | e2e:<workload>
# Everything lands under gpurun_out/<tag>/ ; summaries that are to be judged are copied to profiles/ by hand afterwards.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
export TEXIR_SYNTH_CACHE=${TEXIR_SYNTH_CACHE:-/tmp/texir_synth}
cd $R
irt_line() {  # workload, extra env... -> "value ms_per_step kernel_ms"
  wl=$1; shift
  env "$@" timeout 600 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --no-mat --extra none --no-project 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" 2>&1 | tail -1
}
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case $step in
    tests)    timeout 3000 python -m pytest tests -m gpu -q > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -4 $out/pytest.txt ;;
    tests-x)  timeout 3000 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -4 $out/pytest.txt ;;
    bench)    TEXIR_BENCH_FULL=$out/bench_full.json timeout 1800 python bench.py --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; tail -c 600 $out/bench.err ;;
    driver)   TEXIR_BENCH_FULL=$out/bench_driver_full.json timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "driver-shaped bench rc $?"; tail -c 3200 $out/bench_driver.json ;;
    bench-lean) TEXIR_BENCH_FULL=$out/bench_lean_full.json timeout 900 python bench.py --steps 2 --warmup 1 --no-e2e --extra none > $out/bench_lean.json 2> $out/bench_lean.err; echo "bench rc $?" ;;
    gloo2)    TEXIR_BENCH_FULL=$out/gloo2_full.json TEXIR_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 \
                --workload tiny --steps 1 --warmup 1 --no-cpu --mat --mat-steps 4 --mat-res 512 --mat-cube 32 > $out/gloo2.json 2> $out/gloo2.err; echo "rc $?"
              python - $out/gloo2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1].replace(".json", "_full.json")))
for k in ("material_step","material_step_view_mode"):
    print(k, {a:b for a,b in d.get(k,{}).items() if a not in ("config","roofline")})
PY
              ;;
    ab:*)     IFS=: read -r _ label lib envs <<< "$step"
              envs=${envs//,/ }
              for W in c4 c2 c4_scan house; do
                echo "$label $W shipped $(irt_line $W $envs)" | tee -a $out/ab_$label.txt
                echo "$label $W $lib $(irt_line $W TEXIR_HIP_LIB=$R/build_ab/$lib $envs)" | tee -a $out/ab_$label.txt
              done ;;
    chain:*)  wl=${step#chain:}; timeout 2400 python tools/chain_probe.py --workload $wl --out $out/chain_$wl.json > $out/chain_$wl.log 2>&1; echo "rc $?"; tail -25 $out/chain_$wl.log ;;
    profile:*) wls=${step#profile:}; bash tools/profile_round.sh $tag ${wls//,/ } > $out/profile.log 2>&1; tail -3 $out/profile.log ;;
    matpmc)   bash tools/mat_step_pmc.sh $tag > $out/matpmc.log 2>&1; tail -3 $out/matpmc.log ;;
    mattrace) bash tools/trace_mat_step.sh $tag > $out/mattrace.log 2>&1; tail -20 $out/mattrace.log ;;
    weights:*) wl=${step#weights:}; for w in 1 2 3; do echo "$wl TEXIR_SCHED_WEIGHT=$w $(irt_line $wl TEXIR_SCHED_WEIGHT=$w)" | tee -a $out/ab_weights.txt; done; echo "$wl tuned $(irt_line $wl)" | tee -a $out/ab_weights.txt ;;
    e2e:*)    wl=${step#e2e:}; timeout 1500 python tools/stage_time.py --workload $wl > $out/e2e_$wl.json 2> $out/e2e_$wl.err; echo "rc $?"; tail -c 1500 $out/e2e_$wl.json ;;
    *) echo "unknown step $step" ;;
  esac
done
