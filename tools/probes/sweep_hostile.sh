export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() { wl=$1; shift; v=$(env "$@" timeout 900 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --no-mat --extra none --no-project 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])"); echo "$wl $* -> $v"; }
for wl in c4_scan house; do
run $wl A=1
run $wl TEXIR_IRT_TEXELS_PER_WAVE=1
run $wl TEXIR_SCHED_WEIGHT=1
run $wl TEXIR_SCHED_WEIGHT=2
run $wl TEXIR_SCHED_WEIGHT=3
run $wl TEXIR_MAX_LEAF=2
run $wl TEXIR_BVH_LAYOUT=3
done
