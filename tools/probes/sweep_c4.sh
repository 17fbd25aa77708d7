export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() { v=$(env "$@" timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu --no-mat --extra none --no-project 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "$* -> $v"; }
run A=1
run TEXIR_BVH_LAYOUT=1
run TEXIR_BVH_LAYOUT=2
run TEXIR_BVH_LAYOUT=3
run TEXIR_IRT_LOG2PARTS=4
run TEXIR_IRT_LOG2PARTS=6
run TEXIR_IRT_MIN_PART_CELLS=4
run TEXIR_IRT_MIN_PART_CELLS=16
run TEXIR_MAX_LEAF=2
run TEXIR_MAX_LEAF=3
run TEXIR_SCHED_WEIGHT=2
run A=2
