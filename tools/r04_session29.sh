#!/bin/bash
# round-4 GPU session 29, one box: (i) radiance-texture layouts re-measured on the current kernel (TEXIR_TEX_LAYOUT = 0 row-major, 1 = 8x8 tiles, 2 = one line per footprint);
# (ii) whose fabric traffic is it?  TCC_EA0_RDREQ / TCC_MISS / TCP accesses of the shipped kernel against the no-shade probe build (c2 and c4, full launches)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s29
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() {  # label, layout, bench args
  v=$(TEXIR_TEX_LAYOUT=$2 timeout 400 python bench.py $3 --no-cpu --no-mat --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$1 $v" | tee -a $out/ab.txt
}
for cfg in "c4|--workload c4 --steps 3 --warmup 1" "c2|--workload c2 --steps 5 --warmup 1"; do
  label=${cfg%%|*}; args=${cfg#*|}
  for lay in 2 0 1; do run "$label layout$lay" $lay "$args"; done
done
cd /tmp && export TMPDIR=/tmp
pass() { lib=$1; wl=$2; name=$3; shift 3
  rm -rf /tmp/pmc_$name
  TEXIR_HIP_LIB=$lib timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu --no-mat --extra none > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'irt_group_kernel<false' in r['Kernel_Name']]
d=collections.defaultdict(float)
for r in rows: d[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(d))
PY
}
for wl in c2 c4; do
for v in shipped noshade; do
  lib=$R/texir_code_amd/libtexir_hip.so; [ $v = noshade ] && lib=$R/build_ab/libtexir_noshade.so
  echo "== $wl $v" | tee -a $out/pmc.txt
  pass $lib $wl ${v}_rdreq TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum | tee -a $out/pmc.txt
  pass $lib $wl ${v}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU | tee -a $out/pmc.txt
done
done
