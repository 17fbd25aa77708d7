#!/bin/bash
# round-2 GPU session 23: counters of the specular forward kernel inside the material step (lane utilisation, instructions per wave, waiting)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s23
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
run() { name=$1; shift
  rm -rf /tmp/pmc_$name
  TEXIR_MAT_GRAPH=0 timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --steps 1 --warmup 0 --no-cpu > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$name" <<'PY' | tee -a $R/gpurun_out/r02_s23/pmc.txt
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); ns=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:60]
    if not any(t in k for t in ('spec_kernel','adam_tex','tex_gather','mip_pyr')): continue
    d[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
    ns[(k,r['Dispatch_Id'])]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k in d:
    tot=sum(v for (kk,_),v in ns.items() if kk==k)
    print(sys.argv[2], k, "launches", len(n[k]), "avg_us %.1f" % (tot/len(n[k])/1e3), {c: "%.4g" % (v/len(n[k])) for c,v in d[k].items()})
PY
  tail -n 1 /tmp/pmc_$name.log | cut -c1-200 >> $out/log.txt
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
run sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCC_HIT_sum TCC_MISS_sum
