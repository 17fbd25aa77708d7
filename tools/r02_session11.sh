#!/bin/bash
# round-2 GPU session 11: node-fetch cost in the vector L1 for candidate node formats (tools/tcp_node.hip), timing + tag-lookup counters
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s11
mkdir -p $out
cd $R
( timeout 300 tools/tcp_node ) > $out/tcp_node.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_tcpnode && timeout 400 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d /tmp/pmc_tcpnode -- $R/tools/tcp_node > /tmp/pmc_tcpnode.log 2>&1
  f=$(find /tmp/pmc_tcpnode -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" > $out/tcp_node_pmc.txt <<'PY'
import csv,sys,collections
d=collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k=(int(r['Dispatch_Id']), r['Kernel_Name'][:40])
    d.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
    d[k]['ns']=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
# every second dispatch is the timed one (2000 iterations); fetches per launch = 2000*4 per wave x 8192 waves
for k,v in d.items():
    if v.get("TCP_TOTAL_ACCESSES_sum", 0) > 1e9:
        n = 2000*4*8192.0
        print(k[1], "accesses/fetch %.1f" % (v.get('TCP_TOTAL_CACHE_ACCESSES_sum',0)/n), "total/fetch %.1f" % (v.get('TCP_TOTAL_ACCESSES_sum',0)/n), "clk/fetch/CU %.1f" % (v.get('TCP_GATE_EN1_sum',0)/256/(2000*4*32.0)))
PY
)
paste -d' ' <(cat $out/tcp_node.txt) <(cut -d' ' -f3- $out/tcp_node_pmc.txt | cut -c30-) | cut -c1-200
