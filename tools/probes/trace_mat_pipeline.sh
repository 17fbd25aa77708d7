#!/bin/bash
# kernel timeline (start, end, name) of the last two replayed material steps, from a rocprofv3 kernel trace of bench.py's material leg (any step form)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktp -- python $R/bench.py --no-cpu --steps 1 --warmup 0 --extra none > /tmp/ktp.log 2>&1
f=$(find /tmp/ktp -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_tex' in r['Kernel_Name'] and 'kernel<1>' in r['Kernel_Name']]
start=idx[-3]+1
seg=rows[start:]
t0=int(seg[0]['Start_Timestamp'])
for r in seg:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print('%9.1f -> %9.1f  (%6.1f us)  q%s  %s'%(s,e,e-s,r.get('Queue_Id','?'),r['Kernel_Name'][:90]))
PY
