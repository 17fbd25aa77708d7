#!/bin/bash
# kernel sequence of ONE replayed material step (name, duration), from a rocprofv3 kernel trace of bench.py's material leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktm
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktm -- python $R/bench.py --no-cpu --steps 1 --warmup 0 --extra none --no-project --no-e2e > /tmp/ktm.log 2>&1
f=$(find /tmp/ktm -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the Adam launch (batched over both textures; or the one-channel texture's own) ends a step: take the kernels between the last two of them
idx=[i for i,r in enumerate(rows) if 'adam_tex' in r['Kernel_Name'] and ('kernel<1>' in r['Kernel_Name'] or 'batch_kernel' in r['Kernel_Name'])]
end=idx[-1]; start=idx[-2]+1
seg=rows[start:end+1]
t0=int(seg[0]['Start_Timestamp']); tot=0
for r in seg:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=d
    print('%8.1f us @%8.1f  %s'%(d,(int(r['Start_Timestamp'])-t0)/1e3,r['Kernel_Name'][:100]))
print('kernels',len(seg),'busy %.1f us'%tot,'span %.1f us'%((int(seg[-1]['End_Timestamp'])-t0)/1e3))
# period of the replayed steps (end of one step's last kernel to the end of the next one's) and the idle time between two replays
ends=[int(rows[i]['End_Timestamp']) for i in idx]
per=[(b-a)/1e3 for a,b in zip(ends[-21:-1],ends[-20:])]
gaps=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in idx[-21:-1]]
if per: print('step period over the last %d replays: mean %.1f us, min %.1f; idle between replays: mean %.1f us, min %.1f'%(len(per),sum(per)/len(per),min(per),sum(gaps)/len(gaps),min(gaps)))
PY
