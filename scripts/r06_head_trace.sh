#!/bin/bash
# rocprofv3 kernel trace of the whole-head training step replayed from one HIP graph: timeline of ONE replay (start, duration, queue).
# usage: scripts/r06_head_trace.sh <tag>
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-x}
cat > /tmp/head_trace.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
import bench
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
mark = torch.zeros(4096, device=dev)
orig_sync = torch.cuda.synchronize
out = bench.head_step(dev, 2, steps=6)
print(out)
PY
(cd /tmp && rm -rf /tmp/hp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -o x -- python /tmp/head_trace.py $R > $R/gpurun_out/r06_head_trace_$TAG.log 2>&1)
tail -2 gpurun_out/r06_head_trace_$TAG.log | cut -c1-400
python - $TAG <<'PY'
import csv, sys, glob
tag = sys.argv[1]
f = glob.glob('/tmp/hp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the graph replays are the last 6 + 3 repetitions of the same kernel sequence: take the last one (period = distance between the last
# two occurrences of the final kernel's name pattern start)
# replays are separated by a host synchronisation: cut the trace where the device sat idle for > 40 us and take the last piece
# with a whole step's worth of kernels
segs, cur = [], [rows[0]]
end = int(rows[0]['End_Timestamp'])
for r in rows[1:]:
    if int(r['Start_Timestamp']) - end > 40000:
        segs.append(cur); cur = []
    cur.append(r)
    end = max(end, int(r['End_Timestamp']))
segs.append(cur)
big = [s_ for s_ in segs if len(s_) > 60]
per = len(big[-1]) if big else None
print('segments', len(segs), 'last whole step:', per, 'kernels')
if per:
    step = big[-1]
    t0 = int(step[0]['Start_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
    span = max(int(r['End_Timestamp']) for r in step) - t0
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
    u, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            u += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    print('one replay: %d kernels, span %.1f us, sum of kernel times %.1f us, union busy %.1f us (gaps %.1f us)' % (len(step), span / 1e3, busy / 1e3, u / 1e3, (span - u) / 1e3))
    with open('gpurun_out/r06_head_timeline_%s.txt' % tag, 'w') as o:
        for r in step:
            o.write('%9.1f %8.1f  q%-3s %s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:90]))
PY
