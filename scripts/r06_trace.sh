#!/bin/bash
# rocprofv3 kernel trace (per-dispatch start / end timestamps) of a short bench run: per-kernel stats + a timeline of ONE step
# (which kernels overlap, where the gaps are).  usage: scripts/r06_trace.sh <tag> [extra bench args]
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-x}; shift
(cd /tmp && rm -rf /tmp/bp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-head-step --no-cfg4-step --no-gtc-step --no-stage-timing --spinup-steps 50 "$@" > $R/gpurun_out/r06_trace_$TAG.json 2>/dev/null)
cp $(find /tmp/bp -name '*kernel_stats.csv' | head -1) gpurun_out/r06_kernel_stats_$TAG.csv
python - $TAG <<'PY'
import csv, sys, glob, json
tag = sys.argv[1]
f = glob.glob('/tmp/bp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 6 steps: a step starts at each k_param_prep
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_param_prep') or r['Kernel_Name'].startswith('k_transpose_prep')]
try:
    j = json.loads(open('gpurun_out/r06_trace_%s.json' % tag).read().strip().splitlines()[-1])
    print('bench under rocprof: %.4f ms/step' % j['ms_per_step'])
except Exception as e:
    print('bench line unreadable', e)
if len(idx) > 8:
    a, b = idx[-6], idx[-5]
    t0 = int(rows[a]['Start_Timestamp'])
    step = rows[a:b]
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
    span = max(int(r['End_Timestamp']) for r in step) - t0
    # union of busy intervals
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
    u, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            u += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    u += cur_e - cur_s
    print('one step: %d kernels, span %.1f us, sum of kernel times %.1f us, union busy %.1f us (gaps %.1f us)' % (len(step), span / 1e3, busy / 1e3, u / 1e3, (span - u) / 1e3))
    with open('gpurun_out/r06_timeline_%s.txt' % tag, 'w') as o:
        for r in step:
            o.write('%9.1f %8.1f  q%-3s %s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
st = list(csv.DictReader(open('gpurun_out/r06_kernel_stats_%s.csv' % tag)))
tot = sum(float(r['TotalDurationNs']) for r in st)
nstep = max(1, len(idx))
print('kernel time per step (all streams): %.1f us over %d steps' % (tot / 1e3 / nstep, nstep))
for r in st[:26]:
    print('%-60s %5s calls %8.1f us avg %5.1f %% %7.1f us/step' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot, float(r['TotalDurationNs']) / 1e3 / nstep))
PY
python - <<'PY'
# per-launch durations of the kernels that run once per block, in launch order (block 0 / block 1 alternate in the forward)
import csv, glob
f = glob.glob('/tmp/bp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for key in ('k_mlp_fwd', 'k_mlp_bwd', 'k_cfm_attn_fwd', 'k_cfm_attn_bwd', 'k_ln_pool_fwd', 'k_ln_pool_bwd'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if key in r['Kernel_Name']][-24:]
    print('%-16s' % key, ' '.join('%.0f' % x for x in d))
PY
