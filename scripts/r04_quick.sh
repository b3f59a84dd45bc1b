#!/bin/bash
# Round-4 quick GPU visit: attention-backward stage tests, layer goldens, the default bench line, rocprofv3 kernel stats of the bench.
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-quick}
python -m pytest tests/test_attn_bwd.py tests/test_gpu_parity.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r04_${T}_tests.txt; tail -12 gpurun_out/r04_${T}_tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-head-step > gpurun_out/r04_${T}_bench.json 2> gpurun_out/r04_${T}_bench.err; tail -2 gpurun_out/r04_${T}_bench.err
(cd /tmp && rm -rf /tmp/bp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-head-step --spinup-steps 100 > $R/gpurun_out/r04_${T}_bench_prof.json 2>/dev/null)
cp $(find /tmp/bp -name '*kernel_stats.csv' | head -1) gpurun_out/r04_${T}_kernel_stats.csv
python - "$T" <<'PY'
import csv, json, sys
T = sys.argv[1]
j = json.loads(open('gpurun_out/r04_%s_bench.json' % T).read().strip().splitlines()[-1])
print('bench:', j['value'], j['ms_per_step'], 'roofline', j['roofline']['frac'], j['roofline']['avg_launch_us'])
rk = j.get('roofline_kernels', {})
for k in ('attn_bwd_fused', 'attn_bwd_prep'):
    if k in rk: print(k, rk[k])
rows = list(csv.DictReader(open('gpurun_out/r04_%s_kernel_stats.csv' % T)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print('%-64s %6s calls %8.1f us avg %5.1f %%' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
