#!/bin/bash
# Round-6 quick GPU visit: (selected) GPU tests, the default bench line, one-step timeline + rocprofv3 kernel stats of the bench.
# usage: scripts/r06_quick.sh <tag> [pytest selection ...]      (no selection: layer goldens + attention stage tests)
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-quick}; shift
SEL=${@:-tests/test_attn_bwd.py tests/test_gpu_parity.py}
timeout 1500 python -m pytest $SEL -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06_${T}_tests.txt; tail -12 gpurun_out/r06_${T}_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-head-step --no-cfg4-step > gpurun_out/r06_${T}_bench.json 2> gpurun_out/r06_${T}_bench.err; tail -2 gpurun_out/r06_${T}_bench.err
python - "$T" <<'PY'
import json, sys
T = sys.argv[1]
try:
    j = json.loads(open('gpurun_out/r06_%s_bench.json' % T).read().strip().splitlines()[-1])
    print('bench:', j['value'], j['ms_per_step'], 'roofline', j['roofline']['frac'], j['roofline']['avg_launch_us'], j['roofline'].get('back_to_back_us'))
    rk = j.get('roofline_kernels', {})
    for k, v in rk.items():
        if isinstance(v, dict) and 'us' in v: print('  %-20s %7.1f us  %s' % (k, v['us'], {a: b for a, b in v.items() if a.endswith('frac')}))
except Exception as e:
    print('bench line unreadable:', e)
PY
timeout 900 bash scripts/r06_trace.sh $T --graph > gpurun_out/r06_${T}_trace.txt 2>&1; head -40 gpurun_out/r06_${T}_trace.txt; echo; cat gpurun_out/r06_timeline_${T}.txt
