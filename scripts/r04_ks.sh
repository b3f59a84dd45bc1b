#!/bin/bash
# Round-4: the key-split attention backward: stage tests, stage timing under rocprofv3 (experiments build: group count, ablation builds)
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-a}
run() { # label, env...
  (cd /tmp && rm -rf /tmp/bp && env "${@:2}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o x -- timeout 300 python $R/scripts/r04_attn_bwd_bench.py $LIB 2>&1 | grep "attn_bwd stage")
  python - "$1" <<'PY'
import csv, glob, sys
fs = glob.glob("/tmp/bp/**/*kernel_stats.csv", recursive=True)
if not fs: sys.exit(0)
for r in csv.DictReader(open(fs[0])):
    if any(k in r['Name'] for k in ('attn_bwd', 'sum_splits', 'dkv_gather')):
        print('   %-14s %-36s %5s calls %8.1f us avg' % (sys.argv[1], r['Name'][:36], r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
{
timeout 900 python -m pytest tests/test_attn_bwd.py -x -q -m gpu 2>&1 | tail -3
LIB=$R/build/libcffm_exp.so
run g27 CFFM_BWD_GROUPS=27
for ng in 32; do run g$ng CFFM_BWD_GROUPS=$ng; done
for l in $R/build/libcffm_ks_abl*.so; do [ -f $l ] && LIB=$l && run $(basename $l) ; done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r04_ks_${T}.txt
