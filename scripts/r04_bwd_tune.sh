#!/bin/bash
# attention-backward stage alone under rocprofv3: kernel durations (product lib + any build/libcffm_x*.so variants)
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { # label, env...
  (cd /tmp && rm -rf /tmp/bp && env "${@:2}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o x -- python $R/scripts/r04_attn_bwd_bench.py $LIB 2>&1 | grep "attn_bwd stage")
  python - "$1" <<'PY'
import csv, glob, sys
fs = glob.glob("/tmp/bp/**/*kernel_stats.csv", recursive=True)
if not fs: sys.exit(0)
for r in csv.DictReader(open(fs[0])):
    if any(k in r['Name'] for k in ('attn_bwd', 'sum_splits')):
        print('   %-14s %-36s %5s calls %8.1f us avg' % (sys.argv[1], r['Name'][:36], r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
{
python -m pytest tests/test_attn_bwd.py -x -q -m gpu 2>&1 | tail -2
LIB=; run product X=1
for l in $R/build/libcffm_x*.so; do [ -f $l ] && LIB=$l && run $(basename $l) X=1; done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r04_bwd_tune_${1:-a}.txt
{ LIB=$R/build/libcffm_exp.so; [ -f $LIB ] && for nk in 32 64 236; do run nk$nk CFFM_BWD_NK=$nk; done; } 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a gpurun_out/r04_bwd_tune_${1:-a}.txt
