#!/bin/bash
# parity + timing + rocprofv3 kernel stats of the fused resize + cross entropy (one GPU-box visit)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_sl
timeout 900 python -m pytest tests/test_segloss.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -4
timeout 300 python scripts/segloss_bench.py 2>/dev/null | tail -1 | tee gpurun_out/segloss_bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sl -o sl -- python $OLDPWD/scripts/segloss_bench.py > $OLDPWD/gpurun_out/sl_rocprof.log 2>&1)
find /tmp/prof_sl -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_sl/ \;
grep "k_upce" gpurun_out/prof_sl/sl_kernel_stats.csv | cut -c1-140
