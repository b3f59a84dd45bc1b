#!/bin/bash
# parity + timing + rocprofv3 kernel stats of the fused SegFormer embedding (one GPU-box visit)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_sf
timeout 900 python -m pytest tests/test_segfuse.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
timeout 300 python scripts/segfuse_bench.py 2>/dev/null | tail -1 | tee gpurun_out/segfuse_bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sf -o sf -- python $OLDPWD/scripts/segfuse_bench.py > $OLDPWD/gpurun_out/sf_rocprof.log 2>&1)
find /tmp/prof_sf -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_sf/ \;
grep "k_segfuse" gpurun_out/prof_sf/sf_kernel_stats.csv | cut -c1-120
