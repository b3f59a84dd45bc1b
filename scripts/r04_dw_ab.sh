#!/bin/bash
# Round-4 A/B on one box: the weight-gradient groups (one late group vs two split groups) beside the key-split attention backward
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do
for v in default one split; do
  if [ $v == default ]; then unset CFFM_DW_GROUP; else export CFFM_DW_GROUP=$v; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-head-step 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['ms_per_step'], j['value'], j['config'].get('hip_graph'), j['config'].get('hip_graph_calibration'))"
done; done 2>&1 | tee gpurun_out/r04_dw_ab.txt
