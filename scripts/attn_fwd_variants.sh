#!/bin/bash
# k_cfm_attn_fwd (one-shot) vs k_cfm_attn_fwd_s (split-key, online softmax): parity tests + the kernel's event interval in bench.py
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for v in oneshot split; do
  echo "== CFFM_ATTN_FWD=$v"
  CFFM_ATTN_FWD=$v timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -1
  CFFM_ATTN_FWD=$v timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print(j['value'], j['ms_per_step'], 'attn fwd event interval us', r['avg_launch_us'], 'stage', j['kernels']['cfm_attn_fwd'])
"
done
