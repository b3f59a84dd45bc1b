#!/bin/bash
# A/B of library builds for the attention forward: usage scripts/r05_fwd_ab.sh <tag> lib1.so lib2.so ...
# (two alternations; step time, k_cfm_attn_fwd inside the graph and back to back)
cd "$(dirname "$0")/.." && R=$PWD; mkdir -p gpurun_out
T=$1; shift
for rep in 1 2; do
for l in "$@"; do
  python scripts/bench_with_lib.py $l --steps 30 --warmup 5 --spinup-steps 100 --no-cpu-baseline --no-head-step --no-gtc-step 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']; r=j['roofline']
print('$l', 'step', j['ms_per_step'], 'attn_fwd in graph %.2f us, back to back %s us, frac %s / b2b %s' % (k['cfm_attn_fwd']['avg_us'], r.get('back_to_back_us'), r.get('frac'), r.get('frac_back_to_back')))" | tee -a gpurun_out/r05_fwd_ab_$T.txt
done; done
