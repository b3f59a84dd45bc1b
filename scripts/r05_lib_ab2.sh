#!/bin/bash
# A/B of library builds: usage scripts/r05_lib_ab2.sh <tag> lib1.so lib2.so ...  (two alternations; step time + the stage times that the
# weight-gradient group touches)
cd "$(dirname "$0")/.." && R=$PWD; mkdir -p gpurun_out
T=$1; shift
for rep in 1 2; do
for l in "$@"; do
  python scripts/bench_with_lib.py $l --steps 30 --warmup 5 --spinup-steps 100 --no-cpu-baseline --no-head-step --no-gtc-step 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('$l', 'step', j['ms_per_step'], ' '.join('%s %.1f' % (n, k[n]['avg_us']) for n in ('gemm_dw_group','mlp_bwd_fused','mlp_fwd_fused','gemm_qkv_dx','cfm_attn_bwd') if n in k))" | tee -a gpurun_out/r05_ab_$T.txt
done; done
