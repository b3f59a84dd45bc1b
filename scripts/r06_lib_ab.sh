#!/bin/bash
# A/B of library builds: usage scripts/r06_lib_ab.sh <tag> lib1.so lib2.so ...   (two alternations; step time + the fused Mlp stage times)
cd "$(dirname "$0")/.." && R=$PWD; mkdir -p gpurun_out
T=$1; shift
for rep in 1 2; do
for l in "$@"; do
  python scripts/bench_with_lib.py $l --steps 30 --warmup 5 --spinup-steps 100 --no-cpu-baseline --no-head-step --no-cfg4-step --no-gtc-step 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('$l', 'step', j['ms_per_step'], ' '.join('%s %.1f' % (n, k[n]['avg_us']) for n in ('mlp_fwd_fused','mlp_bwd_fused','gemm_dw_group','gemm_qkv_fwd','gemm_qkv_dx','cfm_attn_fwd') if n in k))" | tee -a gpurun_out/r06_ab_$T.txt
done; done
