#!/bin/bash
# A/B of the CFFA backward launch shape (LNB_WAVES x LNB_SPLIT builds): step time + standalone stage times
cd "$(dirname "$0")/.." && R=$PWD; mkdir -p gpurun_out
for rep in 1 2; do
for l in vss_cffm_amd/libcffm_hip.so build/libcffm_w8.so build/libcffm_w8s2.so build/libcffm_w4s7.so; do
  python scripts/bench_with_lib.py $l --steps 30 --warmup 5 --spinup-steps 100 --no-cpu-baseline --no-head-step --no-gtc-step 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('$l', 'step', j['ms_per_step'], 'tgt', k['ln_pool_bwd']['avg_us'], 'ref', k['ln_pool_bwd_ref']['avg_us'])" | tee -a gpurun_out/r05_lnb_ab.txt
done; done
