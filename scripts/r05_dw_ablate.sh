#!/bin/bash
# k_gemm_group_tt under GEMM_ABLATE builds (timing only; results are garbage): standalone stage time of the weight-gradient group
cd "$(dirname "$0")/.." && R=$PWD; mkdir -p gpurun_out
for a in exp abl8 abl9 abl10 abl12 abl24 abl11; do
  python scripts/bench_with_lib.py build/libcffm_$a.so --steps 10 --warmup 3 --spinup-steps 20 --no-cpu-baseline --no-head-step --no-gtc-step 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('$a', 'step', j['ms_per_step'], 'dw_group', k['gemm_dw_group']['avg_us'], 'mlp_fwd', k['mlp_fwd_fused']['avg_us'], 'mlp_bwd', k['mlp_bwd_fused']['avg_us'], 'qkv_dx', k['gemm_qkv_dx']['avg_us'])" | tee -a gpurun_out/r05_dw_ablate.txt
done
