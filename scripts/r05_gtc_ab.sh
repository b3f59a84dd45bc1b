#!/bin/bash
# A/B of library builds on the CFFM++ prototype block: usage scripts/r05_gtc_ab.sh <tag> lib1.so lib2.so ...   (two alternations; bench.py gtc_step)
cd "$(dirname "$0")/.." && R=$PWD; mkdir -p gpurun_out
T=$1; shift
for rep in 1 2; do
for l in "$@"; do
  CFFM_LIB=$l python - <<PY | tee -a gpurun_out/r05_gtc_ab_$T.txt
import os, sys, torch
sys.path.insert(0, '.')
from vss_cffm_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ['CFFM_LIB'])
import bench
j = bench.gtc_step(torch.device('cuda:0'), 2)
print(os.environ['CFFM_LIB'], ' '.join('%s %.4f ms (attn %.1f / %.1f, dw %.1f, mlp %.1f / %.1f, gemm %.1f us)' % (k, j[k]['ms_per_step'], j[k]['stage_us_per_step']['gtc_attn_fwd'], j[k]['stage_us_per_step']['gtc_attn_bwd'], j[k]['stage_us_per_step']['gemm_dw_group'], j[k]['stage_us_per_step']['mlp_fwd_fused'], j[k]['stage_us_per_step']['mlp_bwd_fused'], j[k]['stage_us_per_step']['linear_gemm']) for k in ('K=8', 'K=100')))
PY
done; done
