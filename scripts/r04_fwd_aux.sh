#!/bin/bash
# Round-4: k_cfm_attn_fwd with different cache policies on its K / V row gathers (does keeping the head's bias fragments in the CU's L1 help?)
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for l in $R/build/libcffm_fwdaux*.so; do
  echo "$(basename $l): $(bash scripts/kstats.sh $l 'attn_fwd' | tail -1)"
done 2>&1 | tee gpurun_out/r04_fwd_aux.txt
