#!/bin/bash
# k_cfm_attn_fwd with parts removed (profiling builds, results are wrong by construction): which resource bounds it?
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp
for a in ${ABL:-0 1 2 3 4 7 8 15}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value vss_cffm_amd/csrc/cffm_hip.hip -o /tmp/abl_$a.so -ldl -DCFFM_EXPERIMENTS -DFWD_ABLATE=$a 2>/dev/null
  echo "FWD_ABLATE=$a: $(bash scripts/kstats.sh /tmp/abl_$a.so 'attn_fwd' | tail -1)"
done
