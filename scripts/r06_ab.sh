#!/bin/bash
# A/B of tuning switches with the -DCFFM_EXPERIMENTS build (build/libcffm_exp.so): usage scripts/r06_ab.sh <tag> "ENV=.. ENV=.." ["ENV.." ...]
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
T=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    line=$(env $cfg python scripts/bench_with_lib.py build/libcffm_exp.so --steps 30 --warmup 5 --no-cpu-baseline --no-head-step --no-cfg4-step --no-gtc-step --no-stage-timing --graph 2>/dev/null | tail -1)
    echo "$cfg :: $(echo $line | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"])')" | tee -a gpurun_out/r06_ab_$T.txt
  done
done
