#!/bin/bash
# the N > 1 paths of bench.py on a one-GPU box (round 4: the same checks on the final build, plus the collective field): self-launched 2 ranks sharing cuda:0 over gloo (three graphs + host-issued
# all-reduces; the head step with SyncBatchNorm over the two ranks), and RCCL itself with a single rank: the DEFAULT one-graph form with
# the all-reduces captured inside, and the three-graph form (CFFM_BENCH_GRAPH_COLLECTIVES=0)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    hs = j.get('head_step') or {}
    print(j['value'], j['ms_per_step'], 'n_gpus', j['n_gpus'], j['rccl'], 'graph', j['config']['hip_graph'], 'sync', j['config']['ranks_in_sync'], '|', j['config']['hip_graph_note'][:90], '| head_step', hs.get('ms_per_step'), hs.get('world'), hs.get('grads_in_sync'), hs.get('norm'), hs.get('error'), '| collective', j.get('collective'))
except Exception as e:
    print('no JSON line:', e)
PY
}
echo "== python bench.py --gpus 2 (self-launch; gloo, both ranks on cuda:0), head step with SyncBatchNorm over the ranks"
CFFM_BENCH_BACKEND=gloo CFFM_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --spinup-steps 20 --no-stage-timing > gpurun_out/r04_d2.log 2> gpurun_out/r04_d2.err; echo rc=$?; tail -c 400 gpurun_out/r04_d2.err; show gpurun_out/r04_d2.log
for v in "1" "0"; do
  echo "== RCCL, single rank, CFFM_BENCH_GRAPH_COLLECTIVES=$v"
  CFFM_BENCH_GRAPH_COLLECTIVES=$v CFFM_BENCH_FORCE_DIST=1 timeout 400 python bench.py --no-cpu-baseline --no-head-step --graph > gpurun_out/r04_dr$v.log 2> gpurun_out/r04_dr$v.err; echo "rc=$?"; tail -c 300 gpurun_out/r04_dr$v.err; show gpurun_out/r04_dr$v.log
done
