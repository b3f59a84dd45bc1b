#!/bin/bash
# the N > 1 paths of bench.py on a one-GPU box, round 5: the gradient exchange as ONE all-reduce behind the whole backward (default) against the
# per-block overlapped form (CFFM_BENCH_EXCHANGE=blockwise), each as one graph with the collectives captured inside and as separate graphs +
# host-issued collectives (CFFM_BENCH_GRAPH_COLLECTIVES=0), with a single RCCL rank; then 2 self-launched ranks sharing cuda:0 over gloo
# (ranks in sync, head step with SyncBatchNorm over the ranks)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    hs = j.get('head_step') or {}
    c = j.get('collective') or {}
    print('%.1f clips/s  %.4f ms' % (j['value'], j['ms_per_step']), 'n_gpus', j['n_gpus'], j['rccl'], 'graph', j['config']['hip_graph'], 'sync', j['config']['ranks_in_sync'], '|', j['config']['hip_graph_note'][:110], '| head_step', hs.get('ms_per_step'), hs.get('world'), hs.get('grads_in_sync'), '| calls', c.get('allreduce_calls_per_step'), 'bytes', c.get('allreduce_bytes_per_step'), 'exposed wait ms', c.get('exposed_wait_ms_per_step'))
except Exception as e:
    print('no JSON line:', e)
PY
}
for ex in whole blockwise; do for v in 1 0; do
  echo "== RCCL, single rank, CFFM_BENCH_EXCHANGE=$ex CFFM_BENCH_GRAPH_COLLECTIVES=$v"
  CFFM_BENCH_EXCHANGE=$ex CFFM_BENCH_GRAPH_COLLECTIVES=$v CFFM_BENCH_FORCE_DIST=1 timeout 400 python bench.py --no-cpu-baseline --no-head-step --no-gtc-step --graph > gpurun_out/r05_dr_${ex}_$v.log 2> gpurun_out/r05_dr_${ex}_$v.err; echo "rc=$?"; grep -v "amdgpu.ids\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r05_dr_${ex}_$v.err | tail -c 300; show gpurun_out/r05_dr_${ex}_$v.log
done; done
echo "== python bench.py --gpus 2 (self-launch; gloo, both ranks on cuda:0), head step with SyncBatchNorm over the ranks"
CFFM_BENCH_BACKEND=gloo CFFM_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --spinup-steps 20 --no-stage-timing --no-gtc-step > gpurun_out/r05_d2.log 2> gpurun_out/r05_d2.err; echo rc=$?; tail -c 400 gpurun_out/r05_d2.err; show gpurun_out/r05_d2.log
echo "== the same with CFFM_BENCH_EXCHANGE=blockwise"
CFFM_BENCH_EXCHANGE=blockwise CFFM_BENCH_BACKEND=gloo CFFM_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --spinup-steps 20 --no-stage-timing --no-gtc-step --no-head-step > gpurun_out/r05_d2b.log 2> gpurun_out/r05_d2b.err; echo rc=$?; tail -c 300 gpurun_out/r05_d2b.err; show gpurun_out/r05_d2b.log
