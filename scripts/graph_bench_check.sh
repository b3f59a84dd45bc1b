cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
echo "== default (graph)"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bg.log 2> gpurun_out/bg.err; echo rc=$?; tail -c 600 gpurun_out/bg.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bg.log').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['config'], j['roofline']['avg_launch_us'], j['roofline']['launches_timed'], j['roofline']['frac'], j['roofline']['note'][-120:])
PY
echo "== eager"; timeout 600 python bench.py --no-cpu-baseline --eager > gpurun_out/be.log 2> gpurun_out/be.err; echo rc=$?; python - <<'PY'
import json
j=json.loads(open('gpurun_out/be.log').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['config']['hip_graph'], j['roofline']['avg_launch_us'], j['roofline']['launches_timed'])
PY
echo "== 2 ranks on one device (gloo hooks), two-graph path"
CFFM_BENCH_BACKEND=gloo CFFM_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --spinup-steps 50 > gpurun_out/b2.log 2> gpurun_out/b2.err; echo rc=$?; tail -c 800 gpurun_out/b2.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/b2.log').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['config'], j['roofline']['avg_launch_us'], j['roofline']['launches_timed'])
PY
echo "== RCCL itself, single rank (CFFM_BENCH_FORCE_DIST): two-graph default, torch DDP, eager"
for v in "" "--ddp" "--eager"; do
  CFFM_BENCH_FORCE_DIST=1 timeout 400 python bench.py --no-cpu-baseline $v > gpurun_out/br.log 2> gpurun_out/br.err; echo "rc=$? stdout lines=$(wc -l < gpurun_out/br.log)"
  python - <<'PY'
import json
j=json.loads(open('gpurun_out/br.log').read())
print(j['value'], j['ms_per_step'], j['config']['hip_graph'], j['config']['hip_graph_calibration'], j['config']['grad_allreduce'])
PY
done
