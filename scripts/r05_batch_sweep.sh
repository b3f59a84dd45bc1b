#!/bin/bash
# bench.py at B = 1 / 2 / 4 / 8 clips per GPU on one box: ms per step, clips/s, the attention forward back to back and its fraction of 8 TB/s
# (the forward launch is ceil(B * 81 * 8 / 1024) rounds of one workgroup latency: DESIGN 3g)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for b in 1 2 3 4 6 8; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-head-step --no-gtc-step 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('B=$b  %.4f ms/step  %.1f clips/s  attn_fwd back to back %s us  frac %s  workgroups %d = %.2f rounds of 1024' % (j['ms_per_step'], j['value'], r.get('back_to_back_us'), r.get('frac_back_to_back'), $b*81*8, $b*81*8/1024.0))" | tee -a gpurun_out/r05_batch_sweep.txt
done
