#!/bin/bash
# CFFM++ prototype layer (BASELINE config 5): GPU parity tests of the fused block + its timing (bench.py gtc_step) + rocprofv3 kernel stats.
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_boundary.py -x -q -m gpu -k "gtc or cffmpp or headpp or config5" 2>&1 | tail -5
python - <<'PY' > gpurun_out/r06_gtc_step.json
import json, sys, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.gtc_step(torch.device('cuda:0'), 2), indent=1))
PY
cat gpurun_out/r06_gtc_step.json
cat > /tmp/gtc_prof.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer_cluster(dim=256, depth=1, num_heads=8, window_size=7).to(dev)
for k in (8, 100):
    x = torch.randn(2, 3600, 256, device=dev, requires_grad=True); c = torch.randn(2, k, 256, device=dev, requires_grad=True); gy = torch.randn(2, 3600, 256, device=dev)
    for _ in range(20):
        for p in m.parameters(): p.grad = None
        m(x, 60, 60, c)[0].backward(gy)
torch.cuda.synchronize()
PY
(cd /tmp && rm -rf /tmp/gp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o x -- python /tmp/gtc_prof.py $R > /dev/null 2>&1)
cp $(find /tmp/gp -name '*kernel_stats.csv' | head -1) gpurun_out/r06_gtc_kernel_stats.csv; head -25 gpurun_out/r06_gtc_kernel_stats.csv | cut -c1-150
