#!/bin/bash
# L2 hit rate of the fused MLP kernels with and without their global stores (PNL_ABLATE=4 build): does the store stream evict the weights?
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for a in 0 4; do
  for pmc in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    (cd /tmp && rm -rf /tmp/pm && rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/pm -o x -- $R/build/r03_mlp_abl$a > /dev/null 2>&1)
    f=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
    python - "$f" "$a" "$pmc" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'][:40]
    if 'mlp' in k and int(r.get('Grid_Size', 0) or 0) >= 225 * 512:
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print('ABLATE=%s %-40s' % (sys.argv[2], k), ' '.join('%s=%.3g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in d.items()))
PY
  done
done
