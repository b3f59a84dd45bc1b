#!/bin/bash
# HBM traffic of the attention kernels from PMC counters: separate rocprofv3 passes for FETCH_SIZE and WRITE_SIZE
# (TCC slots: they do not fit one pass), kernel-trace only -- as MI355X_MICROARCH.md prescribes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rm -rf /tmp/pmc_$c && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $ROOT/scripts/stage_times.py --steps 3 > $ROOT/gpurun_out/pmc/$c.log 2>&1)
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY' > gpurun_out/pmc/$c.summary.csv
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r.get('Counter_Name') != sys.argv[2]: continue
    k = r['Kernel_Name'].split('(')[0]
    acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
print('kernel,dispatches,avg_%s' % sys.argv[2])
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%s,%d,%.3f' % (k, n, s / n))
PY
  head -12 gpurun_out/pmc/$c.summary.csv
done
