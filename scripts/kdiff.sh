#!/bin/bash
# per-kernel averages of scripts/mask_probe.py at two upstream-gradient scales
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp
for sc in 1.0 1e-7; do
  (cd /tmp && rm -rf /tmp/kd_$sc && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kd_$sc -o x -- python $R/scripts/mask_probe.py $sc > /dev/null 2>&1)
done
python - <<'PY'
import csv, glob
def load(sc):
    f = glob.glob('/tmp/kd_%s/**/*kernel_stats.csv' % sc, recursive=True)[0]
    return {r['Name'][:60]: (int(r['Calls']), float(r['AverageNs']) / 1e3) for r in csv.DictReader(open(f))}
a, b = load('1.0'), load('1e-7')
tot_a = tot_b = 0
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1])[:30]:
    ca, ta = a[k]; cb, tb = b.get(k, (0, 0))
    tot_a += ca * ta; tot_b += cb * tb
    print('%-60s %6d %8.1f %8.1f  %+5.1f%%' % (k, ca, ta, tb, 100 * (tb - ta) / ta if ta else 0))
print('total (top 30) ms: %.1f vs %.1f' % (tot_a / 1e3, tot_b / 1e3))
PY
