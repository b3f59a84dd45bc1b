#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of a few fwd+bwd steps; run on the GPU box.
# usage: scripts/kstats.sh [lib.so] [filter-regex]
cd "$(dirname "$0")/.." && R=$PWD
export TMPDIR=/tmp
LIB=${1:-$R/vss_cffm_amd/libcffm_hip.so}
D=/tmp/kstats_$$
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o x -- python $R/scripts/stage_times.py --lib $LIB --steps 10 > /dev/null 2>&1)
python - "$D" "${2:-.}" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r['Name']):
        print('%-60s calls %5s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
