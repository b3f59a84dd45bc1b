#!/bin/bash
# SQ issue/stall counters of the kernels of the rows next to the hot path (segfuse, upce): one rocprofv3 --pmc pass per counter
# group (kernel-trace only), over scripts/segloss_bench.py and scripts/segfuse_bench.py.
cd "$(dirname "$0")/.." && R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for s in segloss segfuse; do
    (cd /tmp && rm -rf /tmp/sqr_${s}_$i && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/sqr_${s}_$i -o p -- python $R/scripts/${s}_bench.py > $R/gpurun_out/pmc/sqr_${s}_$i.log 2>&1)
  done
done
python - <<'PY' | tee gpurun_out/pmc/sq_rows_summary.txt
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/tmp/sqr_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        if not re.search('k_upce|k_segfuse', k): continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in sorted(acc.items()):
    print(k)
    print('   ' + '  '.join('%s=%.4g' % (c, v[1] / v[0]) for c, v in sorted(d.items())))
    g = lambda n: d[n][1] / d[n][0] if n in d else float('nan')
    wc = g('SQ_WAVE_CYCLES')
    print('   shares of wave cycles: issuing %.0f %%, waiting (s_waitcnt / barrier) %.0f %%, VALU-active %.0f %%, LDS-active %.0f %%; LDS bank conflicts %.0f %% of LDS cycles'
          % (100 * g('SQ_ACTIVE_INST_ANY') / wc, 100 * g('SQ_WAIT_ANY') / wc, 100 * g('SQ_ACTIVE_INST_VALU') / wc, 100 * g('SQ_ACTIVE_INST_LDS') / wc,
             100 * g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1)))
PY
