#!/bin/bash
# SQ issue/stall counters of the hot kernels (one rocprofv3 --pmc pass per counter group, kernel-trace only).
# usage: scripts/pmc_sq.sh [kernel-regex]
cd "$(dirname "$0")/.." && R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/sq_$i && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/sq_$i -o p -- python $R/scripts/stage_times.py --steps 2 > $R/gpurun_out/pmc/sq_$i.log 2>&1)
done
python - "${1:-attn|gemm|ln_pool}" <<'PY' | tee gpurun_out/pmc/sq_summary.txt
import csv, glob, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/tmp/sq_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        if not re.search(sys.argv[1], k): continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k)
    print('   ' + '  '.join('%s=%.4g' % (c, v[1] / v[0]) for c, v in sorted(d.items())))
PY
