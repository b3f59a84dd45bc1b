#!/bin/bash
# Per-SIMD utilisation of every hot kernel: SQ_ACTIVE_INST_* are quad-cycles summed over waves
# (MI355X_MICROARCH.md), so busy fraction of a pipe = counter x 4 / (1024 SIMDs x kernel duration x clock).  Durations: rocprofv3
# kernel trace of the same driver (scripts/stage_times.py, eager launches, B = 2).  Output: gpurun_out/r06_pmc_sq.txt
cd "$(dirname "$0")/.." && R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc
bash scripts/pmc_sq.sh "attn|dkv_gather|ln_pool|gemm|mlp|panel|dw_dma" > /dev/null 2>&1
(cd /tmp && rm -rf /tmp/sq_dur && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sq_dur -o p -- python $R/scripts/stage_times.py --steps 5 > /dev/null 2>&1)
python - <<'PY' | tee gpurun_out/r06_pmc_sq.txt
import csv, glob, re
dur = {}
for r in csv.DictReader(open(glob.glob('/tmp/sq_dur/**/*kernel_stats.csv', recursive=True)[0])):
    dur[r['Name'].split('(')[0][:48]] = float(r['AverageNs']) * 1e-9
cur, vals = None, {}
for line in open('gpurun_out/pmc/sq_summary.txt'):
    if not line.startswith('   '):
        cur = line.strip(); vals[cur] = {}
    else:
        for kv in line.split():
            k, v = kv.split('='); vals[cur][k] = float(v)
print('per-SIMD busy fraction of each pipe = SQ_ACTIVE_INST_x (quad-cycles summed over waves) x 4 / (1024 SIMDs x duration x clock), at 2.1 / 2.4 GHz;')
print('MFMA pipe = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x duration x clock) [already cycles per SIMD-set: reported as counter / (4 x 256 CUs)];')
print('instructions per wave from SQ_INSTS_*; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; LDS conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE')
print('%-44s %7s | %-13s %-13s %-13s %-13s %-13s | %6s %6s %6s %6s | %5s %5s' % ('kernel', 'us', 'ANY', 'VALU', 'SALU', 'LDS', 'VMEM', 'valu/w', 'salu/w', 'lds/w', 'mfma/w', 'wait', 'confl'))
for k, d in vals.items():
    t = dur.get(k)
    if not t: continue
    f = lambda c: '%.2f-%.2f' % (d.get(c, 0) * 4 / (1024 * t * 2.4e9), d.get(c, 0) * 4 / (1024 * t * 2.1e9))
    w = d.get('SQ_WAVES', 1)
    print('%-44s %7.1f | %-13s %-13s %-13s %-13s %-13s | %6.0f %6.0f %6.0f %6.0f | %5.2f %5.2f' % (
        k[:44], t * 1e6, f('SQ_ACTIVE_INST_ANY'), f('SQ_ACTIVE_INST_VALU'), f('SQ_ACTIVE_INST_SCA'), f('SQ_ACTIVE_INST_LDS'), f('SQ_ACTIVE_INST_VMEM'),
        d.get('SQ_INSTS_VALU', 0) / w, d.get('SQ_INSTS_SALU', 0) / w, d.get('SQ_INSTS_LDS', 0) / w, d.get('SQ_INSTS_MFMA', 0) / w,
        d.get('SQ_WAIT_ANY', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1), d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
    print('%-44s         | MFMA pipe busy %.2f-%.2f' % ('', d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * t * 2.4e9), d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * t * 2.1e9)))
PY
cp gpurun_out/pmc/sq_summary.txt gpurun_out/r06_pmc_sq_raw.txt
