"""bench.py's output contract on a real GPU: exactly one JSON line on stdout with the driver's fields, the `roofline` and
`cpu_baseline` objects, a real optimizer step inside the timed region (the replayed graph advances the device-side step count)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2', '--spinup-steps', '10'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 4 and j['warmup'] == 2 and j['higher_is_better'] is True and j['scaling'] == 'weak'
    assert j['unit'] == 'clips/s' and j['data'] == 'synthetic' and j['vs_baseline'] is None
    assert 'workload' in j['config'] and 'model' not in j['config']
    assert abs(j['value'] - j['config']['clips_per_gpu'] * 1e3 / j['ms_per_step']) < 1e-2 * j['value']
    ro = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in ro, k
    assert ro['bound'] == 'hbm' and ro['unit'] == 'GB/s' and 0.0 < ro['frac'] < 1.0
    assert abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-3
    cb = j['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] in ('port', 'reference') and cb['value'] > 0
