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
    for k in ('roofline_kernels', 'head_step', 'rccl', 'tolerance'):
        assert k in j, k
    assert j['rccl'] == {'world': 1, 'backend': None}
    rk = j['roofline_kernels']
    for fam in ('gemm_qkv_fwd', 'mlp_fwd_fused', 'mlp_bwd_fused', 'gemm_dw_group', 'attn_bwd_fused', 'ln_pool_fwd', 'ln_pool_bwd', 'ln_pool_bwd_ref'):
        assert fam in rk and 0.0 < rk[fam]['frac'] < 1.0, (fam, rk.get(fam))
    assert j['head_step'] and j['head_step'].get('ms_per_step', 0) > 0, j['head_step']


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` starts its own N ranks; with fewer than N devices it must fail loudly, not report N = 1."""
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1'],
                       cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert '%d GPUs requested' % n in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    """The launcher path itself (torch.distributed.run, 2 ranks) on a one-GPU box: both ranks share cuda:0 over gloo (test
    hooks); the line must say n_gpus 2, name the backend and report the ranks' parameters bit-identical after the run."""
    env = dict(os.environ, PYTHONPATH=ROOT, CFFM_BENCH_ONE_DEVICE='1', CFFM_BENCH_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--spinup-steps', '5',
                        '--no-stage-timing'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['rccl'] == {'world': 2, 'backend': 'gloo'} and j['config']['ranks_in_sync'] is True
    assert j['config']['global_batch'] == 2 * j['config']['clips_per_gpu']


@pytest.mark.gpu
@pytest.mark.parametrize('exchange', ['whole', 'blockwise'])
def test_bench_single_rank_over_rccl_takes_the_one_graph_form(exchange):
    """The default N > 1 path -- ONE graph per step with the RCCL all-reduce captured inside -- cannot be run with real peers on a
    one-GPU box; with CFFM_BENCH_FORCE_DIST the same code runs as a one-rank RCCL group: the capture must be the form taken, the
    parameters stay finite and in sync, and the line carries what the exchange moves and how long the compute stream waits for it.
    Both forms of the exchange: ONE all-reduce of the flat gradient buffer behind the whole backward (default since round 5) and the
    per-block overlapped form (CFFM_BENCH_EXCHANGE=blockwise)."""
    env = dict(os.environ, PYTHONPATH=ROOT, CFFM_BENCH_FORCE_DIST='1', CFFM_BENCH_EXCHANGE=exchange)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2', '--spinup-steps', '10',
                        '--no-cpu-baseline', '--no-head-step', '--no-gtc-step', '--no-cfg4-step'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['rccl'] == {'world': 1, 'backend': 'nccl'}
    assert j['config']['ranks_in_sync'] is True and j['config']['params_finite'] is True
    assert j['config']['hip_graph'] is True and j['config']['hip_graph_note'].startswith('ONE graph per step with the RCCL all-reduce')
    c = j['collective']
    assert c['allreduce_calls_per_step'] == (1 if exchange == 'whole' else 2) and c['allreduce_bytes_per_step'] > 6_000_000
    assert c['capture_form'] == j['config']['hip_graph_note'] and ('ONE all-reduce' in c['capture_form']) == (exchange == 'whole')
    assert 0.0 <= c['exposed_wait_ms_per_step'] < j['ms_per_step']
