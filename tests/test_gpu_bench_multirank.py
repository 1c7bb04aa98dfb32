"""bench.py's N > 1 path, launched the way the driver launches it (torch.distributed.run, one rank per process), on
the one-GPU test box: two ranks share cuda:0 over gloo (NERF_DIST_BACKEND; RCCL refuses two ranks on one device) and
a small hash table keeps it quick.  Checks the contract fields of the JSON line and that the level-sharded table
is what ran."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(**extra_env):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NERF_DIST_BACKEND='gloo', **extra_env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--table-log2', '17', '--no-cpu-baseline', '--no-render', '--profile-steps', '1']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 prints ONE line
    return json.loads(lines[0]), out.stderr


def test_bench_two_ranks_under_torchrun():
    d, _ = _run_bench()
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['unit'] == 'rays/s'
    assert d['value'] > 0 and abs(d['value'] - 2 * 4096 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    assert d['config']['global_rays'] == 8192 and 'sharded by level' in d['config']['parallelism'], d['config']
    assert 'failed' not in d['config']['parallelism']
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and d['higher_is_better'] is True
    assert d['roofline'] is not None and d['cpu_baseline'] is None


def test_bench_falls_back_when_the_sharded_worker_stalls():
    """bench.py's supervisor (N > 1): a worker generation whose first level-sharded step never returns is ended and a
    second generation runs the replicated table, and the line says so."""
    d, err = _run_bench(NERF_TEST_BENCH_STALL='1', NERF_BENCH_STALL_S='15')
    assert 'worker silent' in err
    assert d['n_gpus'] == 2 and d['value'] > 0
    assert 'sharded by level' not in d['config']['parallelism'] and 'replicated table' in d['config']['parallelism'], d['config']
