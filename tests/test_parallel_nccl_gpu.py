"""The node-range-partitioned runner on REAL RCCL (`nccl` backend), one fresh process per GPU, against the numpy oracle:
BASELINE configs 2 and 5 in their multi-rank form (BasicGCN on the user-item graph; HybridBertGCN with 768-d BERT rows on
the user-item-properties graph).  world = 2 needs two GPUs and is skipped on a one-GPU box; world = 1 runs the same worker —
process group, padded layout, all_gather_into_tensor — with a single rank.  bench.py's own launcher is exercised too."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _torchrun(world, script_args, timeout=600, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize('case', ['basic_ui', 'hybrid_uip'])
@pytest.mark.parametrize('world', [1, 2])
def test_partitioned_runner_on_rccl_matches_oracle(world, case):
    if torch.cuda.device_count() < world:
        pytest.skip("needs {} GPUs".format(world))
    r = _torchrun(world, [os.path.join(ROOT, 'tests', 'nccl_worker.py'), case])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'every pair scored once: True' in r.stdout


@pytest.mark.parametrize('case', ['basic_ui', 'hybrid_uip'])
@pytest.mark.parametrize('world', [2, 3])
def test_partitioned_runner_rehearsed_on_one_gpu(world, case):
    """The same worker with `world` PROCESSES sharing one GPU (gloo process group, the layer exchange as an all_reduce of the
    zero-padded table: parallel.SharedDeviceCollectives): everything of a multi-rank run but RCCL itself, on a one-GPU box."""
    r = _torchrun(world, [os.path.join(ROOT, 'tests', 'nccl_worker.py'), case], AMAR_REHEARSE_ONE_GPU='1')
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'every pair scored once: True' in r.stdout and 'rehearsal' in r.stdout


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` end to end on one GPU in rehearsal mode: the launcher, two ranks, one JSON line with n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', AMAR_REHEARSE_ONE_GPU='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--scale', '4', '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and 'roofline' in out and 'rehearsal' in out['config']['parallelism']


@pytest.mark.parametrize('gpus', [1, 2])
def test_bench_launches_its_own_ranks(gpus):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts N ranks itself and rank 0 prints one JSON line."""
    if torch.cuda.device_count() < gpus:
        pytest.skip("needs {} GPUs".format(gpus))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    if gpus == 1:
        env['AMAR_FORCE_DIST'] = '1'                                 # the partitioned runner + RCCL with a single rank
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--steps', '2', '--warmup', '1',
                        '--scale', '4', '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == gpus and out['value'] > 0 and 'roofline' in out
    if gpus > 1:
        assert 'RCCL all-gather' in out['config']['parallelism']


def test_bench_supervisor_runs_a_failed_rank_again_with_eager_steps():
    """bench.py with more than one rank runs each rank's work in a child; a child that dies (here: on purpose, AMAR_BENCH_FAIL_FIRST) is
    run once more with eager steps and the line says so — rehearsed with two ranks on one GPU."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', AMAR_REHEARSE_ONE_GPU='1', AMAR_BENCH_FAIL_FIRST='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'AMAR_STEP_GRAPH'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--scale', '4', '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and 'eager steps timed' in out['config']['retry']
    assert 'once more with eager steps' in r.stderr
