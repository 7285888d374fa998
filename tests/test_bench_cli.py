"""bench.py's N > 1 control flow on a box without GPUs (RLSTED_BENCH_STUB=1 swaps the device plan for a
stand-in and RCCL for gloo): `--gpus N` launches N ranks by itself, the ranks agree on n_gpus, the gather
delivers N x B frames, and exactly one JSON line comes out."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    env = dict(os.environ, RLSTED_BENCH_STUB='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_line(stdout):
    lines = [ln for ln in stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks_by_itself():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4'],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _one_line(r.stdout)
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak'
    assert d['final_gather']['frames_on_root'] == 2 * 4
    assert d['value'] > 0 and 'roofline' in d
    # N > 1: the line also carries BASELINE config 4 through the sharder (cost-weighted partition, one gather on rank 0)
    sw = d['fig2_sweep']
    assert sw['tasks'] == 1152 and sw['frames_on_root'] == 1152 and sum(sw['tasks_per_rank']) == 1152
    # by plan group: the 36 (PSF set, shape) groups are dealt whole, so no plan is set up on two ranks
    assert sum(sw['groups_per_rank']) == 36 and max(sw['cost_per_rank_rel']) < 1.1 and sw['gather_ms'] >= 0
    assert 'broadcast' in sw['psf_sets'] or 'every rank' in sw['psf_sets']


def test_under_the_drivers_launcher_command():
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '2']
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _one_line(r.stdout)
    assert d['n_gpus'] == 2 and d['final_gather']['frames_on_root'] == 4


def test_rank_count_mismatch_is_an_error():
    env = dict(_env(), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b'--gpus 2' in r.stderr
