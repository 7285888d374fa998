"""N > 1 path on CPU: cost-weighted partition + gather, world_size 2, gloo."""
import os
import socket

import numpy as np
import pytest

from rescan_line_sted_amd import sharding


def test_partition_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    # figure-2 like mix: views per task 1..10
    costs = [sharding.task_cost(128 * 128, v, 20) for v in rng.choice([1, 2, 3, 4, 6, 8, 10], 144)]
    for world in (1, 2, 4, 8):
        shards = sharding.partition(costs, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) <= 1.1 * (sum(costs) / world)
        assert shards == sharding.partition(costs, world)          # deterministic
    assert sharding.partition([], 4) == [[], [], [], []]
    assert sharding.partition([1.0], 3) == [[0], [], []]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tasks, out_path):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tasks = list(range(n_tasks))
    costs = [1.0 + (t % 5) for t in tasks]

    def run_local(mine):        # stand-in for the device plan: result encodes the task id
        return np.stack([np.full((4, 6), float(t)) + np.arange(6) for t in mine])
    res = sharding.run_sharded(tasks, costs, run_local, dist)
    if rank == 0:
        np.save(out_path, res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_tasks', [7, 1])
def test_run_sharded_world_size_2_gloo(tmp_path, n_tasks):
    import torch.multiprocessing as mp
    out = str(tmp_path / 'res.npy')
    mp.spawn(_worker, args=(2, _free_port(), n_tasks, out), nprocs=2, join=True)
    res = np.load(out)
    assert res.shape == (n_tasks, 4, 6)
    for t in range(n_tasks):
        assert np.array_equal(res[t], np.full((4, 6), float(t)) + np.arange(6))


def _sweep_worker(rank, world, port, out_path):
    import torch.distributed as dist
    from rescan_line_sted_amd import sweep
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    objects = {'cat': np.zeros((1, 10, 12)), 'rings': np.zeros((1, 8, 8)), 'lines': np.zeros((1, 8, 8))}
    psf_sets = {'point': [None], 'line3': [None] * 3}

    def fake_run_tasks(tasks, objects_, psf_sets_, iterations, *a, **k):   # stand-in for the device plans
        ids = sweep.object_ids(objects_)
        return [np.full(objects_[o].shape[-2:], 100.0 * ids[o] + 10.0 * len(psf_sets_[p]) + s) for o, p, s in tasks]
    sweep.run_tasks = fake_run_tasks
    tasks, est = sweep.figure_2_sweep(objects, psf_sets, seeds=(0, 1, 2), iterations=5, dist=dist)
    if rank == 0:
        ids = sweep.object_ids(objects)
        assert len(est) == len(tasks) == 18
        for (o, p, s), e in zip(tasks, est):
            assert e.shape == objects[o].shape[-2:]
            assert np.all(e == 100.0 * ids[o] + 10.0 * len(psf_sets[p]) + s)
        open(out_path, 'w').write('ok')
    else:
        assert est is None
    dist.barrier()
    dist.destroy_process_group()


def test_mixed_shape_sweep_world_size_2_gloo(tmp_path):
    """figure_2_sweep with objects of two shapes (the figure's 128x128 and 160x160 in miniature):
    cost-weighted shards, one gather of zero-padded frames, cropped back in task order on rank 0."""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'ok.txt')
    mp.spawn(_sweep_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == 'ok'
