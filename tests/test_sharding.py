"""N > 1 path on CPU: cost-weighted partition + gather, world_size 2, gloo (tests/comm_gloo.py stands in
for the RCCL communicator of the C ABI), and the file rendezvous of the RCCL unique id."""
import os
import socket

import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # comm_gloo, also in the spawned workers
from rescan_line_sted_amd import sharding  # noqa: E402


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


def test_partition_by_plan_group():
    """sharding.partition_groups: whole (PSF set, shape) groups per rank -- each rank that touches a group pays the plan's
    set-up -- only groups of 128 tasks and more are cut, into pieces of at least 64; complete, deterministic."""
    rng = np.random.default_rng(1)
    views = rng.choice([1, 2, 3, 4, 6, 8, 10], 36)
    keys, costs = [], []
    for g in range(36):                                  # figure 2: 36 groups of 32 tasks
        shape = (128, 128) if g % 2 == 0 else (160, 160)
        keys += [('set%d' % (g // 2), shape)] * 32
        costs += [sharding.task_cost(shape[0] * shape[1], int(views[g // 2 * 2]), 20)] * 32
    for world in (1, 2, 4, 8):
        shards = sharding.partition_groups(keys, costs, world, setup_frames=24)
        assert sorted(i for s in shards for i in s) == list(range(len(keys)))
        assert shards == sharding.partition_groups(keys, costs, world, setup_frames=24)
        groups_per_rank = [{keys[i] for i in s} for s in shards]
        assert sum(len(g) for g in groups_per_rank) == 36          # no group is shared between ranks
        st = sharding.partition_stats(shards, costs, keys)
        assert st['groups_per_rank'] == [len(g) for g in groups_per_rank] and sum(st['tasks_per_rank']) == len(keys)
        assert max(st['cost_per_rank_rel']) < 1.35
    # a group of 300 tasks is cut into 4 pieces of 75 (>= 64), a group of 100 stays whole
    keys = [('big', (8, 8))] * 300 + [('small', (8, 8))] * 100
    shards = sharding.partition_groups(keys, [1.0] * 400, 4)
    sizes = sorted(len(s) for s in shards)
    assert sizes == [75, 75, 100, 150] and sorted(i for s in shards for i in s) == list(range(400))
    assert sharding.partition_groups([], [], 3) == [[], [], []]


def test_partitions_of_random_task_sets_are_complete_and_deterministic():
    """sharding.partition / partition_groups on random task sets, costs, world sizes and cut parameters: every task exactly once,
    the same answer twice, no group below the cut size on two ranks, no rank loaded beyond the ideal share plus the largest piece."""
    rng = np.random.default_rng(11)
    for trial in range(300):
        n_groups, world = int(rng.integers(0, 30)), int(rng.integers(1, 12))
        keys, costs = [], []
        for g in range(n_groups):
            n = int(rng.integers(1, 400))
            keys += [('g%d' % g, (int(rng.integers(1, 4)),))] * n
            costs += [float(rng.uniform(0.1, 50))] * n
        order = rng.permutation(len(keys))
        keys, costs = [keys[i] for i in order], [costs[i] for i in order]
        split_at, min_piece = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        setup = float(rng.choice([0.0, 24.0]))
        shards = sharding.partition_groups(keys, costs, world, setup_frames=setup, split_at=split_at, min_piece=min_piece)
        assert len(shards) == world and sorted(i for s in shards for i in s) == list(range(len(keys)))
        assert shards == sharding.partition_groups(keys, costs, world, setup_frames=setup, split_at=split_at, min_piece=min_piece)
        sizes = {}
        for k in keys:
            sizes[k] = sizes.get(k, 0) + 1
        for k, n in sizes.items():
            if n < split_at:
                assert sum(1 for s in shards if any(keys[i] == k for i in s)) == 1, (trial, k)
        plain = sharding.partition(costs, world)
        assert sorted(i for s in plain for i in s) == list(range(len(costs)))
        if costs:
            load = [sum(costs[i] for i in s) for s in plain]
            assert max(load) <= sum(costs) / world + max(costs) + 1e-9


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
    from comm_gloo import GlooComm
    res = sharding.run_sharded(tasks, costs, run_local, GlooComm(dist))
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
    from comm_gloo import GlooComm
    tasks, est = sweep.figure_2_sweep(objects, psf_sets, seeds=(0, 1, 2), iterations=5, comm=GlooComm(dist))
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


def test_product_sharding_is_torch_free():
    src = open(sharding.__file__).read()
    assert 'import torch' not in src and 'torch.' not in src.replace('torch.distributed.run', '')


def _id_worker(rank, path, q):
    q.put((rank, sharding.exchange_unique_id(rank, lambda: bytes(range(128)), path=path, timeout=30)))


def test_unique_id_rendezvous_by_file(tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    path = str(tmp_path / 'id')
    procs = [ctx.Process(target=_id_worker, args=(r, path, q)) for r in (1, 2, 0)]   # readers start first
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join()
    assert got == {0: bytes(range(128)), 1: bytes(range(128)), 2: bytes(range(128))}


def test_rendezvous_file_is_private_fresh_and_keyed_by_the_launch(tmp_path, monkeypatch):
    """VERDICT r02 / ADVICE: the id file sits in a 0700 directory of this user, its name carries the launcher's pid AND
    start time (a recycled pid, or a crashed earlier launch with the same port, gives another name), the communicator
    sequence number and the port; rank 0 never hands out a file it did not write (a stale one is removed first) and
    creates its own exclusively with mode 0600."""
    import stat
    monkeypatch.setenv('XDG_RUNTIME_DIR', str(tmp_path))
    monkeypatch.setenv('MASTER_PORT', '29999')
    d = sharding.rendezvous_dir()
    assert d.startswith(str(tmp_path)) and stat.S_IMODE(os.stat(d).st_mode) == 0o700
    p0, p1 = sharding.rendezvous_path(seq=0), sharding.rendezvous_path(seq=1)
    assert p0 != p1 and os.path.dirname(p0) == d and '_29999_' in os.path.basename(p0)
    start = open('/proc/%d/stat' % os.getppid()).read().rsplit(')', 1)[1].split()[19]
    assert os.path.basename(p0).startswith('rccl_%d_%s_' % (os.getppid(), start))
    monkeypatch.setenv('MASTER_PORT', '30000')
    assert sharding.rendezvous_path(seq=0) != p0
    with open(p0, 'wb') as f:                      # a stale id of the right size
        f.write(b'\xff' * 128)
    fresh = bytes(range(128))
    assert sharding.exchange_unique_id(0, lambda: fresh, path=p0) == fresh
    assert open(p0, 'rb').read() == fresh and stat.S_IMODE(os.stat(p0).st_mode) == 0o600
    assert sharding.exchange_unique_id(1, None, path=p0, timeout=5) == fresh
