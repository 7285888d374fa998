"""Multi-GPU sharding of independent simulations.

Every (test object x dose x scan mode x noise seed) task is independent
(separate Deconvolver objects in the reference, line_sted_figure_2.py:39-56),
so the task list is partitioned over ranks -- one process per GPU -- with no
collective in the data path and ONE gather of the final estimates at the end
(RCCL over xGMI when the process group's backend is 'nccl', gloo on CPU).
torch.distributed is used purely as the communication plumbing.
"""
import numpy as np


def task_cost(n_pix, n_psf, iterations):
    """Relative cost of one frame: convolutions per cycle x pixels (views per task
    vary 1..10 in figure 2, so the partition must be cost weighted)."""
    return float(n_pix) * n_psf * (2 + 2 * iterations)


def partition(costs, world_size):
    """Greedy longest-processing-time partition.  Returns a list (one entry per
    rank) of task indices; deterministic, every task assigned exactly once."""
    costs = np.asarray(costs, dtype=np.float64)
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    return [sorted(s) for s in shards]


class DeviceArray:
    """Exposes a device buffer owned by a DeconvPlan through
    __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def gather_to_root(local, counts, dist, root=0):
    """Gather per-rank result stacks on `root`.

    local  : numpy array or torch tensor (n_local, ...) of this rank's results
    counts : number of results on every rank (len == world size)
    dist   : an initialised torch.distributed module (any backend)
    Returns the concatenated stack on root (numpy if `local` was numpy), None elsewhere.
    """
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    was_numpy = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if was_numpy else local.contiguous()
    if dist.get_backend() == 'nccl' and not t.is_cuda:
        t = t.cuda()
    nmax = int(max(counts))
    item = tuple(t.shape[1:])
    pad = torch.zeros((nmax,) + item, dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
    dist.gather(pad, bufs, dst=root)
    if rank != root:
        return None
    out = torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0)
    return out.cpu().numpy() if was_numpy else out


def unshard(shards, gathered):
    """Reorder a root-gathered stack (rank-major) back into task order."""
    order = [i for s in shards for i in s]
    out = np.empty_like(gathered)
    out[np.asarray(order)] = gathered
    return out


def run_sharded(tasks, costs, run_local, dist=None):
    """Partition `tasks`, run this rank's share with run_local(list_of_tasks) ->
    array (n_local, ...), gather on rank 0 and return results in task order
    (rank 0) or None (other ranks).  With dist=None runs everything locally."""
    if dist is None:
        return np.asarray(run_local(list(tasks)))
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = partition(costs, world)
    mine = [tasks[i] for i in shards[rank]]
    local = np.asarray(run_local(mine)) if mine else None
    if local is None:       # a rank without tasks still joins the gather with an empty stack
        import torch
        shape = [None]
        dist.broadcast_object_list(shape, src=next(r for r in range(world) if shards[r]))
        local = np.zeros((0,) + tuple(shape[0]), dtype=np.float64)
    else:
        import torch
        first = next(r for r in range(world) if shards[r])
        shape = [tuple(local.shape[1:])] if rank == first else [None]
        dist.broadcast_object_list(shape, src=first)
    gathered = gather_to_root(local, [len(s) for s in shards], dist)
    return unshard(shards, gathered) if rank == 0 else None
