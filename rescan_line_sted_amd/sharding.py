"""Multi-GPU sharding of independent simulations.

Every (test object x dose x scan mode x noise seed) task is independent
(separate Deconvolver objects in the reference, line_sted_figure_2.py:39-56),
so the task list is partitioned over ranks -- one process per GPU -- with no
collective in the data path and ONE gather of the final estimates at the end.

The transport is RCCL over xGMI through the C ABI (`rl_comm_*`, `rl_gather*` of
include/rlsted.h): `RcclComm`.  Anything with the same four members --
`rank`, `world`, `barrier()`, `allreduce_max(x)`, `gather(array, counts, root)` --
can stand in for it (the CPU tests use a gloo-backed one, tests/comm_gloo.py).
"""
import ctypes
import os
import tempfile
import time

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


def partition_groups(keys, costs, world_size, setup_frames=0.0, split_at=128, min_piece=64):
    """Partition tasks over ranks BY GROUP (keys[i]: the group -- device plan -- task i runs in): every rank that touches a
    group pays the plan's set-up (PSF spectra, normaliser, buffers), so whole groups are dealt, largest first, to the least
    loaded rank; only groups of `split_at` tasks and more are cut, into pieces of at least `min_piece` tasks.  A piece
    costs the sum of its tasks' costs plus `setup_frames` times its mean task cost.  Returns task indices per rank (sorted);
    deterministic, every task assigned exactly once."""
    costs = np.asarray(costs, dtype=np.float64)
    groups = {}
    for i, k in enumerate(keys):
        groups.setdefault(k, []).append(i)
    pieces = []
    for k in sorted(groups, key=repr):
        idx = groups[k]
        n_pieces = 1 if len(idx) < split_at else max(1, len(idx) // max(1, min_piece))
        for part in np.array_split(np.asarray(idx), n_pieces):
            part = [int(i) for i in part]
            c = float(costs[part].sum())
            pieces.append((c + setup_frames * c / len(part), part))
    order = sorted(range(len(pieces)), key=lambda j: (-pieces[j][0], pieces[j][1][0]))
    load = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for j in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        shards[r].extend(pieces[j][1])
        load[r] += pieces[j][0]
    return [sorted(s) for s in shards]


def partition_stats(shards, costs, keys):
    """What a bench line reports about a partition: tasks, groups and relative cost per rank."""
    total = float(sum(costs)) or 1.0
    world = len(shards)
    return {'tasks_per_rank': [len(s) for s in shards],
            'groups_per_rank': [len({keys[i] for i in s}) for s in shards],
            'cost_per_rank_rel': [round(sum(costs[i] for i in s) / (total / world), 4) for s in shards]}


class DeviceArray:
    """Exposes a device buffer owned by a DeconvPlan through
    __cuda_array_interface__ so other libraries can wrap it without a copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def _launcher_nonce():
    """Identifies THIS launch: the launcher's pid together with its start time (clock ticks since boot, field 22 of
    /proc/<pid>/stat -- a recycled pid gets another value), the rendezvous port, the launcher's run id and restart count.
    Every rank of one launch computes the same string; a crashed earlier launch, even with the same pid and port, another."""
    ppid = os.getppid()
    start = '0'
    try:
        with open('/proc/%d/stat' % ppid) as f:
            start = f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return '%d_%s_%s_%s_%s' % (ppid, start, os.environ.get('MASTER_PORT', '0'),
                               ''.join(c for c in os.environ.get('TORCHELASTIC_RUN_ID', 'none') if c.isalnum() or c in '-_')[:40],
                               os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'))


def rendezvous_dir():
    """A directory only this user can enter (0700): the id file cannot be pre-created or replaced by someone else."""
    base = os.environ.get('XDG_RUNTIME_DIR', '')
    if not (base and os.path.isdir(base) and os.access(base, os.W_OK | os.X_OK)):
        base = tempfile.gettempdir()
    d = os.path.join(base, 'rlsted-%d' % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError('%s is not a private directory of this user' % d)
    return d


_comm_seq = [0]     # communicators created by this process so far: every rank creates them in the same order


def rendezvous_path(seq=None):
    """Where rank 0 leaves the RCCL unique id for the other ranks of this launch."""
    return os.path.join(rendezvous_dir(), 'rccl_%s_%d.id' % (_launcher_nonce(), _comm_seq[0] if seq is None else seq))


def exchange_unique_id(rank, make_id, path=None, timeout=300.0, nbytes=128):
    """Rank 0 removes whatever is at `path`, creates the id and publishes it atomically (exclusive create, mode 0600,
    then rename); the others wait for a file of the right size."""
    path = path or rendezvous_path()
    if rank == 0:
        try:
            os.remove(path)            # never hand out a file this call did not write
        except OSError:
            pass
        blob = make_id()
        tmp = '%s.%d.tmp' % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(blob)
        os.replace(tmp, path)
        return blob
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                blob = f.read()
            if len(blob) == nbytes:
                return blob
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError('no RCCL unique id at %s after %.0f s' % (path, timeout))
        time.sleep(0.01)


class RcclComm:
    """rl_comm: one per process, created collectively by all ranks."""

    def __init__(self, rank, world, device=0, path=None):
        from . import _lib
        self._lib = _lib
        self.rank, self.world = int(rank), int(world)
        self.ctx = _lib.Context.get(device)

        def make_id():
            buf = ctypes.create_string_buffer(128)
            _lib.check(_lib.lib.rl_comm_unique_id(buf))
            return buf.raw
        self._path = path or rendezvous_path()
        _comm_seq[0] += 1
        blob = exchange_unique_id(self.rank, make_id, self._path)
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.lib.rl_comm_create(self.ctx.handle, self.rank, self.world, ctypes.c_char_p(blob),
                                           ctypes.byref(self.handle)))
        self.barrier()              # every rank has read the id
        if self.rank == 0:
            try:
                os.remove(self._path)
            except OSError:
                pass

    @classmethod
    def from_env(cls, device=None):
        """Ranks as torch.distributed.run (or any launcher) exports them: RANK, WORLD_SIZE, LOCAL_RANK."""
        rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
        return cls(rank, world, int(os.environ.get('LOCAL_RANK', '0')) if device is None else device)

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.lib.rl_comm_destroy(self.handle)
            self.handle = None

    __del__ = close

    def barrier(self):
        self._lib.check(self._lib.lib.rl_comm_barrier(self.handle))

    def allreduce_max(self, x):
        v = ctypes.c_double(float(x))
        self._lib.check(self._lib.lib.rl_comm_allreduce_max(self.handle, ctypes.byref(v)))
        return v.value

    def gather(self, local, counts, root=0):
        """Host arrays: rank r contributes counts[r] items (local.shape[0] == counts[rank]); returns the
        rank-major concatenation on root (float64), None elsewhere."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        item = int(np.prod(local.shape[1:], dtype=np.int64)) if local.ndim > 1 else 1
        sizes = (ctypes.c_size_t * self.world)(*[int(c) * item for c in counts])
        out = np.empty((int(sum(counts)),) + tuple(local.shape[1:]), dtype=np.float64) if self.rank == root else None
        self._lib.check(self._lib.lib.rl_comm_gather_host(
            self.handle, self._lib.ptr(local) if local.size else None, sizes, int(root),
            self._lib.ptr(out) if out is not None and out.size else None))
        return out

    def gather_device(self, results, counts, root=0):
        """The sweep's gather: counts[r] elements of every rank's sweep.DeviceResults buffer, device to device
        (rl_comm_gather_device: unpadded, the plans' arithmetic type), then ONE download on the root.  Returns the
        rank-major float64 array there, None elsewhere."""
        from .sweep import DeviceResults      # (the root's receive buffer is one more of them)
        total = int(sum(counts))
        sizes = (ctypes.c_size_t * self.world)(*[int(c) for c in counts])
        dt = self._lib.DTYPES[results.dtype]
        if self.rank != root:
            self._lib.check(self._lib.lib.rl_comm_gather_device(self.handle, results.dev, sizes, dt, int(root), None))
            return None
        recv = DeviceResults([(1, total)], results.dtype, self.ctx.device)
        try:
            self._lib.check(self._lib.lib.rl_comm_gather_device(self.handle, results.dev, sizes, dt, int(root), recv.dev))
            self.last_gather_bytes = total * recv.itemsize
            return recv.download()[0].reshape(-1)
        finally:
            recv.free()

    def bcast(self, array, root=0):
        """A float64 array of the root, on every rank (rl_comm_bcast_host); `array` must have the same shape everywhere."""
        a = np.ascontiguousarray(array, dtype=np.float64)
        if a is array:
            a = a.copy()
        self._lib.check(self._lib.lib.rl_comm_bcast_host(self.handle, self._lib.ptr(a), a.size, int(root)))
        return a

    def gather_plan(self, plan, counts, which='estimate', root=0, to_host=True):
        """The first counts[r] frames of every rank's plan buffer, straight from device memory (rl_gather).
        to_host=False leaves the result on the root's device and returns (pointer, elements, dtype)."""
        cs = (ctypes.c_int * self.world)(*[int(c) for c in counts])
        idx = plan.BUFFERS[which]
        if not to_host:
            p, n, dt = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int()
            self._lib.check(self._lib.lib.rl_gather_device(self.handle, plan.handle, idx, int(root), cs, ctypes.byref(p),
                                                           ctypes.byref(n), ctypes.byref(dt)))
            return (p.value, n.value, dt.value)
        shape = (plan.ny, plan.nx) if idx in (0, 3) else (plan.V, plan.ny, plan.nx)
        out = np.empty((int(sum(counts)),) + shape, dtype=np.float64) if self.rank == root else None
        self._lib.check(self._lib.lib.rl_gather(self.handle, plan.handle, idx, int(root), cs,
                                                self._lib.ptr(out) if out is not None else None))
        return out


def unshard(shards, gathered):
    """Reorder a root-gathered stack (rank-major) back into task order."""
    order = [i for s in shards for i in s]
    out = np.empty_like(gathered)
    out[np.asarray(order)] = gathered
    return out


def run_sharded(tasks, costs, run_local, comm=None):
    """Partition `tasks`, run this rank's share with run_local(list_of_tasks) ->
    array (n_local, ...), gather on rank 0 and return results in task order
    (rank 0) or None (other ranks).  With comm=None runs everything locally."""
    if comm is None:
        return np.asarray(run_local(list(tasks)))
    world, rank = comm.world, comm.rank
    shards = partition(costs, world)
    mine = [tasks[i] for i in shards[rank]]
    # a rank without tasks joins the gather with an empty stack (the greedy partition fills rank 0
    # first, so the root always knows the item shape when there is any task at all)
    local = np.asarray(run_local(mine), dtype=np.float64) if mine else np.zeros((0,), dtype=np.float64)
    gathered = comm.gather(local, [len(s) for s in shards], 0)
    return unshard(shards, gathered) if rank == 0 else None
