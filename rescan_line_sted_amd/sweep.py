"""Parameter sweeps over independent simulations (BASELINE config 4: "all test
objects x doses x scan modes x seeds"), the workload of the reference's figure-2
script (line_sted_figure_2.py:29-57: one Deconvolver per PSF set and test image,
create_data_from_object + N x iterate), batched and -- optionally -- sharded over
GPUs.

A task is (object name, PSF-set name, seed).  All tasks that share a PSF set and an image
shape -- a GROUP -- become the frames of one device plan, whatever their seeds: a frame draws its
noise with the Philox key (task seed, object id) -- `rl_deconv_simulate_keyed` -- where the object
id is the position of the object's name in the sorted object names, so a task's result does not
depend on how the sweep was batched or sharded.

How a sweep runs on one rank (round 4; `run_tasks_device`): plans are built once per (PSF set, image
shape, batch, dtype, device) and kept (`plan_for`); every group is ENQUEUED with `rl_batch_submit` --
objects staged in page-locked memory, uploaded on a copy stream, simulated and deconvolved on the
device, the estimates written straight into one device buffer of the rank (`DeviceResults`: fp32 or
fp64, unpadded, task order) -- on one of a few contexts of the GPU in turn, so that the small launches
of neighbouring groups overlap; the host synchronises once, at the end.  Across ranks: whole groups
are dealt to the ranks (`sharding.partition_groups`: a plan's set-up is then paid by one rank), and
ONE `rl_comm_gather_device` brings the device buffers to the root, which downloads once.
"""
import ctypes
import hashlib

import numpy as np

from . import sharding
from ._lib import DTYPES, RL_F32, RNG_PHILOX, Context, DeconvPlan, check, lib, ptr


def make_tasks(objects, psf_sets, seeds):
    """All (object, psf set, seed) combinations, in a deterministic order."""
    return [(o, p, int(s)) for p in sorted(psf_sets) for s in seeds for o in sorted(objects)]


def task_costs(tasks, objects, psf_sets, iterations):
    return [sharding.task_cost(objects[o].shape[-2] * objects[o].shape[-1], len(psf_sets[p]), iterations)
            for o, p, _ in tasks]


def task_groups(tasks, objects):
    """The plan a task runs in: (PSF set, image shape)."""
    return [(p, tuple(objects[o].shape[-2:])) for o, p, _ in tasks]


def object_ids(objects):
    """Stable image ids for the Philox counter: the rank of each object name."""
    return {name: i for i, name in enumerate(sorted(objects))}


# ------------------------------------------------------------------ plans are built once and kept
PLAN_CACHE_MAX = 128
_plans = {}


_set_keys = {}      # id(list of PSF arrays) -> (the list, its arrays, per-array (sum, sum of squares), (digest, views))


def _psf_set_key(psfs):
    """(sha1 of the PSF set, its views as (1, py, px) float64 arrays).  Hashing 18 sets of up to ten 107 x 107 float64 PSFs is milliseconds per sweep -- as much
    as the device needs for a quarter of it -- so a set that is the SAME list of the SAME arrays as last time, with unchanged sums and
    sums of squares (an in-place edit shows there), is not hashed again."""
    ent = _set_keys.get(id(psfs))
    probe = tuple((float(np.sum(p)), float(np.square(p).sum())) for p in psfs)     # (no BLAS call: its thread pool costs more than the sums)
    if ent is not None and ent[0] is psfs and len(ent[1]) == len(psfs) and all(a is b for a, b in zip(ent[1], psfs)) and ent[2] == probe:
        return ent[3]
    # (the views of a set may differ in shape -- DeconvPlan embeds them in a common one -- so the key covers shapes and values)
    views = [np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape((1,) + np.shape(p)[-2:])) for p in psfs]
    h = hashlib.sha1()
    for v in views:
        h.update(repr(v.shape).encode())
        h.update(v.tobytes())
    val = (h.hexdigest(), views)
    if len(_set_keys) > 4 * PLAN_CACHE_MAX:
        _set_keys.clear()
    _set_keys[id(psfs)] = (psfs, list(psfs), probe, val)
    return val


def plan_for(psfs, batch, shape, dtype='f32', device=0, stream=0):
    """The plan of a (PSF set, image shape, batch): built on first use, kept for the next sweep (the reference builds its
    Deconvolvers once per figure too, line_sted_figure_2.py:39-45).  `stream`: which of the device's contexts it lives on."""
    digest, views = _psf_set_key(psfs)
    key = (digest, len(views), int(batch), tuple(shape), dtype, device, stream)
    plan = _plans.pop(key, None)
    if plan is None:
        plan = DeconvPlan(views, batch, shape[0], shape[1], dtype=dtype, device=device, stream=stream)
        while len(_plans) >= PLAN_CACHE_MAX:
            _plans.pop(next(iter(_plans)))          # the least recently used one
    _plans[key] = plan
    return plan


def clear_plans():
    _plans.clear()
    _set_keys.clear()


def unresolved_total(reset=False):
    """Predictions H(est) <= 0 the kept plans met in their sweeps so far (DeconvPlan.unresolved, include/rlsted.h
    rl_deconv_unresolved): 0 on data the plans' arithmetic resolves.  Synchronises the plans' contexts."""
    return sum(plan.unresolved(reset=reset) for plan in _plans.values())


class DeviceResults:
    """The estimates of a list of tasks in ONE device buffer (rl_device_alloc): image i at element offsets[i], shape shapes[i],
    arithmetic type `dtype` -- unpadded, in task order.  What rl_batch_submit writes and rl_comm_gather_device sends."""

    def __init__(self, shapes, dtype='f32', device=0):
        self.ctx = Context.get(device)
        self.shapes = [tuple(s) for s in shapes]
        self.dtype = dtype
        sizes = [s[0] * s[1] for s in self.shapes]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.n = int(self.offsets[-1])
        self.itemsize = 4 if DTYPES[dtype] == RL_F32 else 8
        self.dev = ctypes.c_void_p()
        check(lib.rl_device_alloc(self.ctx.handle, max(self.n, 1) * self.itemsize, ctypes.byref(self.dev)))

    def address(self, i):
        return ctypes.c_void_p(self.dev.value + int(self.offsets[i]) * self.itemsize)

    def download(self):
        """ONE device-to-host transfer; the list of (ny, nx) float64 estimates."""
        flat = np.empty(self.n, dtype=np.float64)
        check(lib.rl_device_download(self.ctx.handle, self.dev, DTYPES[self.dtype], self.n, ptr(flat)))
        return split_flat(flat, self.shapes)

    def free(self):
        if getattr(self, 'dev', None) is not None and self.dev.value and lib is not None:    # (lib: gone at interpreter shutdown)
            lib.rl_device_free(self.ctx.handle, self.dev)
            self.dev = ctypes.c_void_p()

    __del__ = free


def split_flat(flat, shapes):
    out, o = [], 0
    for s in shapes:
        out.append(flat[o:o + s[0] * s[1]].reshape(s))
        o += s[0] * s[1]
    return out


SWEEP_STREAMS = 4      # contexts of one GPU the groups of a sweep are dealt to in turn (their launches overlap); measured on
#                        config 4's 1152 tasks: 1 / 2 / 3 / 4 / 6 / 8 contexts 43.7 / 34.0 / 31.6 / 30.7 / 31.4 / 33.3 ms


def run_tasks_device(tasks, objects, psf_sets, iterations, total_brightness=5e10, dtype='f32', device=0,
                     max_frames_per_plan=256, streams=SWEEP_STREAMS, timing=None):
    """Enqueue the tasks on one GPU (group by group, `rl_batch_submit`), synchronise once; returns their DeviceResults.
    timing (dict, optional): receives 'enqueue_s', the host's share (staging + launches) before the one synchronisation."""
    import time
    t_start = time.perf_counter()
    ids = object_ids(objects)
    res = DeviceResults([objects[o].shape[-2:] for o, _, _ in tasks], dtype, device)
    groups = {}
    for idx, key in enumerate(task_groups(tasks, objects)):
        groups.setdefault(key, []).append(idx)
    used = set()
    n_sub = 0
    for (p, shape), idxs in groups.items():
        for start in range(0, len(idxs), max_frames_per_plan):
            part = idxs[start:start + max_frames_per_plan]
            # (the tasks of a piece are consecutive in `res`: a piece is a run of one group's tasks in task order only if
            # the caller's tasks are grouped; in general every task is submitted to its own address)
            stream = n_sub % max(1, streams)
            n_sub += 1
            plan = plan_for(psf_sets[p], len(part), shape, dtype, device, stream)
            used.add(stream)
            frames = [np.ascontiguousarray(np.asarray(objects[tasks[i][0]], dtype=np.float64).reshape(shape)) for i in part]
            runs = _consecutive_runs(part, res)
            for a, b in runs:                         # maximal runs of tasks that are neighbours in the result buffer
                plan.batch_submit(frames[a:b], total_brightness, [tasks[i][2] for i in part[a:b]], [ids[tasks[i][0]] for i in part[a:b]],
                                  iterations, res.address(part[a]), dtype, rng=RNG_PHILOX)
    if timing is not None:
        timing['enqueue_s'] = time.perf_counter() - t_start
    for st in used:
        Context.get(device, st).synchronize()
    return res


def _consecutive_runs(part, res):
    """[(a, b)): maximal ranges of `part` whose task indices are consecutive (their images are neighbours in `res`)."""
    runs, a = [], 0
    for k in range(1, len(part) + 1):
        if k == len(part) or part[k] != part[k - 1] + 1:
            runs.append((a, k))
            a = k
    return runs


def sort_by_group(tasks, objects):
    """Task indices ordered so that the tasks of one (PSF set, shape) group are neighbours: one submit per piece."""
    keys = task_groups(tasks, objects)
    return sorted(range(len(tasks)), key=lambda i: (keys[i][0], keys[i][1], i))


def run_tasks(tasks, objects, psf_sets, iterations, total_brightness=5e10, dtype='f32', device=0,
              max_frames_per_plan=256):
    """Run tasks on one GPU.  Returns a list of (ny, nx) estimates in task order."""
    order = sort_by_group(tasks, objects)
    res = run_tasks_device([tasks[i] for i in order], objects, psf_sets, iterations, total_brightness, dtype, device,
                           max_frames_per_plan)
    est = res.download()
    res.free()
    out = [None] * len(tasks)
    for k, i in enumerate(order):
        out[i] = est[k]
    return out


def pad_stack(images, shape):
    """Stack 2-D images of different sizes into one (n, shape[0], shape[1]) array, top-left aligned
    and zero filled."""
    out = np.zeros((len(images),) + tuple(shape), dtype=np.float64)
    for k, im in enumerate(images):
        out[k, :im.shape[0], :im.shape[1]] = im
    return out


PLAN_SETUP_FRAMES = 24     # what building a plan costs a rank, in frames of its group's task cost (sharding.partition_groups)


def shard_sweep(tasks, objects, psf_sets, iterations, world):
    """The sweep's partition: whole (PSF set, shape) groups per rank -- a plan is then built, and its set-up paid, on one rank
    only -- groups of 128 tasks and more in pieces of at least 64.  Returns (shards, costs): task indices per rank, group sorted."""
    costs = task_costs(tasks, objects, psf_sets, iterations)
    keys = task_groups(tasks, objects)
    shards = sharding.partition_groups(keys, costs, world, setup_frames=PLAN_SETUP_FRAMES)
    order = {i: k for k, i in enumerate(sort_by_group(tasks, objects))}
    return [sorted(s, key=lambda i: order[i]) for s in shards], costs


def figure_2_sweep(objects, psf_sets, seeds, iterations, total_brightness=5e10, dtype='f32',
                   device=0, comm=None, info=None):
    """The sweep, sharded over the ranks of `comm` (sharding.RcclComm, or anything with its
    interface) when given.  Returns (tasks, estimates) on rank 0 and (tasks, None) elsewhere;
    estimates is an array (n_tasks, ny, nx) when all objects share a shape, otherwise a list of
    (ny, nx) arrays in task order.  The one gather carries the ranks' estimates unpadded: from device buffer to device
    buffer in the plans' arithmetic type (`comm.gather_device`), or -- a stand-in communicator without it -- as flat host
    arrays.  info (dict, optional) receives the partition's statistics."""
    tasks = make_tasks(objects, psf_sets, seeds)
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    shards, costs = shard_sweep(tasks, objects, psf_sets, iterations, world)
    mine = [tasks[i] for i in shards[rank]]
    shapes = [tuple(objects[o].shape[-2:]) for o, _, _ in tasks]
    pix = [sum(shapes[i][0] * shapes[i][1] for i in sh) for sh in shards]
    if info is not None:
        info.update(sharding.partition_stats(shards, costs, task_groups(tasks, objects)))
    if comm is not None and hasattr(comm, 'gather_device'):
        res = run_tasks_device(mine, objects, psf_sets, iterations, total_brightness, dtype, device)
        if info is not None:
            info['unresolved_predictions_this_rank'] = unresolved_total(reset=True)
        flat = comm.gather_device(res, pix, 0)       # root: host float64, rank-major; others: None
        res.free()
    else:
        local = run_tasks(mine, objects, psf_sets, iterations, total_brightness, dtype, device) if mine else []
        flat = np.concatenate([np.asarray(e, dtype=np.float64).ravel() for e in local]) if local else np.zeros(0)
        if comm is not None:
            flat = comm.gather(flat, pix, 0)
    if flat is None:
        return tasks, None
    order = [i for sh in shards for i in sh]
    parts = split_flat(np.asarray(flat), [shapes[i] for i in order])
    est = [None] * len(tasks)
    for k, i in enumerate(order):
        est[i] = parts[k]
    if len(set(shapes)) == 1:
        return tasks, np.stack(est) if est else np.zeros((0,) + (shapes[0] if shapes else (0, 0)))
    return tasks, est
