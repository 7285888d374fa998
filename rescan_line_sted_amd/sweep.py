"""Parameter sweeps over independent simulations (BASELINE config 4: "all test
objects x doses x scan modes x seeds"), the workload of the reference's figure-2
script (line_sted_figure_2.py:29-57: one Deconvolver per PSF set and test image,
create_data_from_object + N x iterate), batched and -- optionally -- sharded over
GPUs with `sharding.run_sharded`.

A task is (object name, PSF-set name, seed).  All tasks that share a PSF set and an image
shape become the frames of one device plan, whatever their seeds: a frame draws its noise with
the Philox key (task seed, object id) -- `rl_deconv_simulate_keyed` -- where the object id is the
position of the object's name in the sorted object names, so a task's result does not depend on
how the sweep was batched or sharded.
"""
import numpy as np

from . import sharding
from ._lib import DeconvPlan, RNG_PHILOX


def make_tasks(objects, psf_sets, seeds):
    """All (object, psf set, seed) combinations, in a deterministic order."""
    return [(o, p, int(s)) for p in sorted(psf_sets) for s in seeds for o in sorted(objects)]


def task_costs(tasks, objects, psf_sets, iterations):
    return [sharding.task_cost(objects[o].shape[-2] * objects[o].shape[-1], len(psf_sets[p]), iterations)
            for o, p, _ in tasks]


def object_ids(objects):
    """Stable image ids for the Philox counter: the rank of each object name."""
    return {name: i for i, name in enumerate(sorted(objects))}


def run_tasks(tasks, objects, psf_sets, iterations, total_brightness=5e10, dtype='f32', device=0,
              max_frames_per_plan=256):
    """Run tasks on one GPU.  Returns a list of (ny, nx) estimates in task order."""
    out = [None] * len(tasks)
    ids = object_ids(objects)
    groups = {}
    for idx, (o, p, s) in enumerate(tasks):
        shape = objects[o].shape[-2:]
        groups.setdefault((p, shape), []).append(idx)
    for (p, shape), idxs in groups.items():
        for start in range(0, len(idxs), max_frames_per_plan):
            part = idxs[start:start + max_frames_per_plan]
            frames = np.stack([np.asarray(objects[tasks[i][0]], dtype=np.float64).reshape(shape) for i in part])
            plan = DeconvPlan(psf_sets[p], len(part), shape[0], shape[1], dtype=dtype, device=device)
            est = plan.batch_run(frames, total_brightness, [tasks[i][2] for i in part], [ids[tasks[i][0]] for i in part],
                                 iterations, rng=RNG_PHILOX)      # rl_batch_run of the C ABI
            for k, i in enumerate(part):
                out[i] = est[k]
            del plan
    return out


def pad_stack(images, shape):
    """Stack 2-D images of different sizes into one (n, shape[0], shape[1]) array, top-left aligned
    and zero filled, so that one gather can carry them."""
    out = np.zeros((len(images),) + tuple(shape), dtype=np.float64)
    for k, im in enumerate(images):
        out[k, :im.shape[0], :im.shape[1]] = im
    return out


def figure_2_sweep(objects, psf_sets, seeds, iterations, total_brightness=5e10, dtype='f32',
                   device=0, comm=None):
    """The sweep, sharded over the ranks of `comm` (sharding.RcclComm, or anything with its
    interface) when given.  Returns (tasks, estimates) on rank 0 and (tasks, None) elsewhere;
    estimates is an array (n_tasks, ny, nx) when all objects share a shape, otherwise a list of
    (ny, nx) arrays in task order (the figure's test objects are 128x128 and 160x160: the
    single gather then carries frames zero-padded to the largest shape, cropped on rank 0)."""
    tasks = make_tasks(objects, psf_sets, seeds)
    costs = task_costs(tasks, objects, psf_sets, iterations)
    shapes = {tuple(objects[o].shape[-2:]) for o, _, _ in tasks}
    big = (max(s[0] for s in shapes), max(s[1] for s in shapes))

    def run_local(mine):
        return pad_stack(run_tasks(mine, objects, psf_sets, iterations, total_brightness, dtype, device), big)
    padded = sharding.run_sharded(tasks, costs, run_local, comm)
    if padded is None:
        return tasks, None
    if len(shapes) == 1:
        return tasks, padded
    return tasks, [padded[i][:objects[o].shape[-2], :objects[o].shape[-1]] for i, (o, _, _) in enumerate(tasks)]
