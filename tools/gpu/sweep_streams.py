"""Manual helper (not a test): BASELINE config 4 on one GPU as a function of the number of contexts the groups are dealt to
(sweep.run_tasks_device streams=...), plans kept; prints ms per 1152-task sweep and the host's enqueue time share.
    python3 tools/gpu/sweep_streams.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import sweep  # noqa: E402
sys.path.insert(0, ROOT)
import bench  # noqa: E402

objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
objects = {n: objs[n][0].astype(np.float64) for n in ('astronaut', 'cat', 'lines', 'rings')}
psf_sets = bench.fig2_psf_sets(False)
tasks = sweep.make_tasks(objects, psf_sets, range(16))
order = sweep.sort_by_group(tasks, objects)
tasks = [tasks[i] for i in order]
for streams in (1, 2, 3, 4, 6, 8):
    sweep.run_tasks_device(tasks, objects, psf_sets, 20, 5e10, 'f32', 0, streams=streams).free()     # builds the plans
    ts, enq = [], []
    for _ in range(5):
        tm = {}
        t0 = time.perf_counter()
        res = sweep.run_tasks_device(tasks, objects, psf_sets, 20, 5e10, 'f32', 0, streams=streams, timing=tm)
        ts.append(time.perf_counter() - t0)
        enq.append(tm['enqueue_s'])
        res.free()
    print('streams %d: median %.1f ms  min %.1f ms  (%d tasks -> %.1f k frames/s); host enqueue %.1f ms of it' % (
        streams, np.median(ts) * 1e3, min(ts) * 1e3, len(tasks), len(tasks) / np.median(ts) / 1e3, np.median(enq) * 1e3), flush=True)
