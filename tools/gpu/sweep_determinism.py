"""Manual stress (not a test): BASELINE config 4 (1152 tasks) run repeatedly on 1 ... 8 contexts of one GPU -- every run must return
bit for bit the estimates of the single-context run (the groups' composition does not depend on the number of contexts; their
launches overlap on up to 8 hardware queues).
    python3 tools/gpu/sweep_determinism.py [REPEATS]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import sweep  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
objects = {n: objs[n][0].astype(np.float64) for n in ('astronaut', 'cat', 'lines', 'rings')}
psf_sets = bench.fig2_psf_sets(False)
tasks = sweep.make_tasks(objects, psf_sets, range(16))
tasks = [tasks[i] for i in sweep.sort_by_group(tasks, objects)]
res = sweep.run_tasks_device(tasks, objects, psf_sets, 20, 5e10, 'f32', 0, streams=1)
ref = np.concatenate([e.ravel() for e in res.download()])
res.free()
assert np.isfinite(ref).all() and ref.max() > 0
bad = 0
for streams in (1, 2, 3, 4, 6, 8):
    for r in range(reps):
        res = sweep.run_tasks_device(tasks, objects, psf_sets, 20, 5e10, 'f32', 0, streams=streams)
        got = np.concatenate([e.ravel() for e in res.download()])
        res.free()
        if not np.array_equal(got, ref):
            bad += 1
            d = np.abs(got - ref)
            print('streams %d run %d: %d values differ, max %.3e (of %.3e)' % (streams, r, int((d > 0).sum()), d.max(), ref.max()), flush=True)
    print('streams %d: %d runs done' % (streams, reps), flush=True)
print('%d runs differed from the single-context run' % bad)
