"""Manual helper: the headline workload with a synchronisation after every step (bench.py today) against all steps
enqueued back to back with one synchronisation at the end."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness, _ = bench.workload(512)
B, steps = 256, 50
plan = _lib.DeconvPlan(psf, B, 512, 512, dtype='f32')
plan.set_object(np.broadcast_to(obj, (B, 512, 512)), brightness)
for w in range(3):
    plan.bench_cycles(20, 1, seed=w)
for rep in range(3):
    plan.ctx.synchronize(); t0 = time.perf_counter()
    for s in range(steps):
        plan.bench_cycles(20, 1, seed=10 + s)
    plan.ctx.synchronize(); a = time.perf_counter() - t0
    plan.ctx.synchronize(); t0 = time.perf_counter()
    plan.bench_cycles(20, steps, seed=10)
    plan.ctx.synchronize(); b = time.perf_counter() - t0
    print('sync per step: %.0f frames/s   one sync: %.0f frames/s' % (B * steps / a, B * steps / b), flush=True)
