"""Manual helper: the frame-pair Richardson-Lucy path (RLSTED_PAIR=1) against the default path -- same measurement,
K iterations, estimate difference; then frames/s of both."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness, _ = bench.workload(512)
rng = np.random.default_rng(3)
B = 6
objs = np.concatenate([obj[None], rng.random((B - 1, 512, 512)) * 200])
res = {}
for pair in ('0', '1'):
    os.environ['RLSTED_PAIR'] = pair
    for dtype in ('f64', 'f32'):
        plan = _lib.DeconvPlan(psf, B, 512, 512, dtype=dtype)
        plan.set_object(objs, brightness)
        plan.simulate(seed=5)
        plan.iterate(1)
        e1 = plan.estimate()
        plan.iterate(9)
        res[(pair, dtype)] = (e1, plan.estimate(), plan.measurement())
for dtype in ('f64', 'f32'):
    a, b = res[('0', dtype)], res[('1', dtype)]
    print(dtype, 'measurement equal', np.array_equal(a[2], b[2]), 'K=1', np.abs(a[0] - b[0]).max() / a[0].max(), 'K=10', np.abs(a[1] - b[1]).max() / a[1].max(), flush=True)
for pr in ('0', '1'):
    a, b = res[(pr, 'f32')][1], res[(pr, 'f64')][1]
    print('f32 vs f64, pair', pr, [float(np.abs(a[f] - b[f]).max() / b[f].max()) for f in range(B)])
for pair in ():
    os.environ['RLSTED_PAIR'] = pair
    plan = _lib.DeconvPlan(psf, 256, 512, 512, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (256, 512, 512)), brightness)
    plan.bench_cycles(20, 2, seed=1)
    t0 = time.perf_counter(); plan.bench_cycles(20, 20, seed=2); el = time.perf_counter() - t0
    print('pair', pair, '%.0f frames/s' % (256 * 20 / el), flush=True)
    del plan
