"""Manual helper (not a test): throughput at the figure-2 object sizes (128x128, 160x160)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
point = [g['2p0x_lr/point_sted_psf'][0]]
line4 = [p[None] for p in g['2p0x_lr/line_sted_psfs'][:, 0]]
for name, psfs, key, B in (('128^2 point V=1', point, 'astronaut', 2048), ('128^2 line-rescan V=4', line4, 'astronaut', 1024),
                           ('160^2 point V=1', point, 'cat', 2048), ('160^2 line-rescan V=4', line4, 'cat', 1024)):
    o = objs[key][0].astype(np.float64)
    n = o.shape[0]
    plan = _lib.DeconvPlan(psfs, B, n, n, dtype='f32')
    plan.set_object(np.broadcast_to(o, (B, n, n)), 5e10)
    plan.bench_cycles(20, 1, seed=1)
    t0 = time.perf_counter(); plan.bench_cycles(20, 3, seed=2); el = time.perf_counter() - t0
    V = len(psfs)
    alg = 4 * n * n * ((2 * V + 2) + 20 * (3 * V + 4))
    fps = 3 * B / el
    print('%-24s L=%d B=%4d  %9.0f frames/s   alg %.1f MB/frame -> %.1f%% of 8 TB/s' % (name, plan.info()['ly'], B, fps, alg / 1e6, alg * fps / 8e12 * 100), flush=True)
    del plan
