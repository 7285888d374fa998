"""Manual helper (not a test): throughput of the other BASELINE modes/precisions."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
cases = [('point V=1 f32', [g['2p0x_lr/point_sted_psf'][0]], 'f32', 128, 512),
         ('point V=1 f64', [g['2p0x_lr/point_sted_psf'][0]], 'f64', 64, 512),
         ('line-rescan V=4 f32', [p[None] for p in g['2p0x_lr/line_sted_psfs'][:, 0]], 'f32', 64, 512),
         ('line-rescan V=3 f32', [p[None] for p in g['1p5x_lr/line_sted_psfs'][:, 0]], 'f32', 64, 512),
         ('point V=1 f32 128x128', [g['2p0x_lr/point_sted_psf'][0]], 'f32', 1024, 128)]
for name, psfs, dtype, B, n in cases:
    o = obj if n == 512 else objs['astronaut'][0].astype(np.float64)
    plan = _lib.DeconvPlan(psfs, B, n, n, dtype=dtype)
    plan.set_object(np.broadcast_to(o, (B, n, n)), 5e10 * (16 if n == 512 else 1))
    plan.bench_cycles(20, 1, seed=1)
    t0 = time.perf_counter(); ms = plan.bench_cycles(20, 3, seed=2); el = time.perf_counter() - t0
    V = len(psfs)
    alg = 4 * n * n * ((2 * V + 2) + 20 * (3 * V + 4))
    fps = 3 * B / el
    print('%-24s B=%4d  %8.0f frames/s   alg %.1f MB/frame -> %.1f%% of 8 TB/s' % (name, B, fps, alg / 1e6, alg * fps / 8e12 * 100))
    del plan
