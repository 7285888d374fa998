"""Manual helper (not a test): the direct 2-D stencil (RLSTED_DIRECT=2) against the FFT path (RLSTED_DIRECT=0) for small PSFs that are
not rank 1 -- time per frame-iteration of the Richardson-Lucy loop, 512 x 512, f32 and float64, 1 and 2 views.
    python3 tools/gpu/direct_vs_fft.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402

rng = np.random.default_rng(1)
n, B, K = 512, 128, 20
obj = rng.random((B, n, n)) * 100
for dtype in ('f32', 'f64'):
    for V in (1, 2):
        for taps in (3, 5, 7, 9, 11, 13, 15):
            psfs = [rng.random((1, taps, taps)) + 0.05 for _ in range(V)]
            t = {}
            for mode in ('0', '2'):
                os.environ['RLSTED_DIRECT'] = mode
                plan = _lib.DeconvPlan(psfs, B, n, n, dtype=dtype)
                del os.environ['RLSTED_DIRECT']
                assert plan.strategy()['direct_stencil'] == (mode == '2')
                plan.set_object(obj, 1e9)
                plan.simulate(seed=1)
                plan.iterate(K)
                plan.ctx.synchronize()
                best = 1e9
                for _ in range(3):
                    plan.reset_estimate()
                    plan.ctx.synchronize()
                    t0 = time.perf_counter()
                    plan.iterate(K)
                    plan.ctx.synchronize()
                    best = min(best, time.perf_counter() - t0)
                t[mode] = best / (B * K) * 1e6
                del plan
            print('%s  V = %d  %2d x %2d taps: FFT path %6.2f us per frame-iteration, direct stencil %6.2f  (%.2fx)' % (
                dtype, V, taps, taps, t['0'], t['2'], t['0'] / t['2']), flush=True)
