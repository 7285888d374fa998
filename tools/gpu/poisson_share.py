"""Manual helper (not a test): what the Poisson draw costs a whole cycle -- the headline workload with the device generator and
with rng = none (noisy = noiseless + 1e-9), interleaved rounds in one process.
    python3 tools/gpu/poisson_share.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psfs = [g['2p0x_lr/point_sted_psf'][0]]
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
B = 1024
plan = _lib.DeconvPlan(psfs, B, 512, 512, dtype='f32')
plan.set_object(np.broadcast_to(obj, (B, 512, 512)), 5e10 * 16)
t = {_lib.RNG_PHILOX: [], _lib.RNG_NONE: []}
for r in range(6):
    for rng in (_lib.RNG_PHILOX, _lib.RNG_NONE):
        plan.bench_cycles(20, 1, rng=rng, seed=r)
        plan.ctx.synchronize()
        t0 = time.perf_counter()
        plan.bench_cycles(20, 3, rng=rng, seed=10 + r)
        plan.ctx.synchronize()
        t[rng].append((time.perf_counter() - t0) / 3)
a, b = np.median(t[_lib.RNG_PHILOX]), np.median(t[_lib.RNG_NONE])
print('cycle of %d frames: %.2f ms with the Philox / PTRS draw, %.2f ms with rng = none: the draw costs the cycle %.1f %%' % (B, a * 1e3, b * 1e3, (a - b) / a * 100))
