"""Manual helper: sparse emitters on a black background (tests/test_gpu_parity.py::test_dark_background_narrow_psf_stays_finite) through
two builds of the library -- how many estimate values are zero / not finite after K iterations of an f32 plan.
    python3 tools/gpu/dark_frame_ab.py LIB_A LIB_B
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_bench import bind  # noqa: E402

rng = np.random.default_rng(77)
ny = nx = 256
obj = np.zeros((2, ny, nx))
for b in range(2):
    obj[b, rng.integers(8, ny - 8, 30), rng.integers(8, nx - 8, 30)] = rng.random(30) + 0.5
yy, xx = np.mgrid[-4:5, -4:5]                         # an elliptical Gaussian at 30 degrees: narrow, and not rank 1 (the FFT path)
u, w = 0.866 * xx + 0.5 * yy, -0.5 * xx + 0.866 * yy
psf = [np.exp(-0.5 * ((u / 1.6) ** 2 + (w / 0.8) ** 2))[None]]
meas = None
for i, path in enumerate(sys.argv[1:]):
    m = bind(os.path.abspath(path), str(i))
    if meas is None:
        p64 = m.DeconvPlan(psf, 2, ny, nx, dtype='f64')
        p64.set_object(obj, 3e3)
        p64.simulate(seed=9)
        meas = p64.measurement()
        p64.iterate(12)
        ref = p64.estimate()
    p32 = m.DeconvPlan(psf, 2, ny, nx, dtype='f32')
    p32.set_measurement(meas)
    p32.iterate(12)
    e = p32.estimate()
    print('%-28s finite %s  zeros %.1f %%  max %.3g  vs float64 plan %.2e' % (
        os.path.basename(path), bool(np.isfinite(e).all()), 100 * (e == 0).mean(), e.max(), np.abs(e - ref).max() / ref.max()))
