import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
rng = np.random.default_rng(0)
for (ny, nx) in ((40, 150), (150, 40)):
    psf = [np.ones((1, 1, 1))]
    plan = _lib.DeconvPlan(psf, 1, ny, nx, dtype='f64')
    print(ny, nx, plan.info())
    x = rng.random((1, ny, nx))
    y = plan.forward(x)[0, 0]
    err = np.abs(y - x[0])
    bad = ~(err < 1e-9)
    print('bad count', bad.sum(), 'of', bad.size)
    if bad.any():
        rows = np.nonzero(bad.any(axis=1))[0]; cols = np.nonzero(bad.any(axis=0))[0]
        print('bad rows', rows[:40], len(rows)); print('bad cols', cols[:60], len(cols))
        print('sample', y[rows[0], cols[:8]], x[0, rows[0], cols[:8]])
