import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psf = [g['2p0x_lr/point_sted_psf'][0]]
n = 512
obj = np.random.default_rng(4321).random((1, n, n)) * 255
plan = _lib.DeconvPlan(psf, 1, n, n, dtype='f32')
plan.set_object(obj, 8e11)
nl = plan.noiseless()
print('psf sum', psf[0].sum(), 'noiseless finite', np.isfinite(nl).all(), nl.max(), 'norm', np.isfinite(plan.normalization()).all())
plan.simulate(seed=1)
plan.iterate(1)
e = plan.estimate()
print('est finite', np.isfinite(e).all(), np.nanmax(e), 'nan count', int(np.isnan(e).sum()))
x = np.random.default_rng(0).random((1, n, n))
print('forward finite', np.isfinite(plan.forward(x)).all())
