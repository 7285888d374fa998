import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
from oracle import line_sted_oracle as orc
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psf = list(g['2p0x_lr/point_sted_psf'])
rng = np.random.default_rng(0)
for n in (128, 512):
    x = rng.random((1, n, n))
    ref = orc.Deconvolver(psf).H(x)[0][0]
    for dtype in ('f32', 'f64'):
        plan = _lib.DeconvPlan(psf, 1, n, n, dtype=dtype)
        h = plan.forward(x)[0, 0]
        nrm = plan.normalization()
        print(n, dtype, 'H err', float(np.abs(h - ref).max() / ref.max()), 'norm range', float(nrm.min()), float(nrm.max()), flush=True)
        del plan
