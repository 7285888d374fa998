"""Manual helper (not a test): f32 accuracy of library builds against the FIRST library's f64 plan, same measurement for all.

    python3 tools/gpu/ab_accuracy.py [--k 20] LIB_A LIB_B ...

Object: the astronaut at 512 x 512 (config 2's frame), point-descan 2.0x PSF; measurement = the f64 plan's Philox draw,
uploaded to every plan.  Prints normwise max|d| / max|ref| and the pixelwise max |d| / ref over pixels above 1e-3 of the maximum.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_bench import bind  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=20)
ap.add_argument('libs', nargs='+')
a = ap.parse_args()
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
psfs = [g['2p0x_lr/point_sted_psf'][0]]
rng = np.random.default_rng(5)
frames = [np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0], rng.random((512, 512)) * 255,
          np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0][::-1].copy(), rng.random((512, 512)) * 255]
B = len(frames)
m0 = bind(a.libs[0], 'ref')
ref = m0.DeconvPlan(psfs, B, 512, 512, dtype='f64')
ref.set_object(np.stack(frames), 5e10 * 16)
ref.simulate(seed=11)
meas = ref.measurement()
ref.iterate(a.k)
e_ref = ref.estimate()
for i, path in enumerate(a.libs):
    m = bind(path, 'acc%d' % i)
    plan = m.DeconvPlan(psfs, B, 512, 512, dtype='f32')
    plan.set_measurement(meas)
    plan.iterate(a.k)
    e = plan.estimate()
    out = []
    for f in range(B):
        d = np.abs(e[f] - e_ref[f])
        big = e_ref[f] > 1e-3 * e_ref[f].max()
        out.append('%.2e / %.2e' % (d.max() / e_ref[f].max(), (d[big] / e_ref[f][big]).max()))
    print('%-28s K=%d  normwise / pixelwise per frame:  %s   %s' % (os.path.basename(path), a.k, '   '.join(out), plan.strategy()), flush=True)
