"""Manual helper: f32-vs-f64 error after 20 iterations, white-noise and astronaut frames at 512^2, with and without frame
pairs (RLSTED_PAIR) and the real PSF-spectrum multiplier."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness, _ = bench.workload(512)
rng = np.random.default_rng(1234)
objs = np.stack([obj, rng.random((512, 512)) * 255, rng.random((512, 512)) * 255, obj[::-1].copy()])
os.environ['RLSTED_PAIR'] = '0'
ref = _lib.DeconvPlan(psf, 4, 512, 512, dtype='f64')
ref.set_object(objs, brightness)
ref.simulate(seed=9)
noisy = ref.measurement()
ref.iterate(20)
r = ref.estimate()
for env in ({'RLSTED_PAIR': '0'}, {'RLSTED_PAIR': '1'}, {'RLSTED_PAIR': '1', 'RLSTED_REAL_PSF': '0'}, {'RLSTED_PAIR': '1', 'RLSTED_ONES_SHORTCUT': '0'},
            {'RLSTED_PAIR': '0', 'RLSTED_REAL_PSF': '0'}):
    for k in ('RLSTED_PAIR', 'RLSTED_REAL_PSF', 'RLSTED_ONES_SHORTCUT'):
        os.environ.pop(k, None)
    os.environ.update(env)
    p = _lib.DeconvPlan(psf, 4, 512, 512, dtype='f32')
    p.set_object(objs, brightness)
    p.set_measurement(noisy)
    p.iterate(20)
    e = p.estimate()
    print(env, ['%.2e' % (np.abs(e[f] - r[f]).max() / r[f].max()) for f in range(4)], flush=True)
