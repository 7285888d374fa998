"""Manual helper (not a test): line-rescan throughput (V views) against the slice budget."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib, psf
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
sets, _ = psf.figure_2_psfs(['2p5x_lr', '3p0x_lr'])
for name, psfs in sets.items():
    if 'point' in name:
        continue
    psfs = [np.asarray(p) for p in psfs]
    for lanes in ('1', '2'):
        for mb in ('108', '216', '432', '864', '100000'):
            os.environ['RLSTED_LANES'] = lanes
            os.environ['RLSTED_CHUNK_MB'] = mb
            plan = _lib.DeconvPlan(psfs, 64, 512, 512, dtype='f32')
            plan.set_object(np.broadcast_to(obj, (64, 512, 512)), 8e11)
            plan.bench_cycles(20, 1, seed=1)
            t0 = time.perf_counter(); plan.bench_cycles(20, 3, seed=2); el = time.perf_counter() - t0
            print('%s lanes=%s chunk=%6s MB: %7.0f frames/s' % (name, lanes, mb, 3 * 64 / el), flush=True)
            del plan
