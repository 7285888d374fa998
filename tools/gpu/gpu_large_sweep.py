"""Manual helper (not a test): 2048^2 throughput against the slice budget and lane count."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
point = [g['2p0x_lr/point_sted_psf'][0]]
line4 = [p[None] for p in g['2p0x_lr/line_sted_psfs'][:, 0]]
n, K = 2048, 20
obj = np.random.default_rng(1234).random((n, n)) * 255
for name, psfs, B in (('point', point, 32), ('line4', line4, 16)):
    for spec in sys.argv[1:]:
        for k in [k for k in os.environ if k.startswith('RLSTED_') and k != 'RLSTED_LIB']:
            del os.environ[k]
        for kv in filter(None, spec.split(',')):
            k, v = kv.split('=')
            os.environ['RLSTED_' + k] = v
        plan = _lib.DeconvPlan(psfs, B, n, n, dtype='f32')
        plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * 256)
        plan.bench_cycles(K, 1, seed=1)
        t0 = time.perf_counter(); plan.bench_cycles(K, 2, seed=2); el = time.perf_counter() - t0
        V = len(psfs)
        alg = 4 * n * n * ((2 * V + 2) + K * (3 * V + 4))
        fps = 2 * B / el
        kt, fpl = plan.time_cycle(K, seed=3)
        print('%-6s %-28s %7.1f frames/s %5.1f%%  frames/launch %d  %s' % (name, spec, fps, alg * fps / 8e12 * 100, fpl,
              {k: round(v[0] * 1e3) for k, v in kt.items() if k[:3] in ('col', 'row')}), flush=True)
        del plan
