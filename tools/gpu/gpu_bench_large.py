"""Manual helper (not a test): throughput of the large-image BASELINE shapes (configs 3 and 5)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
point = [g['2p0x_lr/point_sted_psf'][0]]
line4 = [p[None] for p in g['2p0x_lr/line_sted_psfs'][:, 0]]
for name, psfs, n, B, K, obj_seed in (('2048^2 point V=1 K=20', point, 2048, 16, 20, 1234), ('2048^2 line-rescan V=4 K=20', line4, 2048, 8, 20, 1234),
                                      ('1024^2 point V=1 K=20', point, 1024, 64, 20, 99), ('4096^2 point V=1 K=100', point, 4096, 8, 100, 4321)):
    obj = np.random.default_rng(obj_seed).random((n, n)) * 255
    plan = _lib.DeconvPlan(psfs, B, n, n, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
    plan.bench_cycles(K, 1, seed=1)
    t0 = time.perf_counter(); plan.bench_cycles(K, 2, seed=2); el = time.perf_counter() - t0
    V = len(psfs)
    alg = 4 * n * n * ((2 * V + 2) + K * (3 * V + 4))
    fps = 2 * B / el
    info = plan.info()
    print('%-30s L=%d B=%3d  %8.1f frames/s   alg %.1f MB/frame -> %.1f%% of 8 TB/s' % (name, info['ly'], B, fps, alg / 1e6, alg * fps / 8e12 * 100), flush=True)
    del plan
