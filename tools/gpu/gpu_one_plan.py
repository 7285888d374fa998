"""Manual helper (profiling target): one plan, a few RL iterations.  usage: gpu_one_plan.py SIZE VIEWS BATCH K"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
n, V, B, K = (int(a) for a in sys.argv[1:5])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psfs = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
obj = np.random.default_rng(1234).random((n, n)) * 255
plan = _lib.DeconvPlan(psfs, B, n, n, dtype='f32')
plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
plan.simulate(seed=1)
plan.iterate(K)
print(plan.last_ms())
