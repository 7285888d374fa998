"""Manual helper: the four RL kernels timed ALONE (rl_deconv_time_kernels: back to back on one slice, nothing else in flight).
    python tools/gpu/gpu_kernels_alone.py size:views:batch [...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
for case in sys.argv[1:]:
    n, V, B = (int(x) for x in case.split(':'))
    psf = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
    obj = np.random.default_rng(1).random((n, n)) * 255
    plan = _lib.DeconvPlan(psf, B, n, n, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
    plan.simulate(seed=1)
    kt = plan.time_kernels(10)
    fl = kt['frames_per_rl_launch']
    print(case, 'pairs', plan.strategy()['frame_pairs'], 'frames/launch', fl, {k: round(kt[k] * 1e3) for k in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')}, flush=True)
    del plan
