import os, sys, json
import numpy as np
sys.path.insert(0, '/root/repo')
from rescan_line_sted_amd import _lib
g = np.load('/root/repo/tests/golden/g8_fig2_psfs.npz')
for n, B in ((2048, 16), (1024, 64), (512, 256)):
    psf = [g['2p0x_lr/point_sted_psf'][0]]
    obj = np.random.default_rng(1).random((n, n)) * 255
    plan = _lib.DeconvPlan(psf, B, n, n, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
    plan.simulate(seed=1)
    kt = plan.time_kernels(10)
    fl = kt['frames_per_rl_launch']
    print(n, 'frames/launch', fl, {k: round(v * 1e3 / fl / (n * n / 262144), 2) for k, v in kt.items() if k in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')}, 'us per 512^2-equivalent frame')
