import os, sys, json
import numpy as np
sys.path.insert(0, '/root/repo')
from rescan_line_sted_amd import _lib, psf
objs = np.load('/root/repo/tests/golden/objects.npz')
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
sets, _ = psf.figure_2_psfs(['3p0x_lr', '1p5x_lr'])
for name, psfs in sets.items():
    psfs = [np.asarray(p) for p in psfs]
    V = len(psfs)
    plan = _lib.DeconvPlan(psfs, 64, 512, 512, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (64, 512, 512)), 8e11)
    plan.simulate(seed=1)
    kt = plan.time_kernels(10)
    fl = kt['frames_per_rl_launch']
    print(name, 'V', V, 'frames/launch', fl, {k: round(v * 1e3 / fl, 2) for k, v in kt.items() if k in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')}, 'us per frame')
