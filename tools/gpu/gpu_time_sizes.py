"""Manual helper: each RL kernel timed ALONE (rl_deconv_time_kernels: back to back on one slice, nothing else in
flight), in microseconds per 512^2-equivalent frame (view image for the per-view kernels), for point (V=1) and
line-rescan (V=4) plans at 512 / 1024 / 2048.  -> gpurun_out/r02/kernels_alone.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
out = []
cases = [(2048, 16, 1), (2048, 8, 4), (1024, 64, 1), (512, 256, 1), (512, 64, 4)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for n, B, V in cases:
    psf = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
    obj = np.random.default_rng(1).random((n, n)) * 255
    plan = _lib.DeconvPlan(psf, B, n, n, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
    plan.simulate(seed=1)
    kt = plan.time_kernels(10)
    fl = kt['frames_per_rl_launch']
    row = {'n': n, 'batch': B, 'views': V, 'frames_per_launch': fl,
           'ms_per_launch': {k: kt[k] for k in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')},
           'us_per_512eq_frame': {k: round(kt[k] * 1e3 / fl / (n * n / 262144), 2)
                                  for k in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')}}
    print(row, flush=True)
    out.append(row)
    del plan
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r02'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r02', 'kernels_alone.json'), 'w'), indent=1)
