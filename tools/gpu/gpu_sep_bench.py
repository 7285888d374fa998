"""Manual helper (not a test): frames/s of the separable strategy against the FFT strategy for small rank-1
PSFs (two views, 0 / 90 degrees), 512 x 512 frames, float32.  -> gpurun_out/r02/separable_vs_fft.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402


def gauss(n, s):
    x = np.arange(n) - (n - 1) / 2
    return np.exp(-x ** 2 / (2 * s ** 2))


def run(sep, taps, one=1, B=64, n=512, K=20, reps=3):
    os.environ['RLSTED_SEP'] = str(sep)
    os.environ['RLSTED_SEP_ONE'] = str(one)
    u, v = gauss(taps, taps / 12), gauss(taps, taps / 4)
    plan = _lib.DeconvPlan([np.outer(u, v)[None], np.outer(v, u)[None]], B, n, n, dtype='f32')
    obj = np.random.default_rng(0).random((B, n, n)) * 100
    plan.set_object(obj, 1e9)
    plan.simulate(seed=1)
    plan.iterate(K)
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        plan.iterate(K)
        plan.last_ms()      # synchronises
        best = min(best, time.perf_counter() - t)
    ms = plan.last_ms()['iterate_ms']
    return {'separable': plan.strategy()['separable'], 'frame_iterations_per_s': B * K / (ms / 1e3), 'iterate_ms': ms,
            'wall_s': best}


out = {'shape': [512, 512], 'views': 2, 'batch': 64, 'K': 20, 'dtype': 'f32', 'rows': []}
for taps in (5, 9, 17, 25, 33, 49, 65, 107):
    row = {'taps_per_side': taps, 'separable': run(2, taps), 'separable_two_pass': run(2, taps, one=0), 'fft': run(0, taps)}
    row['ratio'] = row['separable']['frame_iterations_per_s'] / row['fft']['frame_iterations_per_s']
    print(row, flush=True)
    out['rows'].append(row)
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r02'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r02', 'separable_vs_fft_th%s.json' % os.environ.get('RLSTED_SEP_TH', '32')), 'w'), indent=1)
