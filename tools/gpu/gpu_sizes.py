"""Manual helper (not a test): whole-cycle throughput (simulate + K RL iterations) at several sizes / view counts under
several RLSTED_* environments.

    python tools/gpu/gpu_sizes.py CASE [CASE ...] -- ENV [ENV ...]
    CASE = size:views:batch[:K]      e.g. 2048:4:32   4096:1:8:100
    ENV  = comma separated RLSTED_ settings without the prefix, e.g. PAIR=0,LANES=1 ('-' = none)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
args = sys.argv[1:]
cases = args[:args.index('--')] if '--' in args else args
envs = args[args.index('--') + 1:] if '--' in args else ['-']
for case in cases:
    f = [int(x) for x in case.split(':')]
    n, V, B = f[:3]
    K = f[3] if len(f) > 3 else 20
    psfs = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
    obj = np.random.default_rng(1234).random((n, n)) * 255
    for spec in envs:
        for k in [k for k in os.environ if k.startswith('RLSTED_') and k != 'RLSTED_LIB']:
            del os.environ[k]
        for kv in filter(None, spec.replace('-', '').split(',')):
            k, v = kv.split('=')
            os.environ['RLSTED_' + k] = v
        plan = _lib.DeconvPlan(psfs, B, n, n, dtype='f32')
        plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
        plan.bench_cycles(K, 1, seed=1)
        reps = max(2, int(2e9 / (B * n * n * (3 * V + 4) * K)))
        plan.ctx.synchronize()
        t0 = time.perf_counter()
        plan.bench_cycles(K, reps, seed=2)
        plan.ctx.synchronize()
        el = time.perf_counter() - t0
        alg = 4 * n * n * ((2 * V + 2) + K * (3 * V + 4))
        fps = reps * B / el
        kt, fpl = plan.time_cycle(K, seed=3)
        print('%5d V=%d B=%d K=%d %-24s %8.1f frames/s  %5.1f%% of the algorithmic roofline  pairs=%d frames/launch %d  us: %s'
              % (n, V, B, K, spec, fps, alg * fps / 8e12 * 100, plan.strategy()['frame_pairs'], fpl,
                 {k: round(v[0] * 1e3) for k, v in kt.items() if k[:3] in ('col', 'row')}), flush=True)
        del plan
