"""Manual helper (not a test): A/B of run-time switches (RLSTED_* read at plan creation) in ONE process, interleaved rounds.
    python3 tools/gpu/env_ab.py SIZE VIEWS BATCH "A=1,B=2" "A=0" ...      (each argument: one configuration's environment)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402

n, V, B = (int(x) for x in sys.argv[1:4])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psfs = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
obj = np.random.default_rng(1234).random((n, n)) * 255
plans = []
for cfg in sys.argv[4:]:
    env = dict(kv.split('=') for kv in cfg.split(',') if kv)
    os.environ.update(env)
    plan = _lib.DeconvPlan(psfs, B, n, n, dtype=env.pop('DTYPE', 'f32'))   # (DTYPE=f64: a key of this tool, not of the library)
    for k in list(env) + ['DTYPE']:
        os.environ.pop(k, None)
    plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
    plan.bench_cycles(20, 1, seed=1)
    plans.append(plan)
times = [[] for _ in plans]
for r in range(4):
    for i, plan in enumerate(plans):
        plan.ctx.synchronize()
        t0 = time.perf_counter()
        plan.bench_cycles(20, 2, seed=2 + r)
        plan.ctx.synchronize()
        times[i].append((time.perf_counter() - t0) / 2)
for cfg, t in zip(sys.argv[4:], times):
    print('%-40s median %8.2f ms  %8.1f frames/s' % (cfg or '(default)', np.median(t) * 1e3, B / np.median(t)), flush=True)
