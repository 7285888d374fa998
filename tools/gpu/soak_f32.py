"""Manual soak (not a test): f32 plans against float64 plans over random workloads at the sizes people run -- 128 ... 1024 square, 1 / 2 / 4
views, the figure-2 PSFs or a narrow non-separable one, dense or sparse objects, doses from a few photons per frame to 1e12, K = 20 --
looking for frames that are not finite, all zero, or outside 1e-4 of the float64 plan's maximum.
    python3 tools/gpu/soak_f32.py N_CASES [FIRST_SEED]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib as lib  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
points = [k for k in g.files if k.endswith('point_sted_psf')]
lines = [k for k in g.files if k.endswith('line_sted_psfs')]
n_cases, first = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 0
worst, bad, t0 = [], 0, time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(900000 + seed)
    n = int(rng.choice([128, 256, 512, 1024], p=[0.4, 0.3, 0.2, 0.1]))
    V = int(rng.choice([1, 2, 4]))
    kind = int(rng.integers(0, 3))
    if kind == 0 and V == 1:
        psfs = [g[points[int(rng.integers(0, len(points)))]][0]]
    elif kind <= 1:
        stack = g[lines[int(rng.integers(0, len(lines)))]][:, 0]
        psfs = [stack[v % len(stack)][None] for v in range(V)]
    else:
        yy, xx = np.mgrid[-5:6, -5:6]
        psfs = []
        for v in range(V):
            a = rng.uniform(0, np.pi)
            u, w = np.cos(a) * xx + np.sin(a) * yy, -np.sin(a) * xx + np.cos(a) * yy
            psfs.append(np.exp(-0.5 * ((u / rng.uniform(1.2, 2.5)) ** 2 + (w / rng.uniform(0.6, 1.1)) ** 2))[None])
    B = int(rng.choice([1, 2, 3, 4]))
    obj = rng.random((B, n, n))
    flavour = int(rng.integers(0, 3))
    if flavour == 1:
        obj *= rng.random((B, n, n)) < 0.002                      # sparse emitters
        obj[:, n // 2, n // 2] = 1.0
    elif flavour == 2:
        obj[:, : n // 2] *= 1e-6                                   # half the field nearly dark
    brightness = float(10 ** rng.uniform(2, 12))
    p64 = lib.DeconvPlan(psfs, B, n, n, dtype='f64')
    p64.set_object(obj, brightness)
    p64.simulate(seed=seed)
    meas = p64.measurement()
    p64.iterate(20)
    ref = p64.estimate()
    del p64
    p32 = lib.DeconvPlan(psfs, B, n, n, dtype='f32')
    p32.set_measurement(meas)
    p32.iterate(20)
    e = p32.estimate()
    strat = p32.strategy()
    del p32
    ok64 = bool(np.isfinite(ref).all() and ref.max() > 0)
    errs = [float(np.abs(e[b] - ref[b]).max() / ref[b].max()) if ref[b].max() > 0 else float('nan') for b in range(B)]
    fin = bool(np.isfinite(e).all() and e.min() >= 0 and all(e[b].max() > 0 for b in range(B)))
    err = max(errs)
    desc = 'seed %d n %d V %d B %d psf %d flavour %d brightness %.1e pairs %s sep %s' % (seed, n, V, B, kind, flavour, brightness, strat['frame_pairs'], strat['separable'])
    if not ok64 or not fin or not err < 1e-4:
        bad += 1
        print('BAD  %s: f64 ok %s, f32 finite/positive %s, err %.2e' % (desc, ok64, fin, err), flush=True)
    worst.append((err, desc))
worst.sort(reverse=True)
print('%d cases in %.0f s, %d bad; worst errors:' % (n_cases, time.time() - t0, bad))
for err, desc in worst[:8]:
    print('   %.2e  %s' % (err, desc))
print('median error %.2e' % np.median([w[0] for w in worst]))
