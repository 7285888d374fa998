"""Manual helper (not a test): A/B of library builds in ONE process, interleaved rounds (the guide's rule for perf deltas).

    python3 tools/gpu/ab_bench.py [--size 512] [--views 1] [--batch 1024] [--k 20] [--rounds 5] [--dtype f32] LIB_A LIB_B ...

Each library is bound through its own copy of rescan_line_sted_amd._lib.  Prints per library the median / min step time,
frames/s, and the estimate's largest deviation from the first library's (same seed, same objects).
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def bind(path, tag):
    """A private copy of rescan_line_sted_amd._lib bound to `path`.  An OLDER build may lack entry points the current binding
    declares: this tool (only) skips the ones it does not find, so that a kept library of an earlier round can be timed."""
    os.environ['RLSTED_LIB'] = os.path.abspath(path)
    import types
    src = open(os.path.join(ROOT, 'rescan_line_sted_amd', '_lib.py')).read()
    strict = "        fn = getattr(lib, name)          # AttributeError if the ABI drifted\n"
    assert strict in src
    src = src.replace(strict, "        fn = getattr(lib, name, None)\n        if fn is None:\n            continue\n")
    m = types.ModuleType('rescan_line_sted_amd._lib_' + tag)
    m.__package__ = 'rescan_line_sted_amd'
    m.__file__ = os.path.join(ROOT, 'rescan_line_sted_amd', '_lib.py')
    sys.modules[m.__name__] = m
    exec(compile(src, m.__file__, 'exec'), m.__dict__)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--views', type=int, default=1)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--k', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('libs', nargs='+')
    a = ap.parse_args()
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
    psfs = [g['2p0x_lr/point_sted_psf'][0]] if a.views == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:a.views, 0]]
    n = a.size
    obj = np.random.default_rng(1234).random((n, n)) * 255
    mods, plans = [], []
    for i, path in enumerate(a.libs):
        m = bind(path, str(i))
        plan = m.DeconvPlan(psfs, a.batch, n, n, dtype=a.dtype)
        plan.set_object(np.broadcast_to(obj, (a.batch, n, n)), 5e10 * (n / 128) ** 2)
        plan.bench_cycles(a.k, 1, seed=1)       # warm-up
        mods.append(m)
        plans.append(plan)
    times = [[] for _ in plans]
    for r in range(a.rounds):
        for i, plan in enumerate(plans):
            plan.ctx.synchronize()
            t0 = time.perf_counter()
            plan.bench_cycles(a.k, a.reps, seed=2 + r)
            plan.ctx.synchronize()
            times[i].append((time.perf_counter() - t0) / a.reps)
    ests = []
    for plan in plans:
        plan.bench_cycles(a.k, 1, seed=7)
        e = plan.estimate()
        ests.append(e[:min(4, a.batch)].copy())
    for i, path in enumerate(a.libs):
        t = np.array(times[i])
        d = float(np.max(np.abs(ests[i] - ests[0])) / np.max(np.abs(ests[0])))
        print('%-44s median %8.3f ms  min %8.3f ms  %9.0f frames/s (median)  max dev vs first %.2e  %s' % (
            os.path.basename(path), np.median(t) * 1e3, t.min() * 1e3, a.batch / np.median(t), d, plans[i].strategy()), flush=True)


if __name__ == '__main__':
    main()
