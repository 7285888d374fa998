"""Manual helper (not a test): the fused Richardson-Lucy kernel against the four-launch iteration.

    python tools/gpu/gpu_fused.py [--batch 256] [--out gpurun_out/fused.json] [--configs W:WGS:ACQ,...]

For every configuration: bitwise comparison of the estimates after K = 20 iterations with the
four-launch path on the same noisy measurement, then frames/s of the whole simulate + deconvolve cycle.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--k', type=int, default=20)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--no-ref', action='store_true')
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'fused.json'))
ap.add_argument('--configs', default='32:2:0:1,32:2:0:2,16:2:0:2,64:2:0:2,32:2:0:3,16:2:0:1,8:2:0:2')
args = ap.parse_args()

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
n = args.size
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, n // 128, n // 128)))[0]
psf = [g['2p0x_lr/point_sted_psf'][0]]
B, K = args.batch, args.k
alg = 4 * n * n * (4 + K * 7)


def make(env):
    for k in list(os.environ):
        if k.startswith('RLSTED_FUSED'):
            del os.environ[k]
    os.environ.update(env)
    plan = _lib.DeconvPlan(psf, B, n, n, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n // 128) ** 2)
    return plan


def run(plan, label):
    res = {'label': label}
    plan.simulate(seed=7)
    plan.iterate(1)
    res['_est1'] = plan.estimate()
    plan.reset_estimate()
    t0 = time.perf_counter()
    plan.iterate(K)
    res['iterate_wall_ms'] = (time.perf_counter() - t0) * 1e3
    res['iterate_ms'] = plan.last_ms()['iterate_ms']
    est = plan.estimate()
    # second run of the same thing (warm): the number to read
    plan.reset_estimate()
    plan.iterate(K)
    res['iterate_ms_warm'] = plan.last_ms()['iterate_ms']
    est2 = plan.estimate()
    res['rerun_identical'] = bool(np.array_equal(est, est2))
    plan.bench_cycles(K, 1, seed=1)
    t0 = time.perf_counter()
    plan.bench_cycles(K, args.reps, seed=2)
    el = time.perf_counter() - t0
    res['frames_per_s'] = args.reps * B / el
    res['whole_path_frac'] = alg * res['frames_per_s'] / 8e12
    res['rl_iter_us_per_frame'] = res['iterate_ms_warm'] * 1e3 / (B * K)
    return res, est


out = {'batch': B, 'size': n, 'k': K, 'runs': []}
est_ref = est1_ref = None
if not args.no_ref:
    ref_plan = make({'RLSTED_FUSED': '0'})
    r, est_ref = run(ref_plan, 'four-launch')
    est1_ref = r.pop('_est1')
    out['runs'].append(r)
    print(json.dumps(r), flush=True)
    del ref_plan
for cfg in args.configs.split(','):
    w, wgs, acq, st, fl = (cfg.split(':') + ['0'])[:5]
    env = {'RLSTED_FUSED': '1', 'RLSTED_FUSED_W': w, 'RLSTED_FUSED_WGS': wgs, 'RLSTED_FUSED_ACQ': acq, 'RLSTED_FUSED_S': st,
           'RLSTED_FUSED_FLAGS': fl}
    try:
        plan = make(env)
        r, est = run(plan, 'fused W=%s wgs/cu=%s acq=%s streams=%s flags=%s' % (w, wgs, acq, st, fl))
        est1 = r.pop('_est1')
        if est_ref is None:
            est_ref, est1_ref = est, est1
        r['k1_max_rel_diff'] = float(np.abs(est1 - est1_ref).max() / est1_ref.max())
        r['bitwise_equal_to_four_launch'] = bool(np.array_equal(est, est_ref))
        if not r['bitwise_equal_to_four_launch']:
            d = np.abs(est - est_ref)
            r['max_abs_diff'] = float(d.max())
            r['max_rel_diff'] = float(d.max() / est_ref.max())
            r['frames_differing'] = int((d.reshape(B, -1).max(axis=1) > 0).sum())
        del plan
    except Exception as exc:      # report and go on to the next configuration
        r = {'label': cfg, 'error': repr(exc)}
    out['runs'].append(r)
    print(json.dumps(r), flush=True)
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
with open(args.out, 'w') as f:
    json.dump(out, f, indent=1)
