"""Manual helper: one seed of tests/test_gpu_parity.py::test_random_shapes_vs_oracle in detail (which stage deviates, where, under which switches).
    python3 tools/gpu/repro_seed.py SEED [ENV=VALUE ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
seed = int(sys.argv[1])
for kv in sys.argv[2:]:
    k, v = kv.split('=')
    os.environ[k] = v
from rescan_line_sted_amd import _lib as lib  # noqa: E402
from oracle import line_sted_oracle as orc  # noqa: E402

rng = np.random.default_rng(1000 + seed)
target = [64, 192, 256, 576, 1152, 192, 256, 576, 64, 192, 576, 256][seed % 12]
lo = {64: 2, 192: 70, 256: 200, 576: 260, 1152: 600}[target]
py, px = int(rng.integers(1, 40)), int(rng.integers(1, 40))
hy, hx = max((py - 1) // 2, py - 1 - (py - 1) // 2), max((px - 1) // 2, px - 1 - (px - 1) // 2)
ny = int(rng.integers(max(lo - hy, 1), target - hy + 1))
nx = int(rng.integers(max(lo - hx, 1), target - hx + 1))
if target == 1152:
    nx = int(rng.integers(2, 60))
V = int(rng.integers(1, 11)) if ny * nx < 40000 else int(rng.integers(1, 4))
B = int(rng.integers(1, 4))
psfs = [rng.random((1, py, px)) + 0.01 for _ in range(V)]
x = rng.random((B, ny, nx)) * 20
print('seed', seed, 'target', target, 'ny nx', ny, nx, 'psf', py, px, 'V', V, 'B', B)
d = orc.Deconvolver(psfs)
d.create_data_from_object(x, random_seed=seed)
meas = np.stack(d.noisy_measurement, axis=1)
print('measurement min/max', meas.min(), meas.max())
mr = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for dtype in ('f64', 'f32'):
    plan = lib.DeconvPlan(psfs, B, ny, nx, dtype=dtype)
    print(dtype, plan.info(), plan.strategy())
    plan.set_measurement(meas)
    d2 = orc.Deconvolver(psfs)
    d2.noisy_measurement = [m.copy() for m in d.noisy_measurement]
    d2.estimate = np.ones_like(x)
    for it in range(1, 4):
        plan.iterate(1)
        d2.iterate()
        e, r = plan.estimate(), d2.estimate
        k = np.unravel_index(np.argmax(np.abs(e - r)), e.shape)
        print('  iteration', it, 'normwise', mr(e, r), 'at', k, 'device', e[k], 'oracle', r[k], 'oracle min', r.min())
    Hx = plan.forward(d2.estimate)
    ref = d2.H(d2.estimate)
    print('  H(est) normwise', max(mr(Hx[:, v], ref[v]) for v in range(V)), 'min of oracle H(est)', min(float(r.min()) for r in ref))
