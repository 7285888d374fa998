"""Manual helper (not a test): slice size (RLSTED_CHUNK_MB) x lanes (RLSTED_LANES) sweep of the headline workload in one process.
    python3 tools/gpu/sweep_slices.py [size] [views] [batch] [chunk MB list, comma separated]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psfs = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
obj = np.random.default_rng(1234).random((n, n)) * 255
chunks = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [27, 54, 80, 108, 160, 216]
combos = [(l, c) for l in (1, 2, 3, 4) for c in chunks]
res = {}
for rnd in range(2):
    for lanes, chunk in combos:
        os.environ['RLSTED_LANES'] = str(lanes)
        os.environ['RLSTED_CHUNK_MB'] = str(chunk)
        plan = _lib.DeconvPlan(psfs, B, n, n, dtype='f32')
        plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
        plan.bench_cycles(20, 1, seed=1)
        plan.ctx.synchronize()
        t0 = time.perf_counter()
        plan.bench_cycles(20, 2, seed=2)
        plan.ctx.synchronize()
        el = (time.perf_counter() - t0) / 2
        res.setdefault((lanes, chunk), []).append(B / el)
        del plan
for (lanes, chunk), v in sorted(res.items()):
    print('lanes %d chunk %3d MB: %s frames/s' % (lanes, chunk, ' '.join('%8.0f' % x for x in v)), flush=True)
