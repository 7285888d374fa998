"""Manual helper (not a test): determinism soak -- the pipelined two-stream cycle repeated many
times with the same seed must leave the same bits every time (no cross-slice interference)."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness, _ = bench.workload(512)
B = 256
plan = _lib.DeconvPlan(psf, B, 512, 512, dtype='f32')
plan.set_object(np.broadcast_to(obj, (B, 512, 512)), brightness)
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    plan.bench_cycles(20, 3, seed=5)      # three cycles back to back (lanes stay open); the last one (seed 7) stays
    h = hashlib.sha256(plan.estimate().tobytes()).hexdigest()
    if ref is None:
        ref = h
    assert h == ref, 'repetition %d differs' % rep
print('soak ok: %d identical cycles, sha256 %s' % (rep + 1, ref[:16]))
