"""Manual helper (not a test): per-kernel timings of the bench workload, tolerant of
ablation builds whose results are garbage."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness, _ = bench.workload(512)
if len(sys.argv) > 3:      # e.g. 2p0x_lr/line_sted_psfs
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
    psf = [p[None] for p in g[sys.argv[3]][:, 0]]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
plan = _lib.DeconvPlan(psf, B, 512, 512, dtype=sys.argv[2] if len(sys.argv) > 2 else 'f32')
plan.set_object(np.broadcast_to(obj, (B, 512, 512)), brightness)
plan.simulate(seed=1)
print(json.dumps({k: round(v, 4) for k, v in plan.time_kernels(10).items()}))
