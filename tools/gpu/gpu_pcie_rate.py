"""Manual helper (not a test): PCIe-inclusive rate of the batch API -- host float64
objects in, host float64 estimates out, per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness = bench.workload()
for B in (16, 64, 256):
    plan = _lib.DeconvPlan(psf, B, 512, 512, dtype='f32')
    frames = np.ascontiguousarray(np.broadcast_to(obj, (B, 512, 512)))
    for rep in range(3):
        t0 = time.perf_counter()
        plan.set_object(frames, brightness)       # host f64 -> f32 -> HBM, H(obj)
        plan.simulate(seed=rep)
        plan.iterate(20)
        est = plan.estimate()                     # HBM -> host f64
        el = time.perf_counter() - t0
    print('B=%d: %.1f ms per call -> %.0f frames/s including host conversion + PCIe both ways' % (B, el * 1e3, B / el))
