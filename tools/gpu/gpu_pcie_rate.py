"""Manual helper (not a test): PCIe-inclusive rate of the batch API -- host float64 objects in, host float64
estimates out, per call -- with ordinary (pageable) numpy arrays and with page-locked ones (_lib.pinned_empty).
-> gpurun_out/r02/pcie_rate.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rescan_line_sted_amd import _lib  # noqa: E402

obj, psf, brightness, _ = bench.workload(512)
rows = []
for B in (16, 64, 256):
    plan = _lib.DeconvPlan(psf, B, 512, 512, dtype='f32')
    for kind in ('pageable, fresh result array', 'pageable, reused result array', 'pinned'):
        if kind == 'pinned':
            frames, out = _lib.pinned_empty((B, 512, 512)), _lib.pinned_empty((B, 512, 512))
        else:
            frames, out = np.empty((B, 512, 512)), (np.empty((B, 512, 512)) if 'reused' in kind else None)
        frames[:] = obj
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            plan.set_object(frames, brightness)       # host f64 -> HBM (converted on the device), H(obj)
            plan.simulate(seed=rep)
            plan.iterate(20)
            est = plan.estimate(out=out)              # HBM -> host f64
            best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter()
        plan.set_object(frames, brightness)
        plan.ctx.synchronize()
        up = time.perf_counter() - t0
        t0 = time.perf_counter()
        plan.estimate(out=out)
        down = time.perf_counter() - t0
        row = {'batch': B, 'host_arrays': kind, 'ms_per_call': best * 1e3, 'frames_per_s': B / best,
               'upload_GBps': B * 512 * 512 * 8 / up / 1e9, 'download_GBps': B * 512 * 512 * 8 / down / 1e9}
        print(row, flush=True)
        rows.append(row)
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r02'), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'r02', 'pcie_rate.json'), 'w'), indent=1)
