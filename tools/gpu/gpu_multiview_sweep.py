"""Manual helper (not a test): line-rescan throughput (V views, 512 x 512, 64 frames) against the slice budget, the
number of slice streams and the view-fusion mode of the column kernels.  usage: gpu_multiview_sweep.py [dose ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib, psf  # noqa: E402

objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
doses = sys.argv[1:] or ['2p0x_lr']
sets, _ = psf.figure_2_psfs(doses)
rows = []
for name, psfs in sets.items():
    if 'point' in name:
        continue
    psfs = [np.asarray(p) for p in psfs]
    for fuse in ('1', '0'):
        for lanes in ('1', '2'):
            for mb in ('108', '216', '432', '100000'):
                os.environ.update(RLSTED_LANES=lanes, RLSTED_CHUNK_MB=mb, RLSTED_FUSE_VIEWS=fuse)
                plan = _lib.DeconvPlan(psfs, 64, 512, 512, dtype='f32')
                plan.set_object(np.broadcast_to(obj, (64, 512, 512)), 8e11)
                plan.bench_cycles(20, 1, seed=1)
                t0 = time.perf_counter()
                plan.bench_cycles(20, 3, seed=2)
                el = time.perf_counter() - t0
                rows.append({'psfs': name, 'views': len(psfs), 'fuse_views': int(fuse), 'lanes': int(lanes), 'chunk_mb': int(mb),
                             'frames_per_s': 3 * 64 / el})
                print(rows[-1], flush=True)
                del plan
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r02'), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'r02', 'multiview_sweep.json'), 'w'), indent=1)
