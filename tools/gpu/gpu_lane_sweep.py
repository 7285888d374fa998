"""Manual helper (not a test): headline throughput against the number of concurrent slice
streams (RLSTED_LANES) and the slice budget."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness = bench.workload()
B = 256
if os.environ.get('PSFSET'):          # e.g. PSFSET=2p0x_lr/line_sted_psfs B=64
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
    psf = [p[None] for p in g[os.environ['PSFSET']][:, 0]]
    B = int(os.environ.get('B', '64'))
for lanes in (sys.argv[1].split(',') if len(sys.argv) > 1 else ('1', '2', '3', '4')):
    for mb in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('72', '108', '144', '216', '288')):
        os.environ['RLSTED_LANES'] = lanes
        os.environ['RLSTED_CHUNK_MB'] = mb
        plan = _lib.DeconvPlan(psf, B, 512, 512, dtype='f32')
        plan.set_object(np.broadcast_to(obj, (B, 512, 512)), brightness)
        plan.bench_cycles(20, 1, seed=1)
        t0 = time.perf_counter(); plan.bench_cycles(20, 4, seed=2); el = time.perf_counter() - t0
        print('lanes=%s chunk=%6s MB: %8.0f frames/s' % (lanes, mb, 4 * B / el), flush=True)
        del plan
