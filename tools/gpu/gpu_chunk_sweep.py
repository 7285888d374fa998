"""Manual helper (not a test): headline throughput against the RL chunk budget (frames whose
working set is iterated K times before moving on) and the kernel flavour (streaming / tiled)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness = bench.workload()
B = 256
for stream in sys.argv[1].split(',') if len(sys.argv) > 1 else ('1', '0'):
    for mb in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('36', '72', '144', '216', '288', '432', '100000')):
        os.environ['RLSTED_STREAM'] = stream
        os.environ['RLSTED_CHUNK_MB'] = mb
        plan = _lib.DeconvPlan(psf, B, 512, 512, dtype='f32')
        plan.set_object(np.broadcast_to(obj, (B, 512, 512)), brightness)
        plan.bench_cycles(20, 1, seed=1)
        t0 = time.perf_counter(); plan.bench_cycles(20, 4, seed=2); el = time.perf_counter() - t0
        print('stream=%s chunk=%6s MB: %8.0f frames/s' % (stream, mb, 4 * B / el), flush=True)
        del plan
