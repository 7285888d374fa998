"""Manual helper for rocprofv3 runs: ONE short cycle (simulate + K RL iterations) of a plan, nothing else.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d gpurun_out/r04/pmc --output-format csv -- \
        python3 tools/gpu/prof_cycle.py SIZE VIEWS BATCH [K] [LANES] [DTYPE] [FL_FILE]

FL_FILE: the frames an RL launch covers (the slice) are written there: tools/pmc_traffic.py needs them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n, V, B = (int(x) for x in sys.argv[1:4])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
if len(sys.argv) > 5 and int(sys.argv[5]) > 0:
    os.environ['RLSTED_LANES'] = sys.argv[5]
dtype = sys.argv[6] if len(sys.argv) > 6 else 'f32'
from rescan_line_sted_amd import _lib  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psfs = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
obj = np.random.default_rng(1234).random((n, n)) * 255
plan = _lib.DeconvPlan(psfs, B, n, n, dtype=dtype)
plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
plan.bench_cycles(K, 1, seed=1)
plan.ctx.synchronize()
if len(sys.argv) > 7:
    from rescan_line_sted_amd._lib import lib, check
    import ctypes
    # frames per RL launch = the slice: the same rule the run above used (rl_deconv_time_kernels reports it; its own launches are
    # of the same shapes, a handful against the cycle's hundreds)
    fl = plan.time_kernels(1)['frames_per_rl_launch']
    open(sys.argv[7], 'w').write(str(fl))
print('done', plan.strategy())
