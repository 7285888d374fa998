"""Manual helper (not a test): whole-cycle throughput of float64 plans (the reference's arithmetic) at the long transform lengths.
End of round 3: 1024^2 1376, 2048^2 311 (4 views: 90), 4096^2 53 frames/s at K = 20 (1302 / 225 / 72 / 35 before their column tiles went from
2 / 1 to 3 / 2 columns and their twiddles to the compact form) -- ~3x below the f32 plans (2x is the byte ratio):
the f64 column kernels of L = 1152 ... 4608 are still the workgroup-synchronous ones (4 x 10 complex doubles per lane do not fit the
outer-decimation body's register budget).

    python tools/gpu/gpu_f64_sizes.py
"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from rescan_line_sted_amd import _lib
g = np.load('tests/golden/g8_fig2_psfs.npz')
for n, V, B in ((2048, 1, 8), (2048, 4, 4), (1024, 1, 16), (4096, 1, 2)):
    psfs = [g['2p0x_lr/point_sted_psf'][0]] if V == 1 else [p[None] for p in g['2p0x_lr/line_sted_psfs'][:V, 0]]
    obj = np.random.default_rng(1).random((n, n)) * 255
    plan = _lib.DeconvPlan(psfs, B, n, n, dtype='f64')
    plan.set_object(np.broadcast_to(obj, (B, n, n)), 5e10 * (n / 128) ** 2)
    plan.bench_cycles(20, 1, seed=1)
    plan.ctx.synchronize()
    t0 = time.perf_counter()
    plan.bench_cycles(20, 2, seed=2)
    plan.ctx.synchronize()
    el = time.perf_counter() - t0
    print('f64 %d^2 V=%d: %.1f frames/s' % (n, V, 2 * B / el), flush=True)
    del plan
