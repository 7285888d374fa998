"""Manual helper (not a test): the four RL kernels ALONE (rl_deconv_time_kernels) as a function of the frames per launch -- how a
launch's time steps with the number of workgroup rounds it is (512^2 point: 72 column tiles / 64 row groups per frame pair).
    python3 tools/gpu/kernels_vs_batch.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['RLSTED_LANES'] = '1'
os.environ['RLSTED_CHUNK_MB'] = '100000'
from rescan_line_sted_amd import _lib  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psfs = [g['2p0x_lr/point_sted_psf'][0]]
obj = np.random.default_rng(1234).random((512, 512)) * 255
for B in (8, 16, 24, 28, 32, 40, 48, 56, 64, 96, 128):
    plan = _lib.DeconvPlan(psfs, B, 512, 512, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, 512, 512)), 5e10 * 16)
    plan.simulate(seed=1)
    plan.iterate(2)
    kt = plan.time_kernels(20)
    tot = kt['colconv_H'] + kt['rowpass_RATIO'] + kt['colconv_Ht'] + kt['rowpass_UPDATE']
    print('B %3d (%d frames per launch): colconv_H %.1f  RATIO %.1f  colconv_Ht %.1f  UPDATE %.1f us  -> %.2f us per frame-iteration (col WGs %d, row WGs %d)' % (
        B, kt['frames_per_rl_launch'], kt['colconv_H'] * 1e3, kt['rowpass_RATIO'] * 1e3, kt['colconv_Ht'] * 1e3, kt['rowpass_UPDATE'] * 1e3,
        tot * 1e3 / B, 72 * B // 2, 64 * B // 2), flush=True)
    del plan
