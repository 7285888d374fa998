"""Manual helper (not a test): BASELINE config 4 on one GPU -- 4 test objects x 6 doses x 3 scan
modes x 16 seeds = 1152 frames, simulate + 20 RL iterations each, through sweep.run_tasks.  The
18 PSF sets are stand-ins built from the three golden operating points (same shapes and view
counts as the figure's: point 1 view, line-descanned 1-2, line-rescanned 3-4)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import sweep
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
o = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
objects = {n: o[n][0].astype(np.float64) for n in ('astronaut', 'cat', 'lines', 'rings')}
base = {'point': [g['2p0x_lr/point_sted_psf'][0]], 'ld': [p[None] for p in g['1p0x_ld/line_sted_psfs'][:, 0]],
        'lr3': [p[None] for p in g['1p5x_lr/line_sted_psfs'][:, 0]], 'lr4': [p[None] for p in g['2p0x_lr/line_sted_psfs'][:, 0]]}
psf_sets = {}
for d in range(6):
    psf_sets['dose%d_point' % d] = base['point']
    psf_sets['dose%d_ld' % d] = base['ld']
    psf_sets['dose%d_lr' % d] = base['lr3' if d < 3 else 'lr4']
tasks = sweep.make_tasks(objects, psf_sets, range(16))
sweep.run_tasks(tasks[:8], objects, psf_sets, 20)          # warm up (library, plans)
t0 = time.perf_counter()
est = sweep.run_tasks(tasks, objects, psf_sets, 20)
el = time.perf_counter() - t0
print('%d tasks (%d plans of <= 256 frames) in %.2f s -> %.0f frames/s on one GPU, host I/O included'
      % (len(tasks), len({(t[1], objects[t[0]].shape) for t in tasks}), el, len(tasks) / el))
