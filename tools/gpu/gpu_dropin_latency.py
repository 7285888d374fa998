"""Manual helper (not a test): the reference's own usage pattern through the drop-in class -- one
Deconvolver, one 128x128 object, iterate() called 1025 times from Python with record_iteration at
the logarithmic save points (line_sted_figure_2.py:39-56) -- float64 plans."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import line_sted_tools as st
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
for name, psfs in (('point', list(g['2p0x_lr/point_sted_psf'])), ('line, 4 orientations', [p[None] for p in g['2p0x_lr/line_sted_psfs'][:, 0]])):
    with tempfile.TemporaryDirectory() as tmp:
        d = st.Deconvolver(psfs, output_prefix=os.path.join(tmp, 'x_'), verbose=False)
        d.create_data_from_object(objs['rings'].astype(np.float64), total_brightness=5e10, random_seed=0)
        d.iterate()                                   # plan creation, first launch
        t0 = time.perf_counter()
        for i, save in st.logarithmic_progress(range(1024), verbose=False):
            d.iterate()
            if save:
                d.record_iteration()
        el = time.perf_counter() - t0
        print('%-22s 1024 x iterate() + %d record_iteration(): %.3f s  (%.0f us per iteration)' % (name, len(d.saved_iterations), el, el / 1024 * 1e6), flush=True)
