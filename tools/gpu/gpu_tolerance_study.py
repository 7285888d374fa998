"""Manual helper (not a test): BASELINE config 5's tolerance study, float32 half -- how the
float32 plan's Richardson-Lucy estimate drifts from the float64 plan's, iteration by iteration,
on the 4096x4096 tile (synthetic uniform object, point-descan PSF, K = 100).  Writes a small
JSON (iteration -> max|a-b|/max|b| and rms) next to the profiles.  The fp16-storage half of the
study is not built (spectra of ~1e7-count images exceed the fp16 range without per-frame scaling)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psf = [g['2p0x_lr/point_sted_psf'][0]]
obj = np.random.default_rng(4321).random((1, n, n)) * 255
a = _lib.DeconvPlan(psf, 1, n, n, dtype='f64')
b = _lib.DeconvPlan(psf, 1, n, n, dtype='f32')
a.set_object(obj, 5e10 * (n / 128) ** 2)
a.simulate(seed=5)
noisy = a.measurement()
b.set_object(obj, 5e10 * (n / 128) ** 2)
b.set_measurement(noisy)
a.set_measurement(noisy)
rows = []
marks = sorted(set([1, 2, 3, 5, 8, 10, 15, 20, 30, 40, 50, 60, 80, 100, K]))
done = 0
for k in [m for m in marks if m <= K]:
    a.iterate(k - done); b.iterate(k - done); done = k
    ea, eb = a.estimate()[0], b.estimate()[0]
    d = np.abs(ea - eb)
    rows.append({'iteration': k, 'max_over_max': float(d.max() / ea.max()), 'rms_over_rms': float(np.sqrt((d ** 2).mean()) / np.sqrt((ea ** 2).mean()))})
    print(rows[-1], flush=True)
out = {'shape': [n, n], 'psf': '2p0x_lr/point_sted_psf', 'object': 'default_rng(4321).random * 255', 'rows': rows}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'tolerance_study_f32_%d.json' % n), 'w'), indent=1)
