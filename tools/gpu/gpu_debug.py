"""Manual GPU triage helper (not a test): one subprocess per FFT length/dtype."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import sys, numpy as np
sys.path.insert(0, %r)
from rescan_line_sted_amd import _lib
n, dtype = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(0)
psf = [rng.random((1, 9, 9))]
plan = _lib.DeconvPlan(psf, 2, n, n, dtype=dtype)
print(plan.info())
x = rng.random((2, n, n))
y = plan.forward(x)
from oracle import line_sted_oracle as orc
ref = orc.Deconvolver(psf).H(x)[0]
print('H err', np.abs(y[:,0]-ref).max()/ref.max())
plan.set_object(x, 1e6); plan.simulate(seed=1); plan.iterate(2)
print('est finite', np.isfinite(plan.estimate()).all())
''' % ROOT
for n in (40, 150, 240, 512, 1100, 2048):
    for dtype in ('f32', 'f64'):
        env = dict(os.environ, RLSTED_DEBUG_SYNC='1')
        r = subprocess.run([sys.executable, '-c', code, str(n), dtype], env=env, capture_output=True, text=True, timeout=300)
        print('=== n=%d %s rc=%d' % (n, dtype, r.returncode)); print(r.stdout[-600:]); print(r.stderr[-500:])
