"""Manual helper: f32-vs-f64 plan error after K iterations at several sizes (+ frames/s at 512^2)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
psf = list(g['2p0x_lr/point_sted_psf'])
out = {'tag': os.environ.get('TAG', ''), 'rows': []}
for size in (512, 1024, 2048):
    obj = np.random.default_rng(1234).random((1, size, size)) * 255
    p64 = _lib.DeconvPlan(psf, 1, size, size, dtype='f64'); p64.set_object(obj, 5e10 * (size // 128) ** 2); p64.simulate(seed=9)
    noisy = p64.measurement()
    p32 = _lib.DeconvPlan(psf, 1, size, size, dtype='f32'); p32.set_object(obj, 5e10 * (size // 128) ** 2); p32.set_measurement(noisy)
    errs = {}
    done = 0
    for k in (1, 5, 20, 100):
        p64.iterate(k - done); p32.iterate(k - done); done = k
        a, b = p32.estimate()[0], p64.estimate()[0]
        errs[k] = float(np.abs(a - b).max() / b.max())
    out['rows'].append({'size': size, 'err': errs})
    print(size, errs, flush=True)
    del p64, p32
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
plan = _lib.DeconvPlan(psf, 256, 512, 512, dtype='f32'); plan.set_object(np.broadcast_to(obj, (256, 512, 512)), 8e11)
plan.bench_cycles(20, 1, seed=1)
t0 = time.perf_counter(); plan.bench_cycles(20, 10, seed=2); el = time.perf_counter() - t0
out['frames_per_s'] = 2560 / el
print(out['frames_per_s'])
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r04'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r04', 'f32err_%s.json' % out['tag']), 'w'))
