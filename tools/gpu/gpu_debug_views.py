import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
rng = np.random.default_rng(0)
x = rng.random((1, 512, 512))
for key in ('1p5x_lr', '2p0x_lr'):
    psfs = [p[None] for p in g[key + '/line_sted_psfs'][:, 0]]
    out = {}
    for real in ('0', '1'):
        os.environ['RLSTED_REAL_PSF'] = real
        plan = _lib.DeconvPlan(psfs, 1, 512, 512, dtype='f32')
        out[real] = (plan.forward(x)[0], plan.adjoint(plan.forward(x), normalize=False)[0], plan.normalization())
        del plan
    for v in range(len(psfs)):
        print(key, 'H view', v, float(np.abs(out['0'][0][v] - out['1'][0][v]).max() / out['0'][0][v].max()))
    print(key, 'Ht', float(np.abs(out['0'][1] - out['1'][1]).max() / out['0'][1].max()), 'norm', float(np.abs(out['0'][2] - out['1'][2]).max()))
    # single views through V=1 plans
    for v in range(len(psfs)):
        o = {}
        for real in ('0', '1'):
            os.environ['RLSTED_REAL_PSF'] = real
            plan = _lib.DeconvPlan([psfs[v]], 1, 512, 512, dtype='f32')
            o[real] = plan.forward(x)[0, 0]
            del plan
        print(key, 'single view', v, float(np.abs(o['0'] - o['1']).max() / o['0'].max()))
