"""VERDICT r03 item 6, measured on the CPU (scipy.fft computes float32 transforms in float32): Richardson-Lucy with H(estimate)
evaluated (a) directly every iteration and (b) in the proposed increment form  e_k = e_{k-1} + conv(est_k - est_{k-1}, p),  both
in float32, against the float64 iteration on the same measurement.  Config 2's frame: the astronaut at 512 x 512, the 2.0x
point-descan PSF, 1e7-count shot noise, 20 iterations.

    python3 tools/increment_form_study.py

The loop starts from estimate = 1 (ref:522), so the FIRST increment is the whole image: its full-scale transform error enters e
once and is never refreshed, and every later increment adds its own -- the accumulated e_k is never more exact than a fresh
H(estimate_k).  (`ratio - 1` works because the residual is small at EVERY iteration; nothing is accumulated there.)
"""
import os
import sys

import numpy as np
import scipy.fft as sf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
psf = g['2p0x_lr/point_sted_psf'][0][0]
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
obj *= 5e10 * 16 / obj.sum()
n, P, L = 512, psf.shape[0], 640
c = (P - 1) // 2


def conv(x, dt):
    X = sf.rfft2(x.astype(dt), (L, L))
    K = sf.rfft2(psf.astype(dt), (L, L))
    return sf.irfft2(X * K, (L, L))[c:c + n, c:c + n]


rng = np.random.default_rng(0)
meas = rng.poisson(np.maximum(conv(obj, np.float64), 0)) + 1e-9
norm = conv(np.ones((n, n)), np.float64)


def rl(dt, increment, K=20):
    m, nr = meas.astype(dt), norm.astype(dt)
    est = np.ones((n, n), dtype=dt)
    e = conv(est, dt).astype(dt)
    for k in range(K):
        ratio1 = (m - np.maximum(e, 0)) / np.maximum(e, 0)                 # `ratio - 1`, as the f32 plans do
        new = (est * np.maximum(1 + conv(ratio1, dt).astype(dt) / nr, 0)).astype(dt)
        e = (e + conv(new - est, dt).astype(dt)).astype(dt) if increment else conv(new, dt).astype(dt)
        est = new
    return est.astype(np.float64)


ref = rl(np.float64, False)
big = ref > 1e-3 * ref.max()
for name, inc in (('direct H(estimate) every iteration', False), ('increment form e += conv(delta)', True)):
    a = rl(np.float32, inc)
    d = np.abs(a - ref)
    print('%-40s normwise %.2e   pixelwise (pixels > 1e-3 max) %.2e' % (name, d.max() / ref.max(), (d[big] / ref[big]).max()))
