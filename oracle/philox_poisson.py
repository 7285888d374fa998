"""CPU ORACLE -- test infrastructure only.

numpy twin of the device shot-noise generator (spec: DESIGN.md "Device
Poisson").  The reference draws noise with numpy.random.poisson
(line_sted_tools.py:508-511) from the process-global MT19937 stream, which has
no parallel form; the device uses a counter-based generator instead, and this
file is the independent statement of that generator against which the HIP
kernel must agree BIT FOR BIT (tests/test_gpu_parity.py).  Distributional
agreement with numpy's own Poisson sampler is tested separately.

Everything below is float64 numpy arithmetic (+ - * / sqrt floor only, no
fused multiply-add exists in numpy), vectorised over pixels.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_TAG = 0x504F4953


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All arguments uint64 arrays holding 32-bit
    values (broadcastable); returns four such arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0 = np.uint64(k0)
    k1 = np.uint64(k1)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def u53(a, b):
    a = (a >> np.uint64(5)).astype(np.float64)
    b = (b >> np.uint64(6)).astype(np.float64)
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)


def det_log(x):
    x = np.asarray(x, dtype=np.float64)
    bits = x.view(np.uint64)
    e = ((bits >> np.uint64(52)) & np.uint64(0x7ff)).astype(np.int64) - 1023
    m = ((bits & np.uint64(0x000fffffffffffff)) | np.uint64(0x3ff0000000000000)).view(np.float64)
    big = m > 1.4142135623730951
    m = np.where(big, m * 0.5, m)
    e = np.where(big, e + 1, e)
    s = (m - 1.0) / (m + 1.0)
    z = s * s
    p = np.full_like(z, 1.0 / 25.0)
    for d in (23.0, 21.0, 19.0, 17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        p = p * z + 1.0 / d
    p = p * z + 1.0
    return e.astype(np.float64) * 0.6931471805599453 + 2.0 * s * p


def det_exp(x):
    x = np.asarray(x, dtype=np.float64)
    n = np.floor(x * 1.4426950408889634 + 0.5)
    r = (x - n * 0.6931471803691238) - n * 1.9082149292705877e-10
    p = np.full_like(r, 1.0 / 6227020800.0)
    for f in (479001600.0, 39916800.0, 3628800.0, 362880.0, 40320.0, 5040.0,
              720.0, 120.0, 24.0, 6.0):
        p = p * r + 1.0 / f
    p = p * r + 0.5
    p = p * r + 1.0
    p = p * r + 1.0
    two_n = ((n.astype(np.int64) + 1023).astype(np.uint64) << np.uint64(52)).view(np.float64)
    return p * two_n


def det_logfact(k):
    k = np.asarray(k, dtype=np.float64)
    out = np.empty_like(k)
    small = k < 20.5
    if small.any():
        ks = k[small].astype(np.int64)
        f = np.ones(ks.shape, dtype=np.float64)
        for i in range(2, 21):
            f = np.where(ks >= i, f * float(i), f)
        out[small] = det_log(f)
    if (~small).any():
        x = k[~small] + 1.0
        xi = 1.0 / x
        x2 = xi * xi
        c = -1.0 / 1680.0
        c = c * x2 + 1.0 / 1260.0
        c = c * x2 - 1.0 / 360.0
        c = c * x2 + 1.0 / 12.0
        out[~small] = ((x - 0.5) * det_log(x) - x) + 0.9189385332046727 + c * xi
    return out


def poisson(lam, seed, image):
    """lam: (n_pix,) float64 rates of one image (pixel index = position in the
    flattened row-major image); image = frame * n_psf + view.  Returns float64
    integer-valued draws."""
    lam = np.asarray(lam, dtype=np.float64).ravel()
    n = lam.size
    pix = np.arange(n, dtype=np.uint64)
    k0 = int(seed) & 0xFFFFFFFF
    k1 = (int(seed) >> 32) & 0xFFFFFFFF
    out = np.zeros(n)
    done = ~(lam > 0.0)

    # ---- lam < 10: multiplication method --------------------------------
    sm = (~done) & (lam < 10.0)
    if sm.any():
        idx = np.nonzero(sm)[0]
        enlam = det_exp(-lam[idx])
        X = np.zeros(idx.size)
        prod = np.ones(idx.size)
        alive = np.ones(idx.size, dtype=bool)
        for blk in range(64):
            if not alive.any():
                break
            x0, x1, x2, x3 = philox4x32_10(pix[idx], image, blk, _TAG, k0, k1)
            for u in (u53(x0, x1), u53(x2, x3)):
                prod = np.where(alive, prod * u, prod)
                stop = alive & ~(prod > enlam)
                alive = alive & ~stop
                X = np.where(alive, X + 1.0, X)
        out[idx] = X
        done[idx] = True

    # ---- lam >= 10: PTRS --------------------------------------------------
    idx = np.nonzero(~done)[0]
    if idx.size:
        L = lam[idx]
        slam = np.sqrt(L)
        loglam = det_log(L)
        b = 0.931 + 2.53 * slam
        a = -0.059 + 0.02483 * b
        invalpha = 1.1239 + 1.1328 / (b - 3.4)
        vr = 0.9277 - 3.6224 / (b - 2.0)
        k = np.floor(L)
        alive = np.ones(idx.size, dtype=bool)
        with np.errstate(divide='ignore', invalid='ignore'):
            for blk in range(64):
                if not alive.any():
                    break
                x0, x1, x2, x3 = philox4x32_10(pix[idx], image, blk, _TAG, k0, k1)
                U = u53(x0, x1) - 0.5
                V = u53(x2, x3)
                us = 0.5 - np.abs(U)
                kk = np.floor((2.0 * a / us + b) * U + L + 0.43)
                k = np.where(alive, kk, k)
                acc = alive & (us >= 0.07) & (V <= vr)
                rej = alive & ~acc & ((kk < 0.0) | ((us < 0.013) & (V > us)))
                slow = alive & ~acc & ~rej
                acc = acc | (slow & ~(V > 0.0))
                slow = slow & (V > 0.0)
                if slow.any():
                    s = np.nonzero(slow)[0]
                    lhs = (det_log(V[s]) + det_log(invalpha[s])) - det_log(a[s] / (us[s] * us[s]) + b[s])
                    rhs = (kk[s] * loglam[s] - L[s]) - det_logfact(kk[s])
                    ok = lhs <= rhs
                    acc[s[ok]] = True
                alive = alive & ~acc
        k = np.where(alive & (k < 0.0), 0.0, k)
        out[idx] = k
    return out


def noisy_measurement(noiseless, seed):
    """noiseless: (n_images, ny, nx) -> Poisson(noiseless) + 1e-9, image by image."""
    noiseless = np.asarray(noiseless, dtype=np.float64)
    out = np.empty_like(noiseless)
    for i in range(noiseless.shape[0]):
        out[i] = poisson(noiseless[i], seed, i).reshape(noiseless[i].shape) + 1e-9
    return out
