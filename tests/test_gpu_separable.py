"""The separable strategy (sep_kernels.hip): plans whose views are all rank 1 and small run H / H_t as row +
column stencils.  Checked against the FFT strategy of the same library (RLSTED_SEP=0) and against the oracle;
the strategy choice itself is checked through rl_deconv_strategy."""
import os

import numpy as np
import pytest

from conftest import max_rel, fuzz_seeds
from oracle import line_sted_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from rescan_line_sted_amd import _lib
    assert _lib.device_count() >= 1, 'no GPU visible'
    return _lib


def plan_with(lib, sep, *args, one_kernel=True, **kw):
    """A plan created under RLSTED_SEP=sep (0 FFT, 1 automatic, 2 separable whenever rank 1); one_kernel=False
    keeps the two-pass form of the stencils (RLSTED_SEP_ONE=0), True asks for the one-kernel form whenever its tile fits LDS (=2)."""
    knobs = {'RLSTED_SEP': str(sep), 'RLSTED_SEP_ONE': '2' if one_kernel else '0'}
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        return lib.DeconvPlan(*args, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def gauss(n, s, shift=0.0):
    x = np.arange(n) - (n - 1) / 2 - shift
    return np.exp(-x ** 2 / (2 * s ** 2))


def rank1_views(kind):
    if kind == 'row7':            # golden case G4's geometry: a 1 x 7 line
        return [np.array([[1, 2, 3, 4, 3, 2, 1.0]])]
    if kind == 'two_lines':       # 0 / 90 degree line PSFs: u v^T and its transpose
        u, v = gauss(17, 1.2), gauss(17, 4.0)
        return [np.outer(u, v), np.outer(v, u)]
    if kind == 'even_skew':       # even sizes, asymmetric taps: the centre convention (py-1)//2 matters
        return [np.outer(gauss(6, 1.5, 0.7), gauss(4, 1.0, -0.4)), np.outer(gauss(6, 0.8, -1.0), gauss(4, 2.0, 0.3))]
    raise KeyError(kind)


def planes(views):
    return [np.asarray(v)[None] for v in views]          # the reference's (1, py, px) PSF stacks


@pytest.mark.parametrize('one_kernel', [True, False])
@pytest.mark.parametrize('kind,shape', [('row7', (33, 70)), ('two_lines', (128, 128)), ('even_skew', (61, 300))])
def test_separable_matches_fft_strategy_and_oracle(lib, kind, shape, one_kernel):
    views = planes(rank1_views(kind))
    rng = np.random.default_rng(11)
    B, (ny, nx) = 3, shape
    obj = rng.random((B, ny, nx)) * 50
    sep = plan_with(lib, 2, views, B, ny, nx, dtype='f64', one_kernel=one_kernel)
    fft = plan_with(lib, 0, views, B, ny, nx, dtype='f64')
    assert sep.strategy()['separable'] and not fft.strategy()['separable']
    assert max_rel(sep.forward(obj), fft.forward(obj)) < 1e-12
    y = rng.random((B, len(views), ny, nx))
    for normalize in (True, False):
        assert max_rel(sep.adjoint(y, normalize), fft.adjoint(y, normalize)) < 1e-12
    for p in (sep, fft):
        p.set_object(obj, 1e7)
        p.simulate(seed=9)
    assert max_rel(sep.noiseless(), fft.noiseless()) < 1e-12
    assert np.array_equal(sep.measurement(), fft.measurement())      # ... and the same Poisson draws from it
    for p in (sep, fft):
        p.iterate(20)
    assert max_rel(sep.estimate(), fft.estimate()) < 1e-11
    for p in (sep, fft):                                             # continuing without a restart
        p.iterate(5)
    assert max_rel(sep.estimate(), fft.estimate()) < 1e-11
    o = orc.Deconvolver(views)
    o.create_data_from_object(obj[:1], noisy_measurement=[m[None] for m in sep.measurement()[0]])
    for _ in range(25):
        o.iterate()
    assert max_rel(sep.estimate()[0], o.estimate[0]) < 1e-10


def test_separable_f32_within_contract(lib):
    views = planes(rank1_views('two_lines'))
    rng = np.random.default_rng(12)
    obj = rng.random((2, 128, 128)) * 50
    p64 = plan_with(lib, 2, views, 2, 128, 128, dtype='f64')
    p32 = plan_with(lib, 2, views, 2, 128, 128, dtype='f32')
    assert p32.strategy()['separable']
    p64.set_object(obj, 1e7)
    p64.simulate(seed=2)
    p32.set_object(obj, 1e7)
    p32.set_measurement(p64.measurement())
    for p in (p64, p32):
        p.iterate(20)
    assert max_rel(p32.estimate(), p64.estimate()) < 1e-5       # BASELINE f32 contract


def test_strategy_choice(lib, golden):
    big = golden('g8_fig2_psfs')['2p0x_lr/line_sted_psfs'][::2, 0]        # 107 x 107, the 0 / 90 degree views: rank 1 but large
    assert not plan_with(lib, 1, planes(big), 1, 128, 128, dtype='f64').strategy()['separable']
    ref = plan_with(lib, 0, planes(big), 1, 128, 128, dtype='f64')
    x = np.random.default_rng(3).random((1, 128, 128))
    y = np.random.default_rng(4).random((1, 2, 128, 128))
    for one_kernel in (True, False):                                          # 107 taps: 120 KB of LDS in the one-kernel form
        forced = plan_with(lib, 2, planes(big), 1, 128, 128, dtype='f64', one_kernel=one_kernel)
        assert forced.strategy()['separable']
        assert max_rel(forced.forward(x), ref.forward(x)) < 1e-12
        assert max_rel(forced.adjoint(y), ref.adjoint(y)) < 1e-12
    rot = golden('g8_fig2_psfs')['2p0x_lr/line_sted_psfs'][:, 0]
    if len(rot) > 2:                                                          # a 45 degree view is not rank 1
        assert not plan_with(lib, 2, planes(rot), 1, 128, 128).strategy()['separable']
    small = planes([np.outer(gauss(7, 1.0), gauss(5, 2.0))])                 # automatic choice: rank 1 and py + px <= 16
    assert plan_with(lib, 1, small, 1, 64, 64).strategy()['separable']
    assert not plan_with(lib, 1, planes(rank1_views('two_lines')), 1, 64, 64).strategy()['separable']   # 17 + 17 taps: FFT is faster
    sp = plan_with(lib, 1, small, 1, 64, 64)
    sp.set_object(np.ones((1, 64, 64)), 1e6)
    sp.simulate(seed=1)
    with pytest.raises(lib.RlstedError):                                      # per-kernel timing is the FFT strategy's
        sp.time_kernels(2)
    blob = np.outer(gauss(9, 2), gauss(9, 2)) + np.eye(9) * 0.01             # small, full rank
    assert not plan_with(lib, 2, planes([blob]), 1, 64, 64).strategy()['separable']


@pytest.mark.parametrize('seed', fuzz_seeds(12))
def test_separable_random_small_cases_vs_oracle(lib, seed):
    """Automatic choice (py + px <= 16) on ragged shapes: images smaller than a tile, smaller than the PSF,
    1-pixel rows / columns, several views and frames -- H, H_t and three iterations against the oracle."""
    rng = np.random.default_rng(100 + seed)
    ny, nx = int(rng.integers(1, 71)), int(rng.integers(1, 140))
    py, px = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    V, B = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    views = planes([np.outer(rng.random(py) + 0.1, rng.random(px) + 0.1) for _ in range(V)])
    plan = lib.DeconvPlan(views, B, ny, nx, dtype='f64')
    assert plan.strategy()['separable']
    x = rng.random((B, ny, nx)) * 30
    o = orc.Deconvolver(views)
    h = plan.forward(x)
    for f in range(B):
        ref = o.H(x[f][None])
        errs = [max_rel(h[f, v], ref[v][0]) for v in range(V)]
        assert max(errs) < 1e-12, (ny, nx, py, px, V, B, errs)
    y = rng.random((B, V, ny, nx)) + 0.5
    ht = plan.adjoint(y, True)
    for f in range(B):
        assert max_rel(ht[f], o.H_t([y[f, v][None] for v in range(V)])[0]) < 1e-12
    plan.set_object(x)
    plan.set_measurement(y)
    plan.iterate(3)
    for f in range(B):
        d = orc.Deconvolver(views)
        d.create_data_from_object(x[f][None], noisy_measurement=[y[f, v][None] for v in range(V)])
        for _ in range(3):
            d.iterate()
        assert max_rel(plan.estimate()[f], d.estimate[0]) < 1e-11, (ny, nx, py, px, V, B)


@pytest.mark.parametrize('dtype,py,expect_sep', [('f64', 107, True), ('f32', 230, True), ('f64', 300, False)])
def test_long_column_stencils_raise_their_lds_limit_or_fall_back(lib, dtype, py, expect_sep):
    """the two-pass column stencil stages (32 + py - 1) x 64 values -- above the 64 KB default dynamic-LDS
    limit from py = 98 (f64) / 226 (f32) on, e.g. the 107-tap figure-2 line PSFs under RLSTED_SEP=2.  The kernel's limit is
    raised; where the tile would not fit the 160 KB of a compute unit the plan keeps the FFT strategy."""
    u, v = gauss(py, py / 8.0), gauss(9, 1.5)
    psf = [np.outer(u, v)[None] / np.outer(u, v).sum()]
    rng = np.random.default_rng(py)
    x = rng.random((2, 90, 70))
    sep = plan_with(lib, 2, psf, 2, 90, 70, one_kernel=False, dtype=dtype)
    assert sep.strategy()['separable'] == expect_sep
    fft = plan_with(lib, 0, psf, 2, 90, 70, dtype=dtype)
    assert not fft.strategy()['separable']
    assert max_rel(sep.forward(x), fft.forward(x)) < (1e-12 if dtype == 'f64' else 2e-6)
    d = orc.Deconvolver(psf)
    assert max_rel(sep.forward(x)[0, 0], d.H(x[:1])[0][0]) < (1e-12 if dtype == 'f64' else 2e-6)


# ------------------------------------------------------------------ the direct 2-D stencil (PSFs that are not rank 1)
def _direct_plan(lib, mode, psfs, B, ny, nx, dtype, monkeypatch):
    monkeypatch.setenv('RLSTED_DIRECT', str(mode))
    try:
        return lib.DeconvPlan(psfs, B, ny, nx, dtype=dtype)
    finally:
        monkeypatch.delenv('RLSTED_DIRECT')


@pytest.mark.parametrize('py,px,V,shape', [(3, 5, 2, (40, 70)), (7, 7, 1, (128, 128)), (11, 9, 3, (61, 300)), (2, 6, 2, (33, 33)),
                                           (15, 13, 1, (200, 90)), (8, 4, 4, (17, 250))])
def test_direct_stencil_matches_fft_strategy_and_oracle(lib, py, px, V, shape, monkeypatch):
    """PSFs that are small but not rank 1 run as a direct 2-D stencil (sep_kernels.hip k_sep2d DIRECT; the default up to 49 taps,
    RLSTED_DIRECT=2 beyond): H, H_t with and without the normaliser, the noiseless image, the Poisson draws from it and 25 RL
    iterations against the FFT strategy of the same library and against the oracle -- odd / even / two-row taps, 1-4 views, tiles
    that do not fill."""
    rng = np.random.default_rng(py * 100 + px)
    psfs = [rng.random((1, py, px)) + 0.05 for _ in range(V)]
    B, (ny, nx) = 2, shape
    obj = rng.random((B, ny, nx)) * 50
    d = _direct_plan(lib, 2, psfs, B, ny, nx, 'f64', monkeypatch)
    fft = _direct_plan(lib, 0, psfs, B, ny, nx, 'f64', monkeypatch)
    assert d.strategy()['direct_stencil'] and not d.strategy()['separable']
    assert not fft.strategy()['direct_stencil'] and not fft.strategy()['separable']
    if py * px <= 49:
        assert lib.DeconvPlan(psfs, B, ny, nx, dtype='f64').strategy()['direct_stencil']     # the default
    assert max_rel(d.forward(obj), fft.forward(obj)) < 1e-12
    y = rng.random((B, V, ny, nx))
    for normalize in (True, False):
        assert max_rel(d.adjoint(y, normalize), fft.adjoint(y, normalize)) < 1e-12
    for p in (d, fft):
        p.set_object(obj, 1e7)
        p.simulate(seed=9)
    assert max_rel(d.noiseless(), fft.noiseless()) < 1e-12
    assert np.array_equal(d.measurement(), fft.measurement())
    for p in (d, fft):
        p.iterate(25)
    assert max_rel(d.estimate(), fft.estimate()) < 1e-11
    o = orc.Deconvolver(psfs)
    o.create_data_from_object(obj[:1], noisy_measurement=[m[None] for m in d.measurement()[0]])
    for _ in range(25):
        o.iterate()
    assert max_rel(d.estimate()[0], o.estimate[0]) < 1e-10
    f32 = _direct_plan(lib, 2, psfs, B, ny, nx, 'f32', monkeypatch)
    assert f32.strategy()['direct_stencil']
    f32.set_measurement(d.measurement())
    f32.iterate(25)
    assert max_rel(f32.estimate(), d.estimate()) < 1e-5


def test_direct_stencil_keeps_f32_plans_accurate_on_dark_backgrounds(lib, monkeypatch):
    """What the direct stencil is for beyond speed: sparse emitters on a black background under a narrow, non-separable PSF
    (tests/test_gpu_parity.py::test_dark_background_narrow_psf_stays_finite).  The FFT path's f32 predictions of the dark region are
    rounding noise (finite, but 1e-5 ... 4e-5 of the maximum from the float64 plan, and counted by rl_deconv_unresolved); the
    stencil sums non-negative terms only, so every prediction keeps its relative accuracy: f32 within 3e-6 of float64."""
    rng = np.random.default_rng(77)
    ny = nx = 256
    obj = np.zeros((2, ny, nx))
    for b in range(2):
        obj[b, rng.integers(8, ny - 8, 30), rng.integers(8, nx - 8, 30)] = rng.random(30) + 0.5
    yy, xx = np.mgrid[-4:5, -4:5]
    u, w = 0.866 * xx + 0.5 * yy, -0.5 * xx + 0.866 * yy
    psf = [np.exp(-0.5 * ((u / 1.6) ** 2 + (w / 0.8) ** 2))[None]]
    p64 = _direct_plan(lib, 0, psf, 2, ny, nx, 'f64', monkeypatch)
    p64.set_object(obj, 3e3)
    p64.simulate(seed=9)
    meas = p64.measurement()
    p64.iterate(12)
    ref = p64.estimate()
    err = {}
    for mode in (0, 2):
        p32 = _direct_plan(lib, mode, psf, 2, ny, nx, 'f32', monkeypatch)
        assert p32.strategy()['direct_stencil'] == (mode == 2)
        p32.set_measurement(meas)
        p32.iterate(12)
        e = p32.estimate()
        assert np.isfinite(e).all() and e.min() >= 0
        err[mode] = max_rel(e, ref)
    print('f32 against float64, 12 iterations: FFT path %.2e, direct stencil %.2e' % (err[0], err[2]))
    assert err[2] < 3e-6 and err[2] < 0.3 * err[0]
