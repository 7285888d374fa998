"""The CPU oracle (oracle/line_sted_oracle.py) against golden vectors produced
by the real reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import line_sted_oracle as orc
from conftest import max_rel

PSF_TOL = 1e-12     # PSF arrays: same arithmetic, float64
FIT_TOL = 1e-10     # Gaussian-fit widths: same MINPACK iteration restated (oracle/minpack_lm.py)


def _parse(key):
    t, s, e, d, p = key.split('_')
    return t, float(s[1:]), float(e[1:]), float(d[1:]), int(p[1:])


def test_g1_psf_report(golden):
    g = golden('g1_psf_report')
    for key in g['cases']:
        key = str(key)
        psf_type, steps, exc, dep, pulses = _parse(key)
        r = orc.psf_report(psf_type, exc, dep, steps, pulses)
        sc = g[key + '/scalars']
        assert r['resolution_improvement_descanned'] == pytest.approx(sc[0], rel=FIT_TOL)
        if psf_type == 'line':
            assert r['resolution_improvement_rescanned'] == pytest.approx(sc[1], rel=FIT_TOL)
            assert r['line_rescan_ratio'] == int(g[key + '/ratio'][1])
        assert r['excitation_dose'] == pytest.approx(sc[2], rel=1e-12)
        assert r['depletion_dose'] == pytest.approx(sc[3], rel=1e-12, abs=1e-300)
        assert r['expected_emission'] == pytest.approx(sc[4], rel=1e-12)
        if pulses != 1:
            continue
        n = r['psfs']['sted'].shape[1]
        for k, v in r['psfs'].items():
            if key + '/psf/' + k in g:
                assert max_rel(v, g[key + '/psf/' + k]) < PSF_TOL, (key, k)
            else:
                assert max_rel(v[0, n // 2, :], g[key + '/row/' + k]) < PSF_TOL, (key, k)
                assert max_rel(v[0, :, n // 2], g[key + '/col/' + k]) < PSF_TOL, (key, k)


def test_g2_get_width(golden):
    g = golden('g2_get_width')
    for row, fit, (n, w) in zip(g['rows'], g['fits'], g['n_and_width']):
        n = int(n)
        s, f = orc.get_width(row[:n])
        assert abs(s) == pytest.approx(abs(w), rel=FIT_TOL)
        assert max_rel(f, fit[:n]) < 1e-6


def test_g4_conv_conventions(golden):
    g = golden('g4_conv')
    for name in ('odd', 'even', 'row', 'big', 'two'):
        x, psfs, y = g[name + '/x'], g[name + '/psfs'], g[name + '/y']
        d = orc.Deconvolver(list(psfs))
        H = d.H(x)
        for i in range(len(psfs)):
            assert max_rel(H[i], g[name + '/H'][i]) < 1e-13
            # the FFT restatement and the literal double loop agree
            direct = orc.conv_same_direct(x, psfs[i])
            assert max_rel(np.maximum(direct, 0), g[name + '/H'][i]) < 1e-13
        assert max_rel(d.H_t(list(y), normalize=False), g[name + '/Ht_raw']) < 1e-13
        assert max_rel(d.H_t(list(y)), g[name + '/Ht']) < 1e-12
        assert max_rel(d.H_t_normalization, g[name + '/norm']) < 1e-13


@pytest.mark.parametrize('run,ks', [('rings_point_1p5x', (1, 2, 5, 20, 100)),
                                    ('rings_line4_2p0x', (1, 5, 20)),
                                    ('cat_line1_1p0x', (5,))])
def test_g5_simulate_and_rl(golden, run, ks):
    g, psfs, objs = golden('g5_rl'), golden('g8_fig2_psfs'), golden('objects')
    psf_set = list(psfs[str(g[run + '/psf_key'])])
    obj = objs[str(g[run + '/object'])].astype(np.float64)
    d = orc.Deconvolver(psf_set)
    d.create_data_from_object(obj, total_brightness=5e10,
                              noisy_measurement=list(g[run + '/noisy']))
    for i in range(len(psf_set)):
        assert max_rel(d.noiseless_measurement[i], g[run + '/noiseless'][i]) < 1e-13
    for k in range(1, max(ks) + 1):
        d.iterate()
        if k in ks:
            ref = g[run + '/estimate_%d' % k]
            err = np.abs(d.estimate - ref) / np.abs(ref)      # pixelwise
            assert err.max() < 1e-9, (run, k, err.max())
    assert max_rel(d.H_t_normalization, g[run + '/norm']) < 1e-13


def test_g5_numpy_legacy_poisson_is_what_the_reference_draws(golden):
    """create_data_from_object(random_seed=0) -> np.random.seed + np.random.poisson
    on the noiseless images, view by view (ref:508-511)."""
    g, psfs, objs = golden('g5_rl'), golden('g8_fig2_psfs'), golden('objects')
    run = 'rings_point_1p5x'
    d = orc.Deconvolver(list(psfs[str(g[run + '/psf_key'])]))
    d.create_data_from_object(objs['rings'].astype(np.float64), 5e10, random_seed=0)
    # identical lambda to 1e-13 -> identical draws except where a PTRS
    # accept/reject decision sits within rounding; allow a handful
    same = d.noisy_measurement[0] == g[run + '/noisy'][0]
    assert same.mean() > 0.999


def test_g9_logarithmic_progress(golden):
    g = golden('g9_progress')
    for n in (0, 1, 2, 3, 5, 17, 1025):
        assert list(g['n%d' % n]) == orc.logarithmic_progress_flags(n)


@pytest.mark.parametrize('name,which', [('1p5x_ld', 'point'), ('1p5x_lr', 'line'), ('1p0x_ld', 'line')])
def test_g3_tune_psf(golden, name, which):
    """tune_psf at fig-2 operating points (inputs: line_sted_figure_2.py:77-162).
    The Brent search amplifies rounding-level differences of its objective, so
    agreement is ~1e-8, not 1e-12."""
    g = golden('g3_tune_psf')
    pr, lr, pe, le, nori, maxexc, resc = g[name + '/inputs']
    if which == 'point':
        r = orc.tune_psf('point', 'descanned', float(pr), float(pe),
                         max_excitation_brightness=maxexc, steps_per_improved_psf_width=4.)
    else:
        r = orc.tune_psf('line', 'rescanned' if resc else 'descanned', float(lr), float(le),
                         max_excitation_brightness=maxexc, steps_per_improved_psf_width=4.)
    ref = dict(zip([str(k) for k in g['keys']], g[name + '/' + which]))
    for k in ('excitation_brightness', 'depletion_brightness', 'pulses_per_position',
              'excitation_dose', 'depletion_dose', 'expected_emission',
              'resolution_improvement_descanned'):
        assert r[k] == pytest.approx(ref[k], rel=1e-6, abs=1e-9), k


def test_published_dose_table(golden):
    """appendix.html:222-247 of the reference: excitation / depletion dose of the
    six point-STED operating points (the only numbers the reference publishes)."""
    g = golden('g3_tune_psf')
    table = {'1p0x_ld': (5.8, 0.0), '1p5x_ld': (11.4, 500.3), '2p0x_ld': (18.8, 1696.8),
             '2p5x_ld': (30.3, 4159.5), '3p0x_ld': (46.7, 8601.8), '4p0x_ld': (90.7, 27044.0)}
    keys = [str(k) for k in g['keys']]
    for name, (exc, dep) in table.items():
        ref = dict(zip(keys, g[name + '/point']))
        assert abs(ref['excitation_dose'] - exc) < 0.06
        assert abs(ref['depletion_dose'] - dep) < 0.06


def test_spline_rotate_restates_scipy():
    from scipy import ndimage
    rng = np.random.default_rng(0)
    for shape in ((23, 23), (17, 30), (107, 107)):
        a = rng.random(shape)
        for deg in (45, 60, 22.5, 135, 120, 90, 180, 1e-3):
            ref = ndimage.rotate(a[None], angle=deg, axes=(1, 2), reshape=False)[0]
            assert np.abs(orc.spline_rotate(a, deg) - ref).max() < 1e-12, (shape, deg)


@pytest.mark.parametrize('name', ['1p0x_ld', '1p5x_lr', '2p0x_lr', '2p5x_lr', '3p0x_lr'])
def test_g8_fig2_psf_sets(golden, name):
    """psf_comparison_pair (tune_psf x2, fine psf_report x2, normalisation, rotation)
    against the PSF sets the reference's figure-2 script produced (G8; G8b: the 2.5x / 3.0x line-rescan sets)."""
    g3 = golden('g3_tune_psf')
    g8 = golden('g8_fig2_psfs' if name + '/point' in golden('g8_fig2_psfs') else 'g8b_fig2_psfs_more')
    pr, lr, pe, le, nori, maxexc, resc = g3[name + '/inputs']
    c = orc.psf_comparison_pair(pr, lr, pe, le, 'rescanned' if resc else 'descanned', int(nori),
                                max_excitation_brightness=maxexc)
    assert max_rel(c['point_sted_psf'][0], g8[name + '/point_sted_psf'][0]) < 1e-6
    assert len(c['line_sted_psfs']) == int(nori)
    for a, b in zip(c['line_sted_psfs'], g8[name + '/line_sted_psfs']):
        assert max_rel(a, b) < 1e-6                # tune_psf agrees to ~1e-8; see test_g3_tune_psf


def test_g10_quality_metrics(golden):
    """The Fourier-error metrics against what the reference wrote (record_iteration's float32
    TIFF history, line_sted_tools.py:533-547) and against the figure-2 harness' numpy/scipy
    calls (line_sted_figure_2.py:353-390) on the reference's own RL run."""
    g = golden('g10_quality')
    est, truth = g['estimate'][0], g['true_object'][0]
    assert list(g['saved_iterations']) == [2, 3, 5, 9]      # flags at i = 1, 2, 4, 8 (= last)
    fe = orc.fourier_error(est, truth)
    assert max_rel(fe, g['fourier_error']) < 1e-13
    for tag in ('best', 'worst'):
        z = orc.error_vs_spatial_frequency(est, truth, angle_degrees=float(g['angle_' + tag]))
        assert max_rel(z, g['profile_' + tag]) < 1e-12
    # the history on disk is float32 (np_tif coerces, np_tif.py:145-151) of float64 estimates
    hist = orc.ft_error_history(g['estimate_history_tif'].astype(np.float64), truth)
    assert max_rel(hist, g['ft_error_history_tif']) < 2e-5   # inputs known to float32 only


def test_map_coordinates_restates_scipy():
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(8)
    a = rng.random((23, 31))
    ys, xs = rng.uniform(-2, 25, 300), rng.uniform(-2, 33, 300)
    ys[:4], xs[:4] = [0, 22, 22, 0], [0, 30, 0, 30]                 # the corners are inside
    assert np.abs(orc.map_coordinates_cubic(a, ys, xs) - map_coordinates(a, [ys, xs])).max() < 1e-13


def test_g1b_line_dump_intermediates(golden):
    """The two arrays generate_psfs(output_dir=...) writes that are not in its return value (ref:339-341), against the
    TIFs the reference wrote (float32 on disk)."""
    g = golden('g1b_line_dumps')
    exc, dep, steps, pulses = g['args']
    sigma = steps / (2 * np.sqrt(2 * np.log(2)))
    n = 1 + 2 * int(np.round(5 * sigma))
    p = orc.generate_psfs((1, n, n), exc, dep, sigma, psf_type='line', with_intermediates=True)
    assert p['rescan_sted_unscaled'].shape == g['sted_psf_line_rescan_unscaled.tif'].shape
    assert max_rel(p['emission_psf'], g['emission_psf.tif']) < 1e-7
    assert max_rel(p['rescan_sted_unscaled'], g['sted_psf_line_rescan_unscaled.tif']) < 1e-7
    assert max_rel(p['rescan_sted'], g['sted_psf_line_rescan.tif']) < 1e-7
