"""PSF half of the path on the GPU: psf_report / generate_psfs / tune_psf /
gaussian_filter / get_width of the mirror module against golden vectors from the
reference and against the CPU oracle.  float64 on the device.

Tolerances: PSF arrays 1e-12 normwise (same arithmetic, same operation order);
fitted widths 1e-6: the restated MINPACK iteration stops (like the reference's) up to
~1e-6 short of the minimum, so 1-ulp input differences (device exp2 vs numpy power) can move it that far;
tune_psf 1e-6 (Brent amplifies rounding-level differences of its objective).
"""
import os

import numpy as np
import pytest

from conftest import max_rel, fuzz_seeds
from oracle import line_sted_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def st():
    from rescan_line_sted_amd import line_sted_tools
    return line_sted_tools


def _parse(key):
    t, s, e, d, p = key.split('_')
    return t, float(s[1:]), float(e[1:]), float(d[1:]), int(p[1:])


def test_g1_psf_report_matches_reference(st, golden):
    g = golden('g1_psf_report')
    for key in g['cases']:
        key = str(key)
        psf_type, steps, exc, dep, pulses = _parse(key)
        r = st.psf_report(psf_type, exc, dep, steps, pulses, verbose=False)
        sc = g[key + '/scalars']
        want_keys = {'resolution_improvement_descanned', 'excitation_dose', 'depletion_dose',
                     'expected_emission', 'pulses_per_position', 'psfs'}
        if psf_type == 'line':
            want_keys.add('resolution_improvement_rescanned')
        assert set(r.keys()) == want_keys
        assert r['resolution_improvement_descanned'] == pytest.approx(sc[0], rel=1e-6)
        if psf_type == 'line':
            assert r['resolution_improvement_rescanned'] == pytest.approx(sc[1], rel=1e-6)
        assert r['excitation_dose'] == pytest.approx(sc[2], rel=1e-12)
        assert r['depletion_dose'] == pytest.approx(sc[3], rel=1e-12, abs=1e-300)
        assert r['expected_emission'] == pytest.approx(sc[4], rel=1e-12)
        assert r['pulses_per_position'] == pulses
        if pulses != 1:
            continue
        n = r['psfs']['sted'].shape[1]
        assert set(r['psfs'].keys()) == ({'excitation', 'depletion', 'excitation_fraction', 'depletion_fraction',
                                          'sted', 'descan_sted'} | ({'rescan_sted'} if psf_type == 'line' else set()))
        if psf_type == 'point':
            assert r['psfs']['descan_sted'] is r['psfs']['sted']       # alias, ref:249
        for k, v in r['psfs'].items():
            assert v.shape == (1, n, n) and v.dtype == np.float64
            if key + '/psf/' + k in g:
                assert max_rel(v, g[key + '/psf/' + k]) < 1e-12, (key, k)
            else:
                assert max_rel(v[0, n // 2, :], g[key + '/row/' + k]) < 1e-12, (key, k)
                assert max_rel(v[0, :, n // 2], g[key + '/col/' + k]) < 1e-12, (key, k)
        # the reference's exact-equality invariants hold on the device result
        assert r['psfs']['sted'][0, n // 2, :].max() == r['psfs']['sted'].max()
        assert r['psfs']['excitation'][0, n // 2, :].max() == r['psfs']['excitation'].max()


def test_generate_psfs_general_shapes_vs_oracle(st):
    # non-square shapes, and shapes small enough for the 'reflect' boundary to matter
    for shape, sigma in (((1, 31, 45), 3.0), ((1, 44, 30), 2.2), ((1, 9, 9), 2.5), ((1, 15, 21), 4.0)):
        for psf_type in ('point', 'line'):
            got = st.generate_psfs(shape, 0.7, 5.0, sigma, psf_type, verbose=False)
            ref = orc.generate_psfs(shape, 0.7, 5.0, sigma, psf_type)
            ref.pop('line_rescan_ratio', None)
            assert set(got) == set(ref)
            for k in ref:
                assert max_rel(got[k], ref[k]) < 1e-12, (shape, psf_type, k)


@pytest.mark.parametrize('seed', fuzz_seeds(8))
def test_random_psf_parameters_vs_oracle(st, seed):
    """Soak test of the PSF half (RLSTED_FUZZ_SEEDS): generate_psfs on random shapes (odd / even / non-square, small enough for
    the 'reflect' boundary to wrap more than once), blur widths and brightnesses, point and line, and psf_report's scalars and
    PSFs at random operating points -- device against the oracle."""
    rng = np.random.default_rng(70000 + seed)
    shape = (1, int(rng.integers(5, 90)), int(rng.integers(5, 90)))
    sigma = float(rng.uniform(0.6, 7.0))
    exc, dep = float(rng.uniform(0.02, 3.0)), float(rng.choice([0.0, rng.uniform(0.1, 40.0)]))
    psf_type = ('point', 'line')[int(rng.integers(0, 2))]
    if dep == 0.0:                                   # (no depletion: ref:204-207 divides by its maximum)
        dep = 0.5
    if psf_type == 'line' and shape[2] < 5 * sigma:  # the line's row must fit the grid: a truncated row fits to a width of ~0 and a
        shape = (1, shape[1], int(5 * sigma) + 1)    # rescan ratio of 1e4 ... 1e5 (the C ABI refuses ratios above 4096)
    got = st.generate_psfs(shape, exc, dep, sigma, psf_type, verbose=False)
    ref = orc.generate_psfs(shape, exc, dep, sigma, psf_type)
    ref.pop('line_rescan_ratio', None)
    assert set(got) == set(ref)
    for k in ref:
        assert max_rel(got[k], ref[k]) < 1e-10, (shape, psf_type, sigma, exc, dep, k)
    steps = float(rng.uniform(4.0, 16.0))
    pulses = int(rng.choice([1, 1, 2, 5]))
    r = st.psf_report(psf_type, exc, dep, steps, pulses, verbose=False)
    o = orc.psf_report(psf_type, exc, dep, steps, pulses)
    o.pop('line_rescan_ratio', None)                 # (an extra of the oracle: the integer rescan ratio of ref:252-256)
    assert set(r) == set(o)
    for k in o:
        if k == 'psfs':
            for name, v in o['psfs'].items():
                assert max_rel(r['psfs'][name], v) < 1e-11, (psf_type, steps, exc, dep, pulses, name)
        elif k.startswith('resolution'):
            # (a sted PSF below two pixels wide -- steps per excitation width / improvement -- is a spike on the grid: the Gaussian fit
            # of ref:108 is then ill conditioned and its Levenberg-Marquardt path turns on the last bit of its input)
            if steps / o[k] >= 2.0:
                assert r[k] == pytest.approx(o[k], rel=2e-6), (psf_type, steps, exc, dep, pulses, k)
        else:
            assert r[k] == pytest.approx(o[k], rel=1e-11, abs=1e-300), (psf_type, steps, exc, dep, pulses, k)


@pytest.mark.parametrize('seed', fuzz_seeds(6))
def test_random_rotations_filters_and_quality_metrics_vs_oracle(seed):
    """Soak test of the widened rows (RLSTED_FUZZ_SEEDS): the spline rotation of PSF stacks (any angle, odd / even / non-square
    planes), gaussian_filter with per-axis widths, get_width, the Fourier-error metrics and the radial error profile -- device
    against the oracle."""
    from rescan_line_sted_amd import psf, quality
    rng = np.random.default_rng(52000 + seed)
    ny, nx = int(rng.integers(5, 120)), int(rng.integers(5, 120))
    x = rng.random((int(rng.integers(1, 4)), ny, nx))          # (stacks too: fig2:271 clips at the ARRAY's maximum)
    x = x * rng.uniform(0.2, 3.0, (x.shape[0], 1, 1))
    deg = float(rng.choice([0.0, 90.0, -90.0, 180.0, 45.0, rng.uniform(-180, 180)]))
    assert np.abs(psf.rotate(x, deg) - orc.rotate(x, deg)).max() < 1e-11, (ny, nx, deg)
    sig = tuple(float(v) for v in rng.choice([0.0, 0.7, 1.9, 3.3, 6.5], 3))
    assert max_rel(psf.gaussian_filter(x, sig), orc.gaussian_filter(x, sig)) < 1e-12, (ny, nx, sig)
    n = int(rng.integers(16, 200))                   # (a peak at least 1.5 samples wide: below that the fit of ref:653-668 collapses onto one
    t = np.arange(n) - n / 2 + rng.uniform(-2, 2)    # sample and its Levenberg-Marquardt path turns on the last bit of the input)
    row = rng.uniform(0.2, 5) * np.exp(-0.5 * (t / rng.uniform(1.5, max(2.0, n / 8))) ** 2) + rng.random(n) * 1e-3
    assert psf.get_width(row)[0] == pytest.approx(orc.get_width(row)[0], rel=1e-6), n
    sy, sx = int(rng.integers(4, 150)), int(rng.integers(4, 150))
    est, obj = rng.random((sy, sx)) * 10, rng.random((sy, sx)) * 10
    assert max_rel(quality.fourier_error(est, obj), orc.fourier_error(est, obj)) < 1e-11, (sy, sx)
    hist = np.stack([est, est * 0.5 + obj * 0.5, obj + 1e-3])
    assert max_rel(quality.ft_error_history(hist, obj), orc.ft_error_history(hist, obj)) < 1e-11, (sy, sx)
    ang, rad, smp = float(rng.uniform(0, 180)), float(rng.uniform(0.05, 0.45)), int(rng.integers(10, 400))
    got = quality.error_vs_spatial_frequency(est, obj, ang, rad, smp)
    want = orc.error_vs_spatial_frequency(est, obj, ang, rad, smp)
    assert max_rel(np.asarray(got), want) < 1e-9, (sy, sx, ang, rad, smp)


def test_gaussian_filter_vs_oracle():
    from rescan_line_sted_amd import psf
    rng = np.random.default_rng(3)
    a = rng.random((3, 17, 23))
    for sigma in (1.7, (0, 0, 2.5), (0.8, 3.1, 0.0), 6.0):
        assert max_rel(psf.gaussian_filter(a, sigma), orc.gaussian_filter(a, sigma)) < 1e-13


def test_get_width(st, golden):
    g = golden('g2_get_width')
    for row, fit, (n, w) in zip(g['rows'], g['fits'], g['n_and_width']):
        n = int(n)
        s, f = st.get_width(row[:n])
        assert abs(s) == pytest.approx(abs(w), rel=1e-8)
        assert max_rel(f, fit[:n]) < 1e-7


G3_POINTS = [d + s for d in ('1p0x', '1p5x', '2p0x', '2p5x', '3p0x', '4p0x') for s in ('_ld', '_lr')]


@pytest.mark.parametrize('which', ['point', 'line'])
@pytest.mark.parametrize('name', G3_POINTS)
def test_g3_tune_psf_matches_reference(st, golden, name, which):
    """All twelve figure-2 operating points (line_sted_figure_2.py:77-162), point and line tuning each, on the device
    against what the reference's tune_psf returned (G3)."""
    g = golden('g3_tune_psf')
    pr, lr, pe, le, nori, maxexc, resc = g[name + '/inputs']
    if which == 'point':
        r = st.tune_psf('point', 'descanned', float(pr), float(pe),
                        max_excitation_brightness=float(maxexc), steps_per_improved_psf_width=4.)
    else:
        r = st.tune_psf('line', 'rescanned' if resc else 'descanned', float(lr), float(le),
                        max_excitation_brightness=float(maxexc), steps_per_improved_psf_width=4.)
    ref = dict(zip([str(k) for k in g['keys']], g[name + '/' + which]))
    for k in ('excitation_brightness', 'depletion_brightness', 'pulses_per_position',
              'steps_per_excitation_psf_width', 'excitation_dose', 'depletion_dose',
              'expected_emission', 'resolution_improvement_descanned'):
        assert r[k] == pytest.approx(ref[k], rel=1e-6, abs=1e-9), k
    assert 'psfs' in r and r['psf_type'] == which and r['verbose'] is False
    import pickle
    pickle.loads(pickle.dumps(r))            # line_sted_figure_2.py:165 pickles these dicts


@pytest.mark.parametrize('seed', fuzz_seeds(2))
def test_random_operating_points_tune_psf_vs_oracle(st, seed):
    """Soak test of tune_psf (RLSTED_FUZZ_SEEDS): operating points the figures do not use -- resolution improvements 1.2 ... 3.5,
    0.5 ... 20 emissions per molecule, 3 or 4 steps per improved PSF width, all three (type, scan) pairs -- the device's Brent search
    against the oracle's (the same iterates up to rounding: 1e-5 on the tuned values)."""
    rng = np.random.default_rng(63000 + seed)
    psf_type, scan = (('point', 'descanned'), ('line', 'descanned'), ('line', 'rescanned'))[int(rng.integers(0, 3))]
    R = float(rng.uniform(1.2, 3.5))
    em = float(rng.uniform(0.5, 20.0))
    maxexc = float(rng.choice([0.25, 0.5, 1.0]))
    steps = float(rng.choice([3.0, 4.0]))
    got = st.tune_psf(psf_type, scan, R, em, max_excitation_brightness=maxexc, steps_per_improved_psf_width=steps, verbose_results=False)
    want = orc.tune_psf(psf_type, scan, R, em, max_excitation_brightness=maxexc, steps_per_improved_psf_width=steps)
    case = (psf_type, scan, R, em, maxexc, steps)
    for k in ('excitation_brightness', 'depletion_brightness', 'steps_per_excitation_psf_width', 'pulses_per_position',
              'expected_emission', 'excitation_dose', 'depletion_dose', 'resolution_improvement_' + scan):
        # (abs: a rescanned line improves the resolution by ~1.4 with no depletion at all -- below that target the search ends at a
        # depletion brightness of ~1e-11, rounding noise in both implementations)
        assert got[k] == pytest.approx(want[k], rel=1e-5, abs=1e-7), (k,) + case


def test_fig2_psf_set_feeds_the_deconvolver(st, golden, tmp_path):
    """psf_comparison_pair's point branch (line_sted_figure_2.py:220-238) rebuilt from
    the mirror module reproduces the golden fig-2 PSF."""
    g = golden('g8_fig2_psfs')
    keys = ('excitation_brightness', 'depletion_brightness', 'pulses_per_position')
    point = dict(zip(keys, g['2p0x_lr/point'][:3]))
    fine = st.psf_report('point', point['excitation_brightness'], point['depletion_brightness'], 25,
                         point['pulses_per_position'], verbose=False)
    emission = g['2p0x_lr/point'][6]
    psf = emission * fine['psfs']['descan_sted'] / fine['psfs']['descan_sted'].sum()
    assert max_rel(psf, g['2p0x_lr/point_sted_psf'][0]) < 1e-12


def test_error_behaviour(st):
    with pytest.raises(AssertionError):
        st.tune_psf('point', 'rescanned', 1.5, 4.0)          # ref:398-400
    with pytest.raises(AssertionError):
        st.Deconvolver([np.ones((1, 3, 3))], verbose=False).create_data_from_object(
            np.ones((4, 4)))                                  # ref:502
    with pytest.raises(AssertionError):
        st.Deconvolver([np.ones((1, 3, 3))], verbose=False).create_data_from_object(
            np.ones((1, 4, 4), dtype=np.float32))             # ref:503


def test_rotate_matches_reference_psf_sets(st, golden):
    """rotate (line_sted_figure_2.py:264-272) on the device: general angles against
    the CPU oracle, and the complete figure-2 PSF sets (tune_psf x2, fine psf_report x2,
    normalisation, rotation) against the reference's own (G8)."""
    from rescan_line_sted_amd import psf
    rng = np.random.default_rng(5)
    for shape in ((1, 23, 23), (1, 17, 30), (1, 107, 107)):
        a = rng.random(shape)
        for deg in (45, 60, 22.5, 135, 120, 0, 90):
            assert max_rel(psf.rotate(a, deg), orc.rotate(a, deg)) < 1e-12, (shape, deg)
    g3 = golden('g3_tune_psf')
    # G8: three operating points; G8b: the two remaining line-rescan doses of BASELINE config 2 (6 and 8 orientations)
    for name in ('1p0x_ld', '1p5x_lr', '2p0x_lr', '2p5x_lr', '3p0x_lr'):
        g8 = golden('g8_fig2_psfs' if name + '/point' in golden('g8_fig2_psfs') else 'g8b_fig2_psfs_more')
        pr, lr, pe, le, nori, maxexc, resc = g3[name + '/inputs']
        c = psf.psf_comparison_pair(pr, lr, pe, le, 'rescanned' if resc else 'descanned', int(nori),
                                    max_excitation_brightness=float(maxexc))
        assert max_rel(c['point_sted_psf'][0], g8[name + '/point_sted_psf'][0]) < 1e-6
        assert len(c['line_sted_psfs']) == int(nori)
        for a, b in zip(c['line_sted_psfs'], g8[name + '/line_sted_psfs']):
            assert max_rel(a, b) < 1e-6


def test_reduced_figure_2_sweep(golden):
    """BASELINE config 4 in miniature: objects x PSF sets x seeds through sweep.py.  A task's
    result depends on (object, PSF set, seed) only: running it alone in a plan of one frame, with
    its own Philox key, reproduces the sweep's frame bit for bit, whatever it was batched with."""
    from rescan_line_sted_amd import sweep, _lib
    g8, objs = golden('g8_fig2_psfs'), golden('objects')
    objects = {n: objs[n][0].astype(np.float64) for n in ('astronaut', 'lines', 'rings')}
    psf_sets = {'1p5x_point': [g8['1p5x_lr/point_sted_psf'][0]],
                '1p5x_line3': [p[None] for p in g8['1p5x_lr/line_sted_psfs'][:, 0]]}
    tasks, est = sweep.figure_2_sweep(objects, psf_sets, seeds=(0, 7), iterations=4, dtype='f32')
    assert len(tasks) == 12 and est.shape == (12, 128, 128)
    costs = sweep.task_costs(tasks, objects, psf_sets, 4)
    assert max(costs) == 3 * min(costs)                       # 3-view tasks weigh three point tasks
    ids = sweep.object_ids(objects)
    for i in (0, 5, 11):
        o, p, s = tasks[i]
        plan = _lib.DeconvPlan(psf_sets[p], 1, 128, 128, dtype='f32')
        plan.set_object(objects[o][None], 5e10)
        plan.simulate_keyed([s], [ids[o]])
        plan.iterate(4)
        assert np.array_equal(plan.estimate()[0], est[i])
    # batched differently (one frame per plan): same results
    alone = sweep.run_tasks(tasks[:5], objects, psf_sets, 4, max_frames_per_plan=1)
    assert all(np.array_equal(a, e) for a, e in zip(alone, est[:5]))


@pytest.mark.parametrize('seed', fuzz_seeds(3))
def test_random_sweeps_do_not_depend_on_their_batching(seed):
    """Soak test of sweep.py (RLSTED_FUZZ_SEEDS): random objects of several shapes, PSF sets of 1-3 views, seeds, task order,
    frames per plan and contexts -- a task's estimate is that of the task run alone in a one-frame plan: bit for bit where the
    plans do not pair frames (transforms shorter than 256), to f32 rounding of the brighter partner where they do."""
    from rescan_line_sted_amd import sweep
    rng = np.random.default_rng(91000 + seed)
    shapes = [(int(rng.integers(20, 300)), int(rng.integers(20, 300))) for _ in range(int(rng.integers(1, 4)))]
    objects = {'o%d' % i: rng.random(shapes[int(rng.integers(0, len(shapes)))]) * 10 + 0.1 for i in range(int(rng.integers(1, 6)))}
    psf_sets = {'p%d' % i: [rng.random((1, int(rng.integers(1, 12)), int(rng.integers(1, 12)))) + 0.01 for _ in range(int(rng.integers(1, 4)))]
                for i in range(int(rng.integers(1, 4)))}
    tasks = sweep.make_tasks(objects, psf_sets, [int(s) for s in rng.integers(0, 1000, int(rng.integers(1, 5)))])
    tasks = [tasks[i] for i in rng.permutation(len(tasks))][:int(rng.integers(1, len(tasks) + 1))]
    K = int(rng.integers(1, 5))
    # (photons per pixel of the smallest object: below ~1 the measurement is isolated photons, the predictions between them sit at the
    # f32 transforms' noise level whatever their sign, and a frame's estimate depends on rounding -- its partner's included)
    brightness = float(rng.choice([3.0, 300.0, 1e6])) * max(o.size for o in objects.values())
    sweep.clear_plans()
    est = sweep.run_tasks(tasks, objects, psf_sets, K, total_brightness=brightness, max_frames_per_plan=int(rng.choice([2, 3, 8, 256])))
    alone = sweep.run_tasks(tasks, objects, psf_sets, K, total_brightness=brightness, max_frames_per_plan=1)
    # a fraction of a photon per pixel leaves dark regions whose predictions an f32 transform does not resolve:
    # the plans say so (rl_deconv_unresolved), and the estimates of such a sweep are finite but depend on rounding -- also on the partner's
    resolved = sweep.unresolved_total() == 0
    for (o, p_, s_), a, e in zip(tasks, alone, est):
        assert np.isfinite(e).all() and e.min() >= 0 and e.shape == objects[o].shape
        if max(objects[o].shape) + 6 <= 192:
            assert np.array_equal(a, e), (o, p_, s_, objects[o].shape)
        elif resolved:
            assert np.abs(a - e).max() < 6e-6 * max(np.abs(x).max() for x in alone), (o, p_, s_, objects[o].shape)   # (measured: up to 3.0e-6)
    sweep.clear_plans()


def test_keyed_simulate_equals_plain_simulate_and_the_oracle_twin(golden):
    """seeds[f] = s, image_ids[f] = f is rl_deconv_simulate(s); arbitrary keys match the numpy
    twin of the sampler called with the same (seed, image) per view."""
    from rescan_line_sted_amd import _lib
    from oracle import philox_poisson as pp
    g8 = golden('g8_fig2_psfs')
    psfs = [p[None] for p in g8['1p5x_lr/line_sted_psfs'][:, 0]]          # 3 views
    rng = np.random.default_rng(2)
    objs = rng.random((4, 40, 56)) * 3
    plan = _lib.DeconvPlan(psfs, 4, 40, 56, dtype='f64')
    plan.set_object(objs, [30.0 * 40 * 56, 8.0 * 40 * 56, 2e4 * 40 * 56, 0.5 * 40 * 56])   # below / above the PTRS switch at 10
    plan.simulate(seed=77)
    plain = plan.measurement()
    plan.simulate_keyed([77] * 4, np.arange(4))
    assert np.array_equal(plan.measurement(), plain)
    seeds, ids = [5, 2 ** 40 + 3, 5, 9], [7, 0, 123456, 7]
    plan.simulate_keyed(seeds, ids)
    got, lam = plan.measurement(), plan.noiseless()
    for f in range(4):
        for v in range(3):
            ref = pp.poisson(lam[f, v].ravel(), seeds[f], ids[f] * 3 + v).reshape(40, 56) + 1e-9
            assert np.array_equal(got[f, v], ref), (f, v)


# ------------------------------------------------ reconstruction quality
def test_quality_metrics_match_reference(golden):
    """Device Fourier-error metrics against the reference's own numbers (g10)."""
    from rescan_line_sted_amd import quality
    g = golden('g10_quality')
    est, truth = g['estimate'][0], g['true_object'][0]
    fe = quality.fourier_error(est, truth)
    assert max_rel(fe, g['fourier_error']) < 1e-12
    for tag in ('best', 'worst'):
        ang = float(g['angle_' + tag])
        raw = quality.error_vs_spatial_frequency(est, truth, angle_degrees=ang, smooth=False)
        assert max_rel(raw, g['profile_raw_' + tag]) < 1e-11
        assert max_rel(quality.error_vs_spatial_frequency(est, truth, angle_degrees=ang), g['profile_' + tag]) < 1e-11
    hist = quality.ft_error_history(g['estimate_history_tif'].astype(np.float64), truth)
    assert max_rel(hist, g['ft_error_history_tif']) < 2e-5          # inputs and outputs are float32 on disk


@pytest.mark.parametrize('shape', [(1, 1), (2, 3), (7, 1), (33, 64), (160, 160), (100, 37)])
def test_fft2_magnitude_any_shape_vs_numpy(shape):
    from rescan_line_sted_amd import quality
    rng = np.random.default_rng(sum(shape))
    x, t = rng.random((3,) + shape), rng.random(shape)
    ref = np.abs(np.fft.fftshift(np.fft.fftn(x - t, axes=(1, 2)), axes=(1, 2)))
    assert max_rel(quality.fourier_error(x, t), ref / (shape[0] * shape[1])) < 1e-12
    assert max_rel(quality.ft_error_history(x, t), np.log(1 + ref)) < 1e-12
    assert max_rel(quality.fourier_error(x[0], t), orc.fourier_error(x[0], t)) < 1e-12


def test_map_coordinates_vs_oracle():
    from rescan_line_sted_amd import quality
    rng = np.random.default_rng(9)
    a = rng.random((40, 53))
    ys, xs = rng.uniform(-3, 43, 700), rng.uniform(-3, 56, 700)
    ys[:4], xs[:4] = [0, 39, 39, 0], [0, 52, 0, 52]
    got = quality.map_coordinates(a, [ys, xs])
    assert np.abs(got - orc.map_coordinates_cubic(a, ys, xs)).max() < 1e-13
    assert quality.map_coordinates(a, np.zeros((2, 0))).shape == (0,)


def test_record_iteration_writes_reference_history(st, golden, tmp_path):
    """Deconvolver.record_iteration through the device FT-error path reproduces the files the
    reference wrote for the same run (same measurement, same save schedule)."""
    from rescan_line_sted_amd import np_tif
    g = golden('g10_quality')
    psf = list(golden('g8_fig2_psfs')['1p5x_lr/point_sted_psf'])
    d = st.Deconvolver(psf, output_prefix=str(tmp_path) + '/q_', verbose=False)
    d.create_data_from_object(golden('objects')['rings'].astype(np.float64), total_brightness=5e10, random_seed=0)
    d.noisy_measurement = [m.copy() for m in g['noisy']]
    for _, flag in st.logarithmic_progress(range(9), verbose=False):
        d.iterate()
        if flag:
            d.record_iteration()
    assert d.saved_iterations == list(g['saved_iterations'])
    eh = np_tif.tif_to_array(str(tmp_path) + '/q_estimate_history.tif')
    fh = np_tif.tif_to_array(str(tmp_path) + '/q_estimate_FT_error_history.tif')
    assert eh.dtype == np.float32 and fh.dtype == np.float32
    assert max_rel(eh, g['estimate_history_tif']) < 1e-6
    assert max_rel(fh, g['ft_error_history_tif']) < 1e-6


# ---------------------------------------------------------------- batched psf_report (figure 1's sweep)
def figure_1_parameter_sets():
    """line_sted_figure_1.py:33-48 (480 combinations) plus the fine-grid report of :65-71 per (type, exc, dep)."""
    sets = [(p, exc, dep, samps, pulses) for exc in (0.25, 1, 4) for dep in (0, 1, 3, 9, 27) for samps in (4, 6, 8, 12)
            for pulses in (1, 2, 4, 8) for p in ('line', 'point')]
    assert len(sets) == 480
    return sets


def test_psf_report_batch_is_bit_for_bit_the_unbatched_call():
    from rescan_line_sted_amd import psf
    sets = figure_1_parameter_sets()
    pick = sets[::7] + [('line', 0.25, 9, 25, 1), ('point', 4, 27, 30, 2), ('line', 1, 0, 30, 1)]
    batch = psf.psf_report_batch(pick, with_psfs=True)
    for p, b in zip(pick, batch):
        one = psf.psf_report(*p, verbose=False)
        assert sorted(one) == sorted(b)
        for k in one:
            if k != 'psfs':
                assert one[k] == b[k] or (np.isnan(one[k]) and np.isnan(b[k])), (p, k, one[k], b[k])
        assert sorted(one['psfs']) == sorted(b['psfs'])
        for k in one['psfs']:
            assert np.array_equal(one['psfs'][k], b['psfs'][k]), (p, k)


def test_psf_report_batch_figure_1_sweep_time_and_golden(golden):
    import time
    from rescan_line_sted_amd import psf
    sets = figure_1_parameter_sets()
    psf.psf_report_batch(sets[:16])                       # warm-up (workspace allocation)
    t0 = time.perf_counter()
    reports = psf.psf_report_batch(sets)
    el = time.perf_counter() - t0
    print('480 psf_report sets in one batch: %.1f ms' % (el * 1e3))
    assert el < 0.25                                      # ~20 launches + 2 syncs per set unbatched: seconds
    g = golden('g1_psf_report')
    hits = 0
    for p, r in zip(sets, reports):
        key = '%s_s%d_e%g_d%g_p%d/scalars' % (p[0], p[3], p[1], p[2], p[4])
        if key in g.files:
            sc = g[key]
            hits += 1
            assert abs(r['resolution_improvement_descanned'] - sc[0]) < 1e-6 * abs(sc[0])      # a fitted width
            for k, i in (('excitation_dose', 2), ('depletion_dose', 3), ('expected_emission', 4)):
                assert abs(r[k] - sc[i]) <= 1e-9 * max(abs(sc[i]), 1e-30), (p, k)
    assert hits >= 20


def test_psf_report_output_dir_writes_the_reference_file_set(st, golden, tmp_path):
    """psf_report / generate_psfs(output_dir=...) (ref:311-346): the same nine (line) / five (point) file names as the
    reference, including emission_psf.tif and sted_psf_line_rescan_unscaled.tif (rl_psf_generate_line_extras), whose
    contents match what the reference wrote (G1b; float32 on disk) and the oracle."""
    from rescan_line_sted_amd import np_tif
    g = golden('g1b_line_dumps')
    exc, dep, steps, pulses = g['args']
    out = str(tmp_path / 'line')
    st.psf_report('line', float(exc), float(dep), float(steps), float(pulses), verbose=False, output_dir=out)
    assert sorted(os.listdir(out)) == [str(f) for f in g['files']]
    for name in ('emission_psf.tif', 'sted_psf_line_rescan_unscaled.tif', 'sted_psf_line_rescan.tif'):
        got = np_tif.tif_to_array(os.path.join(out, name))
        assert got.shape == g[name].shape and got.dtype == np.float32
        assert max_rel(got, g[name]) < 1e-6, name
    sigma = steps / (2 * np.sqrt(2 * np.log(2)))
    n = 1 + 2 * int(np.round(5 * sigma))
    p = orc.generate_psfs((1, n, n), exc, dep, sigma, psf_type='line', with_intermediates=True)
    from rescan_line_sted_amd import psf
    extras = psf._line_extras(n, n, exc, dep, sigma, p['line_rescan_ratio'])
    assert max_rel(extras['emission_psf.tif'], p['emission_psf']) < 1e-13
    assert max_rel(extras['sted_psf_line_rescan_unscaled.tif'], p['rescan_sted_unscaled']) < 1e-12
    out = str(tmp_path / 'point')
    st.generate_psfs((1, 27, 27), 0.25, 9.0, 3.4, psf_type='point', output_dir=out, verbose=False)
    assert len(os.listdir(out)) == 5


def test_line_extras_argument_errors_are_reported():
    """rl_psf_generate_line_extras: a rescan ratio < 1 (the caller must pass the ratio a previous generate reported: it sizes
    the second buffer) and a NULL context are errors, not crashes; NULL output pointers are allowed."""
    import ctypes
    from rescan_line_sted_amd import _lib, psf
    ctx = psf._ctx()
    rc = _lib.lib.rl_psf_generate_line_extras(ctx.handle, 27, 27, 0.25, 9.0, 3.4, 0, None, None)
    assert rc != 0 and b'rescan_ratio' in _lib.lib.rl_last_error()
    assert _lib.lib.rl_psf_generate_line_extras(None, 27, 27, 0.25, 9.0, 3.4, 3, None, None) != 0
    em = np.empty((27, 27))
    _lib.check(_lib.lib.rl_psf_generate_line_extras(ctx.handle, 27, 27, 0.25, 9.0, 3.4, 3, _lib.ptr(em), None))
    assert em.sum() == pytest.approx(1.0, rel=1e-12) and em.argmax() == 13 * 27 + 13      # a normalised Gaussian centred on the array
