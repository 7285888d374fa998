"""SURVEY.md row f-3 on the CPU: the numpy restatement of line_sted_figure_3.simulate_imaging against the
golden G11 (recorded from the reference's own function definitions, tests/golden/make_golden_fig3.py),
and its restatements of scipy.ndimage against scipy itself."""
import warnings

import numpy as np
import pytest

from oracle import figure3_oracle as f3

TOL = 1e-10


def run_case(golden, key, simulate):
    g = golden('g11_fig3')
    obj_name, rest = key.split('_', 1)
    imaging_type = rest.rsplit('_R', 1)[0]
    psf_width, R, n_orient, pulses, pad = g[key + '/args']
    frames = []

    def record(rot, which_pos, obj, exc, glow, inst, cum, new_signal, reconstruction, pulses_delivered, exposures):
        frames.append((rot, which_pos, [exc, glow, inst, cum, new_signal, reconstruction], pulses_delivered, exposures))
    res = simulate(g['obj/' + obj_name], imaging_type, psf_width, R, int(n_orient), int(pulses), int(pad), record)
    return g, frames, res


def check_case(g, key, frames):
    assert [int(f[0]) for f in frames] == g[key + '/frame_rot'].tolist()
    assert [int(f[1]) for f in frames] == g[key + '/frame_pos'].tolist()
    assert [f[3] for f in frames] == g[key + '/frame_pulses'].tolist()
    assert [-1 if f[4] == 'N/A' else f[4] for f in frames] == g[key + '/frame_exposures'].tolist()
    sums = np.array([[a.sum() for a in f[2]] for f in frames])
    maxs = np.array([[a.max() for a in f[2]] for f in frames])
    assert np.allclose(sums, g[key + '/frame_sums'], rtol=TOL, atol=TOL)
    assert np.allclose(maxs, g[key + '/frame_maxs'], rtol=TOL, atol=TOL)
    for i, full in zip(g[key + '/full_index'], g[key + '/full']):
        for a, b in zip(frames[int(i)][2], full):      # arrays are normalised to their maxima: absolute tolerance
            assert np.abs(a - b).max() < TOL, (key, int(i))


def test_g11_cases_cover_every_imaging_type(golden):
    cases = [str(c) for c in golden('g11_fig3')['cases']]
    assert {c.split('_', 1)[1].rsplit('_R', 1)[0] for c in cases} == set(f3.IMAGING_TYPES)
    assert {c.rsplit('_R', 1)[1] for c in cases} >= {'1', '2'}


@pytest.mark.parametrize('key', ['rings_descan_point_R1', 'rings_nondescan_multipoint_R1', 'lines_descan_line_R1',
                                 'lines_rescan_line_R1', 'rings_descan_point_R2', 'rings_nondescan_multipoint_R2', 'lines_descan_line_R2',
                                 'lines_rescan_line_R2', 'rings_rescan_line_R3'])
def test_g11_simulate_imaging(golden, key):
    g, frames, _ = run_case(golden, key, f3.simulate_imaging)
    check_case(g, key, frames)


def test_scipy_restatements():
    ndimage = pytest.importorskip('scipy.ndimage')
    rng = np.random.default_rng(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for shape in ((1, 37, 41), (1, 64, 64), (1, 82, 82)):
            x = rng.random(shape)
            for s in ((0, 3, -5), (0, -7, 0), (0, 12, 11)):
                assert np.abs(np.clip(ndimage.shift(x, s), 0, 1.1 * x.max()) - f3.shift(x, s)).max() < 1e-13
            for deg in (45.0, 30.0, 150.0, -60.0, -90.0, -135.0):
                ref = np.clip(ndimage.rotate(x, angle=deg, axes=(1, 2), mode='nearest', reshape=False), 0, 1.1 * x.max())
                assert np.abs(ref - f3.rotate(x, deg)).max() < 1e-12, (shape, deg)
        for ny, f in ((37, 0.5), (58, 0.2), (60, 0.1), (178, 0.5), (242, 0.1)):
            p = rng.random((ny, 23))
            assert np.abs(ndimage.zoom(p, zoom=(f, 1)) - f3.zoom_y(p, f)).max() < 1e-13
        p = rng.random((5, 20))
        assert np.abs(ndimage.spline_filter1d(p, order=3, axis=1, mode='nearest') -
                      f3._spline_prefilter_axis_reflect(p, 1)).max() < 1e-13
