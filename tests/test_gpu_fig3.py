"""SURVEY.md row f-3 on the device: rescan_line_sted_amd.line_sted_figure_3.simulate_imaging (scan loop
batched over scan positions in csrc/fig3_kernels.hip) against the golden G11 recorded from the
reference's own function definitions, and against the CPU oracle on shapes the golden does not hold.
float64 throughout; arrays reach the figure code normalised to their maxima: absolute tolerance 1e-10."""
import os

import numpy as np
import pytest

from oracle import figure3_oracle as f3
from conftest import fuzz_seeds
from test_fig3_oracle import check_case, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fig3():
    from rescan_line_sted_amd import _lib, line_sted_figure_3
    assert _lib.device_count() >= 1, 'no GPU visible'
    return line_sted_figure_3


def device_simulate(fig3):
    def simulate(obj, imaging_type, psf_width, R, n_orient, pulses, pad, record):
        def generate_figure(filename, obj_, exc, glow, inst, cum, new_signal, reconstruction, pulses_delivered, exposures):
            base = os.path.basename(filename)       # <type>_<rot>deg_<comparison>_<position>.svg, as the reference names it
            rot = int(base.split('deg_')[0].rsplit('_', 1)[1])
            pos = int(base.rsplit('_', 1)[1].split('.')[0])
            record(rot, pos, obj_, exc, glow, inst, cum, new_signal, reconstruction, pulses_delivered, exposures)
        return fig3.simulate_imaging(obj, imaging_type, psf_width, R, n_orient, pulses, pad, comparison_name='case',
                                     generate_figure=generate_figure)
    return simulate


@pytest.mark.parametrize('key', ['rings_descan_point_R1', 'rings_nondescan_multipoint_R1', 'lines_descan_line_R1',
                                 'lines_rescan_line_R1', 'rings_descan_point_R2', 'rings_nondescan_multipoint_R2',
                                 'lines_descan_line_R2', 'lines_rescan_line_R2', 'rings_rescan_line_R3'])
def test_g11_simulate_imaging_on_the_device(fig3, golden, key):
    g, frames, res = run_case(golden, key, device_simulate(fig3))
    check_case(g, key, frames)
    # 10 repeats of the last frame name per orientation (:259-261)
    assert len(res['filenames']) == len(frames) + 10 * len(res['reconstructions'])


def test_rotate_image_vs_oracle(fig3):
    rng = np.random.default_rng(5)
    for shape in ((1, 37, 41), (1, 64, 64), (1, 90, 50)):
        x = rng.random(shape)
        for deg in (45.0, 30.0, -60.0, 90.0, -135.0, 150.0):
            assert np.abs(fig3.rotate(x, deg) - f3.rotate(x, deg)).max() < 1e-12, (shape, deg)
    assert np.array_equal(fig3.rotate(x, 0), x)


@pytest.mark.parametrize('imaging_type,n_orient', [('descan_point', 1), ('nondescan_multipoint', 1), ('descan_line', 3),
                                                   ('rescan_line', 3)])
def test_other_shapes_vs_oracle(fig3, imaging_type, n_orient):
    """Non-square object, odd sizes, non-integer R, pulses_per_position 3: device against the CPU oracle."""
    rng = np.random.default_rng(11)
    obj = rng.random((1, 27, 38)) + 1e-6
    pad = 9 if n_orient == 1 else int(0.45 * 38)
    want, got = [], []
    f3.simulate_imaging(obj, imaging_type, 9, 1.5, n_orient, 3, pad,
                        lambda rot, pos, *a: want.append((rot, pos, a)))
    device_simulate(fig3)(obj, imaging_type, 9, 1.5, n_orient, 3, pad, lambda rot, pos, *a: got.append((rot, pos, a)))
    assert [(int(r), p) for r, p, _ in got] == [(int(r), p) for r, p, _ in want]
    for (_, _, a), (_, _, b) in zip(got, want):
        for x, y in zip(a[:7], b[:7]):
            assert np.abs(x - y).max() < 1e-10
        assert a[7:] == b[7:]


@pytest.mark.parametrize('seed', fuzz_seeds(4))
def test_random_simulate_imaging_vs_oracle(fig3, seed):
    """Soak test of simulate_imaging (RLSTED_FUZZ_SEEDS): random object shapes, imaging types, PSF widths, non-integer resolution
    improvements, orientations, pulses per position and paddings -- device against the CPU oracle, frame by frame."""
    rng = np.random.default_rng(41000 + seed)
    imaging_type = ('descan_point', 'nondescan_multipoint', 'descan_line', 'rescan_line')[int(rng.integers(0, 4))]
    line = imaging_type.endswith('line')
    ny, nx = int(rng.integers(12, 44)), int(rng.integers(12, 44))
    obj = rng.random((1, ny, nx)) + 1e-6
    n_orient = int(rng.integers(1, 4)) if line else 1
    pad = int(rng.integers(3, 12)) if n_orient == 1 else int(0.45 * max(ny, nx)) + int(rng.integers(0, 3))
    psf_width = float(rng.uniform(4.0, 11.0))
    R = float(rng.choice([1.0, 1.5, 2.0, rng.uniform(1.0, 3.0)]))
    psf_width = max(psf_width, 2.2 * R)                # (the scan step round(psf_width / (4 R)) must not be 0: fig3:102 would divide by it)
    pulses = int(rng.integers(1, 4))
    want, got = [], []
    f3.simulate_imaging(obj, imaging_type, psf_width, R, n_orient, pulses, pad, lambda rot, pos, *a: want.append((rot, pos, a)))
    device_simulate(fig3)(obj, imaging_type, psf_width, R, n_orient, pulses, pad, lambda rot, pos, *a: got.append((rot, pos, a)))
    case = (imaging_type, ny, nx, n_orient, pad, psf_width, R, pulses)
    assert [(int(r), p) for r, p, _ in got] == [(int(r), p) for r, p, _ in want], case
    for (_, _, a), (_, _, b) in zip(got, want):
        for x, y in zip(a[:7], b[:7]):
            assert np.abs(x - y).max() < 1e-10, case
        assert a[7:] == b[7:], case
