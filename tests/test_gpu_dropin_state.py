"""State semantics of the Deconvolver mirror where the reference's object is just a bag of attributes
(ADVICE r01): assigning to `estimate`, new data after some iterations, H / H_t on other shapes, and the
multi-view record_data -> load_data_from_tif round trip."""
import numpy as np
import pytest

from conftest import max_rel, fuzz_seeds
from oracle import line_sted_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def st():
    from rescan_line_sted_amd import _lib, line_sted_tools
    assert _lib.device_count() >= 1, 'no GPU visible'
    return line_sted_tools


@pytest.fixture(scope='module')
def views(golden):
    return [p[None] for p in golden('g8_fig2_psfs')['2p0x_lr/line_sted_psfs'][:2, 0]]


@pytest.fixture(scope='module')
def obj(golden):
    return golden('objects')['rings'].astype(np.float64)


def test_assigning_the_estimate_continues_from_it(st, views, obj, tmp_path):
    d = st.Deconvolver(views, str(tmp_path) + '/', verbose=False)
    d.create_data_from_object(obj, 5e10, random_seed=3)
    o = orc.Deconvolver(views)
    o.create_data_from_object(obj, 5e10, noisy_measurement=d.noisy_measurement)
    d.estimate = 7 * np.ones_like(obj)      # before the first iteration: overwritten by ones (ref:521-522)
    for _ in range(2):
        d.iterate()
        o.iterate()
    assert max_rel(d.estimate, o.estimate) < 1e-10
    d.estimate = 2 * d.estimate
    o.estimate = 2 * o.estimate
    d.iterate()
    o.iterate()
    assert max_rel(d.estimate, o.estimate) < 1e-10


def test_new_data_keeps_the_estimate(st, views, obj, tmp_path):
    d = st.Deconvolver(views, str(tmp_path) + '/', verbose=False)
    d.create_data_from_object(obj, 5e10, random_seed=3)
    o = orc.Deconvolver(views)
    o.create_data_from_object(obj, 5e10, noisy_measurement=d.noisy_measurement)
    for _ in range(3):
        d.iterate()
        o.iterate()
    d.create_data_from_object(np.flip(obj, 1).copy(), 5e10, random_seed=4)      # ref:496-512 does not touch estimate
    o.create_data_from_object(np.flip(obj, 1).copy(), 5e10, noisy_measurement=d.noisy_measurement)
    d.iterate()
    o.iterate()
    assert d.num_iterations == 4
    assert max_rel(d.estimate, o.estimate) < 1e-10


def test_operators_on_other_shapes_leave_the_state_alone(st, views, obj, tmp_path):
    a = st.Deconvolver(views, str(tmp_path) + '/', verbose=False)
    b = st.Deconvolver(views, str(tmp_path) + '/', verbose=False)
    for d in (a, b):
        d.create_data_from_object(obj, 5e10, random_seed=5)
        d.iterate()
    x = np.random.default_rng(0).random((2, 40, 48))
    ref = orc.Deconvolver(views)
    h = a.H(x)                                   # another shape: a plan of its own
    assert all(max_rel(h[v], ref.H(x)[v]) < 1e-12 for v in range(2))
    assert max_rel(a.H_t(h), ref.H_t(ref.H(x))) < 1e-12
    a.iterate()
    b.iterate()
    assert np.array_equal(a.estimate, b.estimate)


def test_multi_view_record_and_load_round_trip(st, views, obj, tmp_path):
    prefix = str(tmp_path) + '/'
    d = st.Deconvolver(views, prefix, verbose=False)
    d.create_data_from_object(obj, 5e10, random_seed=6)
    d.record_data()                              # noisy_measurement.tif: the two views stacked on axis 0 (ref:562-564)
    for _ in range(3):
        d.iterate()
    e = st.Deconvolver(views, prefix, verbose=False)
    e.load_data_from_tif(prefix + 'noisy_measurement.tif')
    for _ in range(3):
        e.iterate()
    assert max_rel(e.estimate, d.estimate) < 1e-6      # the TIF stores float32


def test_estimate_is_one_array_updated_in_place(st, views, obj, tmp_path):
    """ref:531 `self.estimate *= ...`: whoever holds d.estimate sees the next iteration in it."""
    d = st.Deconvolver(views, str(tmp_path) + '/', verbose=False)
    d.create_data_from_object(obj, 5e10, random_seed=3)
    d.iterate()
    held = d.estimate
    first = held.copy()
    d.iterate()
    assert d.estimate is held and not np.array_equal(held, first)
    d.num_iterations = 0                 # the reference starts over with a NEW array of ones (ref:521-522)
    d.iterate()
    assert d.estimate is not held and max_rel(d.estimate, first) < 1e-12


def test_pinned_host_arrays_and_out_argument(st, views, obj):
    from rescan_line_sted_amd import _lib
    plan = _lib.DeconvPlan(views, obj.shape[0], obj.shape[1], obj.shape[2], dtype='f64')
    frames = _lib.pinned_empty(obj.shape)
    frames[:] = obj
    plan.set_object(frames, 5e10)
    plan.simulate(seed=1)
    plan.iterate(3)
    out = _lib.pinned_empty(obj.shape)
    got = plan.estimate(out=out)
    assert got is out
    assert np.array_equal(got, plan.estimate())
    with pytest.raises(ValueError):
        plan.estimate(out=np.empty(3))
    del frames, out, got                 # the blocks are released with their last views


@pytest.mark.parametrize('seed', fuzz_seeds(6))
def test_random_method_sequences_match_the_reference_class(st, seed, tmp_path):
    """The mirror class and the oracle's restatement of the reference class driven through the same random
    sequence of calls (new data after iterations, estimate assignment, H / H_t on the data's and on other
    shapes, restarts through num_iterations = 0) hold the same estimate throughout."""
    rng = np.random.default_rng(500 + seed)
    V = int(rng.integers(1, 4))
    psfs = [(rng.random((1, int(rng.integers(1, 8)) * 2 + 1, 9)) + 0.05) for _ in range(V)]
    psfs = [p / p.sum() for p in psfs]
    shape = (int(rng.integers(1, 3)), int(rng.integers(8, 40)), int(rng.integers(8, 48)))
    d = st.Deconvolver(psfs, str(tmp_path) + '/', verbose=False)
    o = orc.Deconvolver(psfs)
    have_data = False
    log = []
    for step in range(16):
        op = str(rng.choice(['data', 'iterate', 'iterate', 'iterate', 'H', 'H_other', 'H_t', 'assign', 'restart']))
        if not have_data and op not in ('data', 'H', 'H_other'):
            op = 'data'
        log.append(op)
        if op == 'data':
            obj = rng.random(shape) * 10
            s = int(rng.integers(0, 1000))
            d.create_data_from_object(obj, 2e5, random_seed=s)
            o.create_data_from_object(obj, 2e5, noisy_measurement=d.noisy_measurement)
            assert all(max_rel(a, b) < 1e-12 for a, b in zip(d.noiseless_measurement, o.noiseless_measurement)), log
            have_data = True
        elif op == 'iterate':
            d.iterate()
            o.iterate()
            assert d.num_iterations == o.num_iterations
        elif op in ('H', 'H_other'):
            sh = shape if op == 'H' else (1, shape[1] + 3, shape[2] + 5)
            x = rng.random(sh)
            assert all(max_rel(a, b) < 1e-12 for a, b in zip(d.H(x), o.H(x))), log
        elif op == 'H_t':
            y = [rng.random(shape) + 0.1 for _ in range(V)]
            assert max_rel(d.H_t(y), o.H_t(y)) < 1e-12, log
        elif op == 'assign' and d.num_iterations > 0:
            e = rng.random(shape) + 0.5
            d.estimate = e
            o.estimate = e.copy()
        elif op == 'restart':
            d.num_iterations = 0
            o.num_iterations = 0
        if d.num_iterations > 0:
            assert max_rel(d.estimate, o.estimate) < 1e-10, log


def test_views_of_different_psf_shapes(st, tmp_path):
    """The reference convolves every view with its own PSF, whatever its shape (ref:573-576, 584-588: fftconvolve per view).
    The device plan has one (py, px) for all views: smaller PSFs are zero-embedded around their centre tap
    (_lib.common_psf_shape, every DeconvPlan), which changes nothing -- H, H_t, the normaliser and the iterations equal the oracle's
    per-view convolutions, for odd / even / 1-row shapes whose centres (p - 1) // 2 differ."""
    rng = np.random.default_rng(12)
    psfs = [rng.random((1, 9, 11)) + 0.01, rng.random((1, 6, 4)) + 0.01, rng.random((1, 1, 7)) + 0.01, rng.random((1, 8, 3)) + 0.01]
    x = rng.random((2, 40, 52)) * 30
    d = st.Deconvolver(psfs, str(tmp_path) + '/', verbose=False)
    o = orc.Deconvolver(psfs)
    assert [p.shape for p in d.psfs] == [p.shape for p in psfs]          # the public attribute keeps the caller's arrays
    for a, b in zip(d.H(x), o.H(x)):
        assert max_rel(a, b) < 1e-12
    y = [rng.random(x.shape) for _ in psfs]
    assert max_rel(d.H_t(y, normalize=False), o.H_t(y, normalize=False)) < 1e-12
    assert max_rel(d.H_t(y), o.H_t(y)) < 1e-12
    d.create_data_from_object(x, 1e7, random_seed=4)
    o.create_data_from_object(x, 1e7, noisy_measurement=d.noisy_measurement)
    for _ in range(6):
        d.iterate()
        o.iterate()
    assert max_rel(d.estimate, o.estimate) < 1e-10


def test_psfs_with_depth(st, tmp_path):
    """The reference's fftconvolve is n-dimensional (ref:574,586): out[z] = sum_k conv2d(x[z + c - k], psf[k]), c = (pz - 1) // 2.
    On single-slice data only the PSF's plane k = c meets the data -- the device plan takes that plane and the drop-in equals the
    oracle's 3-D convolutions (odd and even depths); data of several slices with such a PSF is not built and says so."""
    rng = np.random.default_rng(13)
    x = rng.random((1, 36, 44)) * 20
    for pz in (3, 4):
        psfs = [rng.random((pz, 7, 9)) + 0.01, rng.random((1, 5, 5)) + 0.01]
        d = st.Deconvolver(psfs, str(tmp_path) + '/', verbose=False)
        o = orc.Deconvolver(psfs)
        for a, b in zip(d.H(x), o.H(x)):
            assert max_rel(a, b) < 1e-12
        d.create_data_from_object(x, 1e6, random_seed=2)
        o.create_data_from_object(x, 1e6, noisy_measurement=d.noisy_measurement)
        for _ in range(5):
            d.iterate()
            o.iterate()
        assert max_rel(d.estimate, o.estimate) < 1e-10
    d = st.Deconvolver([rng.random((3, 5, 5))], str(tmp_path) + '/', verbose=False)
    with pytest.raises(NotImplementedError, match='couples the 2 z slices'):
        d.H(rng.random((2, 20, 20)))


def test_deconvolver_warns_when_predictions_are_unresolved(st, tmp_path):
    """Sparse emitters, a narrow PSF, float32 arithmetic: the mirror's estimate stays finite and the first fetch warns that the
    iterations met predictions the plan could not resolve (the reference's arithmetic: 1 / 0); the float64 default does not."""
    import warnings
    rng = np.random.default_rng(3)
    obj = np.zeros((1, 128, 128))
    obj[0, rng.integers(8, 120, 12), rng.integers(8, 120, 12)] = 1.0
    yy, xx = np.mgrid[-4:5, -4:5]
    u, w = 0.866 * xx + 0.5 * yy, -0.5 * xx + 0.866 * yy
    psf = np.exp(-0.5 * ((u / 1.6) ** 2 + (w / 0.8) ** 2))[None]
    for dtype, expect in (('f32', True), ('f64', False)):
        d = st.Deconvolver([psf], output_prefix=str(tmp_path / dtype) + '_', verbose=False, dtype=dtype)
        d.create_data_from_object(obj, total_brightness=2e3, random_seed=1)
        for _ in range(6):
            d.iterate()
        with warnings.catch_warnings(record=True) as w_:
            warnings.simplefilter('always')
            est = d.estimate
            again = d.estimate                               # (once per Deconvolver)
        assert np.isfinite(est).all() and est.max() > 0 and again is est
        hits = [x for x in w_ if 'H(estimate) <= 0' in str(x.message)]
        assert len(hits) == (1 if expect else 0), (dtype, [str(x.message) for x in w_])
