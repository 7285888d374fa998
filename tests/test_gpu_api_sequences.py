"""Random call sequences through the C ABI against a state model kept with the oracle: whatever order a caller
mixes set_measurement / iterate / forward / adjoint / set_estimate / reset in, the plan's estimate follows the
reference's arithmetic (line_sted_tools.py:520-531, 567-594).  Both strategies (FFT, separable), float64."""
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


def rl_step(views, meas, est):
    out = np.empty_like(est)
    for f in range(est.shape[0]):
        d = orc.Deconvolver(views)
        d.noisy_measurement = [meas[f, v][None] for v in range(len(views))]
        d.estimate = est[f][None].copy()
        d.num_iterations = 1                      # continue from d.estimate (ref:521 resets only at 0)
        d.iterate()
        out[f] = d.estimate[0]
    return out


@pytest.mark.parametrize('seed', fuzz_seeds(8))
@pytest.mark.parametrize('strategy', ['fft', 'separable'])
def test_random_call_sequences_follow_the_reference(lib, strategy, seed):
    rng = np.random.default_rng(1000 + seed)
    B, V = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    ny, nx = int(rng.integers(3, 40)), int(rng.integers(3, 60))
    if strategy == 'separable':
        views = [np.outer(rng.random(int(rng.integers(1, 6))) + 0.1, rng.random(5) + 0.1)[None] for _ in range(V)]
        views = [np.outer(rng.random(3) + 0.1, rng.random(5) + 0.1)[None] for _ in range(V)]
    else:
        views = [(rng.random((7, 9)) + 0.05)[None] for _ in range(V)]
    plan = lib.DeconvPlan(views, B, ny, nx, dtype='f64')
    assert plan.strategy()['separable'] == (strategy == 'separable')
    o = orc.Deconvolver(views)
    meas = rng.random((B, V, ny, nx)) * 20 + 0.5
    plan.set_measurement(meas)
    est = None
    log = []
    for step in range(14):
        op = rng.choice(['iterate', 'iterate', 'forward', 'adjoint', 'set_measurement', 'set_estimate', 'reset', 'check'])
        log.append(op)
        if op == 'iterate':
            k = int(rng.integers(1, 4))
            if est is None:
                est = np.ones((B, ny, nx))
            for _ in range(k):
                est = rl_step(views, meas, est)
            plan.iterate(k)
        elif op == 'forward':
            x = rng.random((B, ny, nx)) * 5
            h = plan.forward(x)
            for f in range(B):
                ref = o.H(x[f][None])
                assert max(max_rel(h[f, v], ref[v][0]) for v in range(V)) < 1e-12, log
        elif op == 'adjoint':
            y = rng.random((B, V, ny, nx)) + 0.1
            a = plan.adjoint(y, True)
            for f in range(B):
                assert max_rel(a[f], o.H_t([y[f, v][None] for v in range(V)])[0]) < 1e-12, log
        elif op == 'set_measurement':
            meas = rng.random((B, V, ny, nx)) * 20 + 0.5
            plan.set_measurement(meas)                       # new data starts a new run at the ABI (rlsted.h); the
            if est is not None and rng.integers(0, 2):       # reference's "keep the estimate" is read + set_estimate
                plan.set_estimate(est)
            else:
                est = None
        elif op == 'set_estimate':
            est = rng.random((B, ny, nx)) * 3 + 0.2
            plan.set_estimate(est)
        elif op == 'reset':
            plan.reset_estimate()
            est = None
        if est is not None and op in ('iterate', 'check', 'set_estimate'):
            assert max_rel(plan.estimate(), est) < 1e-10, log
    if est is None:
        est = np.ones((B, ny, nx))
        est = rl_step(views, meas, est)
        plan.iterate(1)
    assert max_rel(plan.estimate(), est) < 1e-10, log
    assert max_rel(plan.measurement(), meas) < 1e-15
