"""Device Poisson spec: Philox known-answer vectors, the numpy twin
(oracle/philox_poisson.py) against the host build of the HIP sampler header,
and distributional agreement with numpy's own Poisson sampler.  CPU only."""
import ctypes

import numpy as np
import pytest

from oracle import philox_poisson as pp
from test_emulated_kernels import emu  # noqa: F401  (fixture)


def test_philox_known_answers(emu):
    # Random123 kat_vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = pp.philox4x32_10(*[np.uint64(c) for c in ctr], key[0], key[1])
        assert tuple(int(g) for g in got) == want
        out = (ctypes.c_uint * 4)()
        emu.emu_philox(*ctr, *key, out)
        assert tuple(out) == want


@pytest.mark.parametrize('seed,image', [(0, 0), (12345678901234567, 7)])
def test_numpy_twin_matches_sampler_header_bit_for_bit(emu, seed, image):
    rng = np.random.default_rng(5)
    lam = np.concatenate([[0.0, 1e-300, 0.5, 9.99, 9.999999999, 10.0, 10.000001, 20.5, 1e3, 1e7, 3.7e9],
                          rng.random(3000) * 12,             # multiplication method / PTRS border
                          rng.random(3000) * 200,
                          10 ** (rng.random(3000) * 9)])     # up to 1e9
    out = np.empty_like(lam)
    emu.emu_poisson(lam.ctypes.data_as(ctypes.c_void_p), lam.size,
                    ctypes.c_ulonglong(seed), ctypes.c_uint(image),
                    out.ctypes.data_as(ctypes.c_void_p))
    twin = pp.poisson(lam, seed, image)
    assert np.array_equal(out, twin)
    assert np.all(out == np.floor(out)) and np.all(out >= 0)


def test_fast_first_attempt_is_a_prefix_of_the_full_sampler(emu):
    """The device draws noise in two launches (fast squeeze-accept for everyone, full
    sampler for the rest); a pixel finished by the fast kernel must get exactly the
    value the full sampler gives."""
    rng = np.random.default_rng(9)
    lam = np.concatenate([[0.0, 5.0, 10.0], 10 ** (1 + rng.random(20000) * 7)])
    full = np.empty_like(lam)
    fast = np.empty_like(lam)
    flags = np.zeros(lam.size, dtype=np.int32)
    vp = ctypes.c_void_p
    emu.emu_poisson(lam.ctypes.data_as(vp), lam.size, ctypes.c_ulonglong(77), ctypes.c_uint(2), full.ctypes.data_as(vp))
    emu.emu_poisson_fast(lam.ctypes.data_as(vp), lam.size, ctypes.c_ulonglong(77), ctypes.c_uint(2),
                         fast.ctypes.data_as(vp), flags.ctypes.data_as(vp))
    acc = flags.astype(bool)
    assert acc[0] and not acc[1]                    # lam = 0 finished, lam < 10 deferred
    assert 0.6 < acc[3:].mean() < 0.95              # the squeeze accepts most pixels (~79 % at large rates)
    assert acc[3:][lam[3:] > 1e4].mean() > 0.75     # P(us >= 0.07) * P(V <= vr) ~ 0.86 * 0.92
    assert np.array_equal(fast[acc], full[acc])


def test_det_functions_are_accurate():
    x = 10 ** np.linspace(-300, 300, 20001)
    assert np.max(np.abs(pp.det_log(x) - np.log(x)) / np.maximum(np.abs(np.log(x)), 1)) < 4e-16
    t = -np.linspace(0, 10, 10001)
    assert np.max(np.abs(pp.det_exp(t) / np.exp(t) - 1)) < 1e-15
    from scipy.special import gammaln
    k = np.concatenate([np.arange(0, 200.0), 10 ** np.linspace(2, 9, 500) // 1])
    assert np.max(np.abs(pp.det_logfact(k) - gammaln(k + 1)) / np.maximum(gammaln(k + 1), 1)) < 1e-14


@pytest.mark.parametrize('lam', [0.5, 3.0, 9.99, 10.0, 50.0, 1e3, 1e7])
def test_distribution_matches_poisson(lam):
    n = 200000
    draws = pp.poisson(np.full(n, lam), seed=2024, image=3)
    se_mean = np.sqrt(lam / n)
    assert abs(draws.mean() - lam) < 5 * se_mean
    # variance of a Poisson = lam; standard error of the sample variance
    se_var = np.sqrt((lam + 2 * lam * lam) / n) * 1.5
    assert abs(draws.var() - lam) < 5 * se_var
    if lam <= 50:   # chi-square against the exact pmf, and against numpy's sampler
        from scipy.stats import poisson as sp
        kmax = int(lam + 8 * np.sqrt(lam) + 8)
        obs = np.bincount(draws.astype(int), minlength=kmax + 1)[:kmax + 1]
        exp = sp.pmf(np.arange(kmax + 1), lam) * n
        keep = exp > 5
        chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
        assert chi2 < keep.sum() + 6 * np.sqrt(2 * keep.sum())
        ref = np.bincount(np.random.RandomState(1).poisson(lam, n), minlength=kmax + 1)[:kmax + 1]
        chi2n = ((ref[keep] - exp[keep]) ** 2 / exp[keep]).sum()
        assert chi2n < keep.sum() + 6 * np.sqrt(2 * keep.sum())   # sanity of the yardstick


def test_streams_are_independent_across_pixels_images_seeds():
    lam = np.full(4096, 100.0)
    a = pp.poisson(lam, 1, 0)
    assert np.array_equal(a, pp.poisson(lam, 1, 0))
    for b in (pp.poisson(lam, 2, 0), pp.poisson(lam, 1, 1)):
        assert abs(np.corrcoef(a, b)[0, 1]) < 0.08
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.08
