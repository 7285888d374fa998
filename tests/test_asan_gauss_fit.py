"""csrc/gauss_fit.cpp (MINPACK lmdif restated: get_width, line_sted_tools.py:653-668) built for the HOST and checked
against the widths the reference's curve_fit returned (G2).  tools/asan_emu.sh runs this file with an
AddressSanitizer / UBSan build of the same source (RLSTED_GAUSSFIT_LIB); without it a plain g++ build is used, so the
product's fit code is covered on the GPU-less box either way."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def fit_lib():
    so = os.environ.get('RLSTED_GAUSSFIT_LIB')
    if not so:
        out = os.path.join(ROOT, 'build', 'host')
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, 'libgaussfit.so')
        srcs = [os.path.join(ROOT, 'rescan_line_sted_amd', 'csrc', 'gauss_fit.cpp'), os.path.join(ROOT, 'tools', 'asan_gauss_fit_main.cpp')]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'include')]
                                  + srcs + ['-o', so])
    lib = ctypes.CDLL(so)
    lib.host_gauss_fit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib


def test_host_build_of_the_gauss_fit_matches_the_reference_widths(fit_lib, golden):
    g = golden('g2_get_width')
    for row, (n, w) in zip(g['rows'], g['n_and_width']):
        n = int(n)
        y = np.ascontiguousarray(row[:n], dtype=np.float64)
        p = np.zeros(3)
        fit_lib.host_gauss_fit(y.ctypes.data_as(ctypes.c_void_p), n, p.ctypes.data_as(ctypes.c_void_p))
        assert abs(p[2]) == pytest.approx(abs(w), rel=1e-8)


def test_degenerate_rows_do_not_read_out_of_bounds(fit_lib):
    """All-zero, constant, single-spike and length-3 rows: the fit may fail to converge, it must not misbehave."""
    for y in (np.zeros(17), np.ones(35), np.eye(1, 27, 13)[0], np.array([0.0, 1.0, 0.0]), np.full(129, 1e300)):
        y = np.ascontiguousarray(y, dtype=np.float64)
        p = np.zeros(3)
        fit_lib.host_gauss_fit(y.ctypes.data_as(ctypes.c_void_p), y.size, p.ctypes.data_as(ctypes.c_void_p))
