"""Index-logic check of the HIP workgroup bodies without a GPU.

tests/emu/emu.cpp compiles the very same templates the HIP kernels are made of
(rescan_line_sted_amd/csrc/conv_kernels.hpp) for the host and runs them with
one OS thread per GPU thread.  Here the emulated kernels are chained exactly as
the device plan chains them and compared with the CPU oracle.  CPU only.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import line_sted_oracle as orc
from conftest import max_rel, ROOT

EMU_DIR = os.path.join(ROOT, 'tests', 'emu')
ROW_FWD, ROW_INV, ROW_RATIO, ROW_UPDATE, ROW_ADJ = range(5)


@pytest.fixture(scope='module')
def emu():
    if os.environ.get('RLSTED_EMU_LIB'):      # tools/asan_emu.sh: the AddressSanitizer / UBSan build of the same source
        return ctypes.CDLL(os.environ['RLSTED_EMU_LIB'])
    so = os.path.join(EMU_DIR, 'libemu.so')
    src = os.path.join(EMU_DIR, 'emu.cpp')
    deps = [src] + [os.path.join(ROOT, 'rescan_line_sted_amd', 'csrc', f)
                    for f in ('conv_kernels.hpp', 'fft_core.hpp', 'fft_configs.hpp', 'philox_poisson.hpp')]
    if (not os.path.exists(so) or
            os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps)):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                               '-Wno-unknown-pragmas', '-pthread', src, '-o', so])
    return ctypes.CDLL(so)


def _slack(a):
    """Copy of `a` that is followed in memory by 16 KiB of slack: the lean row bodies load
    whole 64-lane segments, so lanes past the end of the last row read (and discard) bytes
    behind the array -- the device plan allocates RL_STREAM_SLACK behind every buffer for this."""
    a = np.ascontiguousarray(a)
    raw = np.zeros(a.nbytes + 16384, dtype=np.uint8)
    out = raw[:a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class EmuPlan:
    """Mirrors the device plan: buffers + kernel sequence, f64 or f32."""

    def __init__(self, lib, psfs, ny, nx, Ly, Lx, dtype=np.float64):
        self.lib, self.ny, self.nx, self.Ly, self.Lx = lib, ny, nx, Ly, Lx
        self.rt = np.dtype(dtype)
        self.ct = np.complex128 if self.rt == np.float64 else np.complex64
        self.sfx = 'f64' if self.rt == np.float64 else 'f32'
        self.V = len(psfs)
        self.kx = Lx // 2 + 1
        self.pitch = (self.kx + 7) // 8 * 8
        ph = np.zeros((self.V, Ly, self.pitch), dtype=self.ct)
        for v, p in enumerate(psfs):
            Py, Px = p.shape[1:]
            cy, cx = (Py - 1) // 2, (Px - 1) // 2
            assert Ly >= ny + max(cy, Py - 1 - cy) and Lx >= nx + max(cx, Px - 1 - cx)
            w = np.zeros((Ly, Lx))
            yy = (np.arange(Py) - cy) % Ly
            xx = (np.arange(Px) - cx) % Lx
            w[np.ix_(yy, xx)] = p[0]
            ph[v, :, :self.kx] = np.fft.rfft2(w) / (Ly * Lx)
        self.psf_hat_natural = ph
        t, c, q = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.transposed = lib.emu_geometry(Ly, ctypes.byref(t), ctypes.byref(c), ctypes.byref(q))
        assert self.transposed in (0, 1)
        if self.transposed:      # [view][kx][Ly], as the device plan stores it for wave-private lengths
            pht = np.zeros((self.V, Ly, self.pitch), dtype=self.ct).reshape(-1)
            pht[:self.V * self.kx * Ly] = np.ascontiguousarray(ph[:, :, :self.kx].transpose(0, 2, 1)).reshape(-1)
            self.psf_hat = pht
        else:
            self.psf_hat = ph

    # Spectra live in device memory row-major (conv_kernels.hpp spec_off); the kernels' unclamped loads need slack behind them.
    def _to_blocked(self, nat):
        return _slack(nat)

    def _to_natural(self, blk):
        return blk

    def _spectra(self, spec_in, spec_out):
        bin_ = self._to_blocked(spec_in) if spec_in is not None else None
        bout = bin_ if (spec_out is spec_in and spec_in is not None) else (
            self._to_blocked(spec_out) if spec_out is not None else None)
        return bin_, bout

    def row(self, mode, gy, spec_in=None, spec_out=None, src=None, dst=None,
            norm=None, scale=None):
        nat_out = spec_out
        spec_in, spec_out = self._spectra(spec_in, spec_out)
        try:
            self._row(mode, gy, spec_in, spec_out, src, dst, norm, scale)
        finally:
            if nat_out is not None:
                nat_out[...] = self._to_natural(spec_out)

    def _row(self, mode, gy, spec_in, spec_out, src, dst, norm, scale):
        f = getattr(self.lib, 'emu_row_' + self.sfx)
        rc = f(self.Lx, mode, _p(spec_in) if spec_in is not None else None,
               _p(spec_out) if spec_out is not None else None,
               _p(src) if src is not None else None,
               _p(dst) if dst is not None else None,
               _p(norm) if norm is not None else None,
               _p(scale) if scale is not None else None,
               self.ny, self.nx, self.pitch, self.V, gy)
        assert rc == 0

    def col(self, spec_in, spec_out, frames, kind):
        nat_out = spec_out
        spec_in, spec_out = self._spectra(spec_in, spec_out)
        try:
            self._col(spec_in, spec_out, frames, kind)
        finally:
            nat_out[...] = self._to_natural(spec_out)

    def _col(self, spec_in, spec_out, frames, kind):
        """kind: 'H' (frame spectrum -> V images), 'HT' (per view), 'HT_SUM' (views summed in
        the Fourier domain -> one image per frame).  Same mode selection as rlsted.cpp col_t()."""
        if kind is True:
            kind = 'H'
        elif kind is False:
            kind = 'HT'
        f = getattr(self.lib, 'emu_col_' + self.sfx)
        mode = 0
        if self.V > 1 and self.transposed and kind != 'HT':
            mode = 1 if kind == 'H' else 2
        assert not (kind == 'HT_SUM' and mode == 0 and self.V > 1)
        rc = f(self.Ly, _p(spec_in), _p(spec_out), _p(self.psf_hat), self.ny,
               self.kx, self.pitch, self.V, frames,
               1 if kind == 'H' else self.V, 0 if kind == 'H' else 1, mode)
        assert rc == 0

    def spec(self, n):
        # NaN-poisoned so that any read of a never-written element shows up
        return _slack(np.full((n, self.ny, self.pitch), np.nan + 1j * np.nan, dtype=self.ct))

    def H(self, x):
        B = x.shape[0]
        x = np.ascontiguousarray(x, dtype=self.rt)
        sa, sb = self.spec(B), self.spec(B * self.V)
        self.row(ROW_FWD, B, spec_out=sa, src=x)
        self.col(sa, sb, B, True)
        out = np.full((B * self.V, self.ny, self.nx), np.nan, dtype=self.rt)
        self.row(ROW_INV, B * self.V, spec_in=sb, dst=out)
        return out.reshape(B, self.V, self.ny, self.nx), sa

    def normalization(self):
        ones = np.ones((1, self.ny, self.nx), dtype=self.rt)
        sa, sb = self.spec(1), self.spec(self.V)
        self.row(ROW_FWD, 1, spec_out=sa, src=ones)
        self.col(sa, sb, 1, True)
        norm = np.full((self.ny, self.nx), np.nan, dtype=self.rt)
        self.row(ROW_ADJ, 1, spec_in=sb, dst=norm)
        return norm

    def rl(self, meas, K, fuse=False):
        B = meas.shape[0]
        meas = _slack(np.ascontiguousarray(meas.reshape(B * self.V, self.ny, self.nx), dtype=self.rt))
        norm = _slack(self.normalization())
        est = _slack(np.ones((B, self.ny, self.nx), dtype=self.rt))
        sa, sb = self.spec(B), self.spec(B * self.V)
        self.row(ROW_FWD, B, spec_out=sa, src=est)
        for _ in range(K):
            self.col(sa, sb, B, True)
            self.row(ROW_RATIO, B * self.V, spec_in=sb, spec_out=sb, src=meas)
            if fuse and self.V > 1:
                self.col(sb, sa, B, 'HT_SUM')
                keepV, self.V = self.V, 1          # the UPDATE row pass sees one (summed) view
                self.row(ROW_UPDATE, B, spec_in=sa, spec_out=sa, dst=est, norm=norm)
                self.V = keepV
            else:
                self.col(sb, sb, B, False)
                self.row(ROW_UPDATE, B, spec_in=sb, spec_out=sa, dst=est, norm=norm)
        return est, norm


def test_row_forward_is_rfft_of_padded_rows(emu):
    rng = np.random.default_rng(1)
    for (ny, nx, L) in ((6, 40, 64), (5, 33, 64), (4, 130, 192)):
        pl = EmuPlan(emu, [np.ones((1, 1, 1))], ny, nx, 64, L)
        x = rng.random((2, ny, nx))
        sc = np.array([2.0, 0.5])
        out = pl.spec(2)
        pl.row(ROW_FWD, 2, spec_out=out, src=x, scale=sc)
        ref = np.fft.rfft(x * sc[:, None, None], n=L, axis=2)
        assert max_rel(out[:, :, :pl.kx], ref) < 1e-14


@pytest.mark.parametrize('L,nx', [(64, 40), (192, 130), (576, 512)])
@pytest.mark.parametrize('sub_one', [0, 1])
def test_ratio_is_neutral_where_the_prediction_is_not_positive(emu, L, nx, sub_one):
    """conv_kernels.hpp rl_ratio: the reference clamps H(est) at 0 and divides (ref:575, 524) -- inf, then NaN through the next
    convolution.  The kernels make such a pixel neutral: ratio 1 (residual 0 with `ratio - 1`).  ROW_RATIO on rows whose
    prediction is positive, zero and negative by turns, against numpy with the same rule; every value finite."""
    emu.emu_set_sub_one.argtypes = [ctypes.c_int]
    ny = 6
    pl = EmuPlan(emu, [np.ones((1, 1, 1))], ny, nx, 64, L)
    rng = np.random.default_rng(L + sub_one)
    pred = rng.random((1, ny, nx)) + 0.5
    pred[0, :, ::3] = -0.125                    # rounding noise below zero (an exact 0 comes back from the transform as noise of either
    pred[0, :, 1::7] = -0.25                    # sign, ~1e-16: then meas / noise ~ 1e15 is what the rule -- and the reference -- give)
    meas = rng.random((1, ny, nx)) * 5 + 1e-9
    sin = pl.spec(1)
    sin[...] = 0
    sin[:, :, :pl.kx] = np.fft.rfft(pred, n=L, axis=2) / L         # the inverse row pass is unnormalised
    out = pl.spec(1)
    try:
        emu.emu_set_sub_one(sub_one)
        pl.row(ROW_RATIO, 1, spec_in=sin, spec_out=out, src=_slack(meas))
    finally:
        emu.emu_set_sub_one(0)
    ok = pred > 0
    ratio = np.where(ok, (meas - pred if sub_one else meas) / np.where(ok, pred, 1.0), 0.0 if sub_one else 1.0)
    back = np.fft.irfft(out[:, :, :pl.kx], n=L, axis=2)[:, :, :nx]
    assert np.isfinite(out[:, :, :pl.kx]).all()
    assert max_rel(back, ratio) < 1e-13


def test_column_pass_is_circular_convolution_along_y(emu):
    rng = np.random.default_rng(2)
    for Ly, ny in ((64, 37), (192, 130), (256, 161)):
        pl = EmuPlan(emu, [rng.random((1, 5, 3)), rng.random((1, 4, 6))], ny, 8, Ly, 64)
        B = 2
        sin = pl.spec(B)
        sin[:, :, :pl.kx] = rng.random((B, ny, pl.kx)) + 1j * rng.random((B, ny, pl.kx))
        sout = pl.spec(B * pl.V)
        pl.col(sin, sout, B, True)
        for b in range(B):
            for v in range(pl.V):
                pad = np.zeros((Ly, pl.kx), dtype=complex)
                pad[:ny] = sin[b, :, :pl.kx]
                ref = np.fft.ifft(np.fft.fft(pad, axis=0) * pl.psf_hat_natural[v, :, :pl.kx], axis=0)[:ny] * Ly
                assert max_rel(sout[b * pl.V + v, :, :pl.kx], ref) < 1e-13
        # H_t indexing, in place
        s2 = pl.spec(B * pl.V)
        s2[:, :, :pl.kx] = rng.random((B * pl.V, ny, pl.kx)) + 0j
        keep = s2.copy()
        pl.col(s2, s2, B, False)
        for i in range(B * pl.V):
            pad = np.zeros((Ly, pl.kx), dtype=complex)
            pad[:ny] = keep[i, :, :pl.kx]
            ref = np.fft.ifft(np.fft.fft(pad, axis=0) * pl.psf_hat_natural[i % pl.V, :, :pl.kx], axis=0)[:ny] * Ly
            assert max_rel(s2[i, :, :pl.kx], ref) < 1e-13


@pytest.mark.parametrize('ny,nx,Ly,Lx,pshapes', [
    (40, 48, 64, 64, [(1, 9, 11)]),
    (33, 31, 64, 64, [(1, 7, 7), (1, 8, 10), (1, 1, 7)]),    # odd sizes, even PSF, 3 views
    (128, 128, 192, 192, [(1, 107, 107)]),
    (24, 530, 64, 576, [(1, 5, 47)]),        # the 576 row transforms (wave private, radix 9-8-8)
    (530, 6, 576, 64, [(1, 47, 3)]),         # the 576 column transforms
])
def test_forward_model_matches_oracle(emu, ny, nx, Ly, Lx, pshapes):
    rng = np.random.default_rng(3)
    psfs = [rng.random(s) for s in pshapes]
    pl = EmuPlan(emu, psfs, ny, nx, Ly, Lx)
    x = rng.random((2, ny, nx))
    got, _ = pl.H(x)
    d = orc.Deconvolver(psfs)
    ref = d.H(x)
    for v in range(len(psfs)):
        assert max_rel(got[:, v], ref[v]) < 1e-13
    norm = pl.normalization()
    assert max_rel(norm, d.H_t([np.ones((1, ny, nx))] * len(psfs), normalize=False)[0]) < 1e-13


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-11), (np.float32, 2e-5)])
def test_richardson_lucy_matches_oracle(emu, dtype, tol):
    rng = np.random.default_rng(4)
    ny, nx = 33, 31
    psfs = [rng.random((1, 7, 7)), rng.random((1, 5, 9))]
    obj = rng.random((1, ny, nx)) * 50
    d = orc.Deconvolver(psfs)
    d.create_data_from_object(obj, random_seed=0)
    for _ in range(4):
        d.iterate()
    pl = EmuPlan(emu, psfs, ny, nx, 64, 64, dtype)
    meas = np.array(d.noisy_measurement)[None, :, 0]       # (B=1, V, ny, nx)
    est, _ = pl.rl(meas, 4)
    assert max_rel(est[0], d.estimate[0]) < tol


def test_multi_view_column_modes_on_wave_private_length(emu):
    """Ly = 256 (wave private): H with one shared forward transform, H_t summed in the
    Fourier domain, and the fused RL iteration, against the oracle."""
    rng = np.random.default_rng(6)
    ny, nx = 150, 40
    psfs = [rng.random((1, 9, 5)), rng.random((1, 7, 7)), rng.random((1, 11, 3))]
    pl = EmuPlan(emu, psfs, ny, nx, 256, 64)
    assert pl.transposed == 1
    x = rng.random((2, ny, nx))
    got, _ = pl.H(x)                                        # COL_H_MULTI
    d = orc.Deconvolver(psfs)
    ref = d.H(x)
    for v in range(3):
        assert max_rel(got[:, v], ref[v]) < 1e-13
    obj = rng.random((1, ny, nx)) * 30
    d.create_data_from_object(obj, random_seed=0)
    for _ in range(3):
        d.iterate()
    meas = np.array(d.noisy_measurement)[None, :, 0]
    est_f, _ = pl.rl(meas, 3, fuse=True)                    # COL_HT_SUM + single-view UPDATE
    est_u, _ = pl.rl(meas, 3, fuse=False)
    assert max_rel(est_u[0], d.estimate[0]) < 1e-11
    assert max_rel(est_f[0], d.estimate[0]) < 1e-11         # no negative lobes here: clamp of sum == sum of clamps


def test_golden_rl_through_emulated_kernels(emu, golden):
    """rings 128x128, the real 107x107 fig-2 point PSF, 2 RL iterations (f64)."""
    g, psfs, objs = golden('g5_rl'), golden('g8_fig2_psfs'), golden('objects')
    psf = [psfs['1p5x_lr/point_sted_psf'][0]]
    pl = EmuPlan(emu, psf, 128, 128, 192, 192)
    obj = objs['rings'].astype(np.float64)
    obj = obj * (5e10 / obj.sum())
    got, _ = pl.H(obj)
    assert max_rel(got[0, 0], g['rings_point_1p5x/noiseless'][0, 0]) < 1e-13
    est, norm = pl.rl(g['rings_point_1p5x/noisy'][None, :, 0], 2)
    assert max_rel(norm, g['rings_point_1p5x/norm'][0]) < 1e-13
    ref = g['rings_point_1p5x/estimate_2'][0]
    assert (np.abs(est[0] - ref) / np.abs(ref)).max() < 1e-9


@pytest.mark.parametrize('Ly,Lx,V,fuse', [(256, 256, 1, False), (64, 192, 2, False), (256, 256, 3, True), (64, 64, 2, False),
                                          (64, 64, 5, False)])
def test_ratio_minus_one_iteration(emu, Ly, Lx, V, fuse):
    """RowParams::sub_one (the f32 plans' default, conv_kernels.hpp rl_ratio): ROW_RATIO stores rowFFT(ratio - 1), ROW_UPDATE
    multiplies by max(1 + sum_v conv(ratio_v - 1, p_v) / norm, 0).  In float64 that is the plain iteration to rounding,
    through the lean single-view bodies (256), the multi-view update (192 / 64: rowpass_body PRESUM adds the views' spectra on
    their way in -- three at a time, so 2 and 5 views -- and clamps the sum after the one inverse transform) and the
    Fourier-domain view sum."""
    emu.emu_set_sub_one.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(11 + V)
    ny, nx = 21, 30
    psfs = [rng.random((1, 9, 7)) + 0.01 for _ in range(V)]            # non-negative: H_t(ones) is the normaliser
    obj = rng.random((2, ny, nx)) * 40
    d = orc.Deconvolver(psfs)
    d.create_data_from_object(obj, random_seed=0)
    K = 2 if Ly == 256 else 3          # (the 256 x 256 emulation costs ~15 s per iteration and frame pair)
    for _ in range(K):
        d.iterate()
    meas = np.array(d.noisy_measurement).transpose(1, 0, 2, 3)                          # (B=2, V, ny, nx)
    pl = EmuPlan(emu, psfs, ny, nx, Ly, Lx)
    try:
        emu.emu_set_sub_one(0)
        plain, _ = pl.rl(meas, K, fuse=fuse)
        emu.emu_set_sub_one(1)
        sub, _ = pl.rl(meas, K, fuse=fuse)
    finally:
        emu.emu_set_sub_one(0)
    assert max_rel(sub, plain) < 1e-12
    assert max_rel(sub, d.estimate) < 1e-11


def test_ratio_minus_one_l576_f32(emu):
    """The headline geometry (576 = 9*8*8 with cross-lane tails), f32, lean bodies, with and without `ratio - 1`."""
    emu.emu_set_sub_one.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(12)
    ny, nx = 20, 530
    psfs = [rng.random((1, 5, 41))]
    obj = rng.random((2, ny, nx)) * 40
    d = orc.Deconvolver(psfs)
    d.create_data_from_object(obj, random_seed=0)
    for _ in range(2):
        d.iterate()
    meas = np.array(d.noisy_measurement)[0][:, None]                                    # (B=2, V=1, ny, nx)
    pl = EmuPlan(emu, psfs, ny, nx, 64, 576, np.float32)
    try:
        emu.emu_set_sub_one(0)
        plain, _ = pl.rl(meas, 2)
        emu.emu_set_sub_one(1)
        sub, _ = pl.rl(meas, 2)
    finally:
        emu.emu_set_sub_one(0)
    assert max_rel(plain, d.estimate) < 2e-5 and max_rel(sub, d.estimate) < 2e-5


# ------------------------------------------------- long column transforms on the wave-private core
@pytest.mark.parametrize('Li,M,ny,kx,real_psf', [(256, 4, 900, 11, 0), (256, 2, 437, 8, 1), (576, 4, 2048, 9, 0),
                                                   (576, 4, 2001, 3, 1), (256, 8, 1900, 5, 0), (576, 8, 4096, 2, 1),
                                                   (256, 16, 1000, 21, 1)])      # (M = 16: M = 4 on 16-column tiles)
@pytest.mark.parametrize('park', [0, 1, 2])
def test_outer_decimation_column_pass(emu, Li, M, ny, kx, real_psf, park):
    """colconv_outer_body: L = M * Li as M core transforms plus one radix-M step in registers (the f32
    column kernel of L = 1152 = 2 x 576, 2304 = 4 x 576 and 4608 = 8 x 576).  Against numpy: IFFT_y(FFT_y(x zero padded to L) * psf_hat),
    rows < ny.  park: some of the waiting core results per lane wait in LDS instead of registers (PARK: 3 of 4 x 10 used in place
    during the radix-4 steps, 14 of 8 x 10 brought back for them; park = 2: the float64 kernels' 10 and 24) and the twiddles are read from an LDS copy -- the same values either way."""
    emu.emu_set_park.argtypes = [ctypes.c_int]
    L, V, frames = (4 if M == 16 else M) * Li, 2, 1
    pitch = (kx + 7) // 8 * 8
    rng = np.random.default_rng(Li + M + ny)
    x = np.zeros((frames, ny, pitch), dtype=np.complex128)
    x[:, :, :kx] = rng.standard_normal((frames, ny, kx)) + 1j * rng.standard_normal((frames, ny, kx))
    ph = rng.standard_normal((V, kx, L)) + (0 if real_psf else 1j) * rng.standard_normal((V, kx, L))
    out = np.zeros((frames * V, ny, pitch), dtype=np.complex128)
    psf_arg = np.ascontiguousarray(ph.real if real_psf else ph.astype(np.complex128))
    try:
        emu.emu_set_park(park)
        rc = emu.emu_col_outer_f64(Li, M, _p(_slack(x)), _p(out), _p(psf_arg), real_psf, ny, kx, pitch, V, frames, 1, 0)
    finally:
        emu.emu_set_park(0)
    assert rc == 0
    full = np.zeros((frames, L, kx), dtype=np.complex128)
    full[:, :ny] = x[:, :, :kx]
    spec = np.fft.fft(full, axis=1)                                   # (frames, L, kx)
    for f in range(frames):
        for v in range(V):
            ref = np.fft.ifft(spec[f] * ph[v].T, axis=0)[:ny] * L     # the kernels leave the 1/L to psf_hat's scale
            assert max_rel(out[f * V + v][:, :kx], ref) < 1e-12, (f, v)


@pytest.mark.parametrize('Li,M,ny,kx,real_psf,sum_views', [(256, 4, 900, 11, 0, 0), (256, 4, 901, 9, 1, 1), (576, 4, 2048, 9, 1, 1),
                                                             (576, 4, 2001, 3, 0, 0), (256, 8, 1900, 5, 1, 1), (256, 2, 437, 8, 0, 1)])
def test_split_column_pass(emu, Li, M, ny, kx, real_psf, sum_views):
    """The SPLIT column pass of colconv_outer_body (round 3): COL_SPLIT_FWD parks the column spectra in register-slot order,
    COL_SPLIT_INV multiplies one view's in and transforms back (H: the frame's spectrum is transformed once for its V views),
    COL_SPLIT_INV_SUM sums the V views' products first (H_t: one inverse transform per frame).  Against numpy; the parked
    spectra start as NaN, so every slot that is read was written."""
    L, V, frames = M * Li, 3, 2
    pitch = (kx + 7) // 8 * 8
    rng = np.random.default_rng(Li + M + ny + sum_views)
    n_in = frames * V if sum_views else frames
    n_out = frames if sum_views else frames * V
    x = np.zeros((n_in, ny, pitch), dtype=np.complex128)
    x[:, :, :kx] = rng.standard_normal((n_in, ny, kx)) + 1j * rng.standard_normal((n_in, ny, kx))
    ph = rng.standard_normal((V, kx, L)) + (0 if real_psf else 1j) * rng.standard_normal((V, kx, L))
    out = np.full((n_out, ny, pitch), np.nan + 0j, dtype=np.complex128)
    psf_arg = np.ascontiguousarray(ph.real if real_psf else ph.astype(np.complex128))
    rc = emu.emu_col_outer_split_f64(Li, M, _p(_slack(x)), _p(out), _p(psf_arg), real_psf, ny, kx, pitch, V, frames, sum_views)
    assert rc == 0
    full = np.zeros((n_in, L, kx), dtype=np.complex128)
    full[:, :ny] = x[:, :, :kx]
    spec = np.fft.fft(full, axis=1)
    for f in range(frames):
        if sum_views:
            ref = np.fft.ifft(sum(spec[f * V + v] * ph[v].T for v in range(V)), axis=0)[:ny] * L
            assert max_rel(out[f][:, :kx], ref) < 1e-12, f
        else:
            for v in range(V):
                ref = np.fft.ifft(spec[f] * ph[v].T, axis=0)[:ny] * L
                assert max_rel(out[f * V + v][:, :kx], ref) < 1e-12, (f, v)


# ------------------------------------------------- frame pairs: two frames in one complex image
@pytest.mark.parametrize('L,ny,nx,frames,sfx', [(576, 5, 512, 4, 'f64'), (576, 3, 301, 3, 'f64'), (256, 6, 200, 2, 'f64'),
                                                 (576, 2, 512, 2, 'f32'), (256, 3, 97, 5, 'f32')])
def test_frame_pair_row_kernels(emu, L, ny, nx, frames, sfx):
    """rowpair_body: frames 2p / 2p+1 are the real / imaginary part of one complex row transform; the spectrum of a pair
    is [ny][L] complex.  ROW_FWD, ROW_RATIO and ROW_UPDATE against numpy (unnormalised transforms both ways, like
    the per-frame kernels); with an odd frame count the last pair's imaginary part is a PHANTOM copy of its real part's frame
    (it reads that frame's images and is never stored: no value depends on whether the partner exists)."""
    rt = np.float64 if sfx == 'f64' else np.float32
    ct = np.complex128 if sfx == 'f64' else np.complex64
    tol = 1e-12 if sfx == 'f64' else 3e-5
    rng = np.random.default_rng(L + ny + nx)
    pairs = (frames + 1) // 2
    f = getattr(emu, 'emu_row_pair_' + sfx)

    def padded(a, phantom=None):   # (frames, ny, nx) real -> (pairs, ny, L) complex; phantom: the odd batch's last imaginary part
        z = np.zeros((pairs, ny, L), dtype=np.complex128)
        for i in range(frames):
            if i % 2 == 0:
                z[i // 2, :, :nx] += a[i]
            else:
                z[i // 2, :, :nx] += 1j * a[i]
        if frames % 2:
            z[pairs - 1, :, :nx] += 1j * (a[frames - 1] if phantom is None else phantom)
        return z

    est = _slack((rng.random((frames, ny, nx)) + 0.5).astype(rt))
    spec = _slack(np.full((pairs, ny, L), np.nan + 1j * np.nan, dtype=ct))
    assert f(L, ROW_FWD, None, _p(spec), _p(est), None, None, ny, nx, frames, 0) == 0
    ref = np.fft.fft(padded(est.astype(np.float64)), axis=2)
    assert max_rel(spec, ref) < tol
    # ROW_RATIO: the spectrum of positive rows (H(estimate) > 0), ratio = measurement / z
    zpos = rng.random((frames, ny, nx)) * 3 + 0.2
    s_pos = _slack((np.fft.fft(padded(zpos), axis=2) / L).astype(ct))
    meas = _slack((rng.random((frames, ny, nx)) * 5 + 1).astype(rt))
    out = _slack(np.full((pairs, ny, L), np.nan + 1j * np.nan, dtype=ct))
    assert f(L, ROW_RATIO, _p(s_pos), _p(out), _p(meas), None, None, ny, nx, frames, 0) == 0
    assert max_rel(out, np.fft.fft(padded(meas.astype(np.float64) / zpos), axis=2)) < (1e-11 if sfx == 'f64' else 3e-5)
    # ROW_UPDATE on an arbitrary spectrum whose inverse has both signs (the clamp matters):
    # est *= max(z, 0) / norm, spectrum of the new estimate
    s_in = _slack((rng.standard_normal((pairs, ny, L)) + 1j * rng.standard_normal((pairs, ny, L))).astype(ct))
    z = np.fft.ifft(s_in.astype(np.complex128), axis=2) * L
    e = np.stack([(z[i // 2].real if i % 2 == 0 else z[i // 2].imag)[:, :nx] for i in range(frames)])
    norm = _slack((rng.random((ny, nx)) + 0.5).astype(rt))
    est0 = est.astype(np.float64).copy()
    assert f(L, ROW_UPDATE, _p(s_in), _p(out), None, _p(est), _p(norm), ny, nx, frames, 0) == 0
    want = est0 * np.maximum(e, 0) / norm.astype(np.float64)
    assert max_rel(est, want) < (1e-12 if sfx == 'f64' else 2e-6)
    # (the phantom partner of an odd batch's last frame: that frame's estimate times the factor the imaginary part carries)
    ph = est0[frames - 1] * np.maximum(z[pairs - 1].imag[:, :nx], 0) / norm.astype(np.float64) if frames % 2 else None
    assert max_rel(out, np.fft.fft(padded(want, ph), axis=2)) < tol
    # the same two modes on `ratio - 1` (RowParams::sub_one): the stored spectrum is rowFFT(ratio - 1), the update
    # factor max(1 + z / norm, 0)
    emu.emu_set_sub_one.argtypes = [ctypes.c_int]
    try:
        emu.emu_set_sub_one(1)
        assert f(L, ROW_RATIO, _p(s_pos), _p(out), _p(meas), None, None, ny, nx, frames, 0) == 0
        assert max_rel(out, np.fft.fft(padded(meas.astype(np.float64) / zpos - 1.0), axis=2)) < (1e-11 if sfx == 'f64' else 3e-5)
        est[...] = est0.astype(rt)
        assert f(L, ROW_UPDATE, _p(s_in), _p(out), None, _p(est), _p(norm), ny, nx, frames, 0) == 0
    finally:
        emu.emu_set_sub_one(0)
    want = est0 * np.maximum(1.0 + e / norm.astype(np.float64), 0)
    assert max_rel(est, want) < (1e-12 if sfx == 'f64' else 2e-6)
    ph = est0[frames - 1] * np.maximum(1.0 + z[pairs - 1].imag[:, :nx] / norm.astype(np.float64), 0) if frames % 2 else None
    assert max_rel(out, np.fft.fft(padded(want, ph), axis=2)) < tol


# ------------------------------------------------- bodies specialised for a compile-time image size (round 4)
def test_compile_time_size_bodies(emu):
    """colconv_wave_body<..., NYC, CT> and rowpair_body<..., NXC, SUBC> (the device instantiates them for 512 x 512 frames at
    L = 576) in their L = 256, 192-row / 192-pixel instantiation: identical to the generic bodies' results -- same arithmetic,
    the row / pixel tests folded at compile time (rows of the tile that do not exist written as constants, pad columns
    loaded and stored like the others, the phantom partner of an odd batch)."""
    L, n, frames = 256, 192, 3
    pairs = (frames + 1) // 2
    rng = np.random.default_rng(77)
    emu.emu_set_special.argtypes = [ctypes.c_int]
    emu.emu_set_sub_one.argtypes = [ctypes.c_int]
    f = emu.emu_row_pair_f64

    def run_rows(special):
        emu.emu_set_special(special)
        emu.emu_set_sub_one(1)
        try:
            r = np.random.default_rng(78)
            zpos = r.random((frames + 1, 7, n)) * 3 + 0.2
            z = np.zeros((pairs, 7, L), dtype=np.complex128)
            for i in range(frames + 1):
                z[i // 2, :, :n] += (1j if i % 2 else 1) * zpos[i]
            s_pos = _slack(np.fft.fft(z, axis=2) / L)
            meas = _slack(r.random((frames, 7, n)) * 5 + 1)
            out = _slack(np.full((pairs, 7, L), np.nan + 1j * np.nan))
            assert f(L, ROW_RATIO, _p(s_pos), _p(out), _p(meas), None, None, 7, n, frames, 0) == 0
            ratio_spec = out.copy()
            est = _slack(r.random((frames, 7, n)) + 0.5)
            norm = _slack(r.random((7, n)) + 0.5)
            s_in = _slack(r.standard_normal((pairs, 7, L)) + 1j * r.standard_normal((pairs, 7, L)))
            assert f(L, ROW_UPDATE, _p(s_in), _p(out), None, _p(est), _p(norm), 7, n, frames, 0) == 0
            return ratio_spec, est.copy(), out.copy()
        finally:
            emu.emu_set_special(0)
            emu.emu_set_sub_one(0)

    a, b = run_rows(0), run_rows(1)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # column pass: 192 rows of 256 (and 128: the two-columns-per-access tile I/O), one view, 33 spectrum columns (pitch 40: pad
    # columns beside the last tile's valid ones)
    for n in (192, 128):
        _check_special_column_pass(emu, rng, L, n)


def _check_special_column_pass(emu, rng, L, n):
    pl = EmuPlan(emu, [rng.random((1, 5, 3))], n, 36, L, 64)
    sin = pl.spec(2)
    sin[:, :, :pl.kx] = rng.random((2, n, pl.kx)) + 1j * rng.random((2, n, pl.kx))
    outs = []
    for special in (0, 1):
        emu.emu_set_special(special)
        try:
            sout = pl.spec(2)
            pl.col(sin, sout, 2, True)
            outs.append(sout[:, :, :pl.kx].copy())
        finally:
            emu.emu_set_special(0)
    assert np.array_equal(outs[0], outs[1])
    for b_ in range(2):
        pad = np.zeros((L, pl.kx), dtype=complex)
        pad[:n] = sin[b_, :, :pl.kx]
        ref = np.fft.ifft(np.fft.fft(pad, axis=0) * pl.psf_hat_natural[0, :, :pl.kx], axis=0)[:n] * L
        assert max_rel(outs[1][b_], ref) < 1e-13


@pytest.mark.parametrize('M,real_psf,rows', [(4, 1, 192), (2, 0, 192), (8, 1, 192), (4, 0, 128), (2, 1, 128)])
def test_outer_body_with_the_row_count_at_compile_time(emu, M, real_psf, rows):
    """colconv_outer_body<..., NYC> (the device: images of M x 512 rows on the 576 core; here M x 192 rows on the 256 core): every
    residue class has whole 64-row steps, the rows of a class's tile that do not exist are constants, pad columns travel with the
    tile -- bit for bit the generic body's result, and numpy's."""
    emu.emu_set_special.argtypes = [ctypes.c_int]
    Li, ny, kx, V, frames = 256, M * rows, 11, 1, 2      # (rows = 128: whole 128-row steps -- the two-columns-per-access tile I/O)
    L, pitch = M * Li, 16
    rng = np.random.default_rng(900 + M)
    x = np.zeros((frames, ny, pitch), dtype=np.complex128)
    x[:, :, :kx] = rng.standard_normal((frames, ny, kx)) + 1j * rng.standard_normal((frames, ny, kx))
    ph = rng.standard_normal((V, kx, L)) + (0 if real_psf else 1j) * rng.standard_normal((V, kx, L))
    psf_arg = np.ascontiguousarray(ph.real if real_psf else ph.astype(np.complex128))
    outs = []
    for special in (0, 1):
        out = np.zeros((frames * V, ny, pitch), dtype=np.complex128)
        emu.emu_set_special(special)
        try:
            assert emu.emu_col_outer_f64(Li, M, _p(_slack(x)), _p(out), _p(psf_arg), real_psf, ny, kx, pitch, V, frames, 1, 0) == 0
        finally:
            emu.emu_set_special(0)
        outs.append(out[:, :, :kx].copy())
    assert np.array_equal(outs[0], outs[1])
    full = np.zeros((frames, L, kx), dtype=np.complex128)
    full[:, :ny] = x[:, :, :kx]
    spec = np.fft.fft(full, axis=1)
    for f in range(frames):
        ref = np.fft.ifft(spec[f] * ph[0].T, axis=0)[:ny] * L
        assert max_rel(outs[1][f], ref) < 1e-12
