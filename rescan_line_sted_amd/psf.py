"""PSF generation half of the line_sted_tools mirror: psf_report, generate_psfs,
tune_psf, get_width (reference: figure_generation/line_sted_tools.py:75-476,
653-668; "ref:NNN" = line numbers there).

The array work (truncated-Gaussian blurs, saturation maths, the rescan scan
loop, reductions) runs in float64 HIP kernels (csrc/psf_kernels.hip); the
Gaussian width fit is the restated MINPACK iteration in csrc/gauss_fit.cpp; the
scalar Brent search of tune_psf is host Python below, as in the reference.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import lib, check, ptr, as_f64

_TYPES = {'point': 0, 'line': 1}
_POINT_KEYS = ('excitation', 'depletion', 'excitation_fraction', 'depletion_fraction', 'sted')
_LINE_KEYS = _POINT_KEYS + ('descan_sted', 'rescan_sted')
_FWHM = 2 * np.sqrt(2 * np.log(2))


def _ctx():
    return _lib.Context.get(int(os.environ.get('RLSTED_DEVICE', '0')))


def gaussian_filter(a, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter (mode='reflect') for 3-D float64 input, on the device."""
    a = as_f64(a)
    if a.ndim != 3:
        raise ValueError('expected a 3-D array')
    s3 = as_f64(np.broadcast_to(np.asarray(sigma, dtype=np.float64), (3,)))
    out = np.empty_like(a)
    check(lib.rl_gaussian_filter(_ctx().handle, ptr(a), ptr(out), a.shape[0], a.shape[1], a.shape[2],
                                 ptr(s3), float(truncate)))
    return out


def get_width(x):
    """ref:653-668: fit a Gaussian to the 1-D array x (coordinates 0..len-1, start
    [1, len/2, 1]); returns (sigma, fitted_curve).  sigma's sign is not
    constrained, exactly like the reference."""
    y = as_f64(np.asarray(x).ravel())
    p = np.zeros(3)
    info = ctypes.c_int()
    check(lib.rl_gauss_fit(ptr(y), y.size, ptr(p), ctypes.byref(info)))
    if info.value not in (1, 2, 3, 4):
        # scipy.optimize.curve_fit raises here (ref SURVEY 8b error conventions)
        raise RuntimeError('Optimal parameters not found: MINPACK info %d' % info.value)
    A, mu, sigma = p
    coords = np.arange(y.size, dtype=np.float64)
    return sigma, A * np.exp(-(coords - mu) ** 2 / (2. * sigma ** 2))


def _as_dict(psf_type, arrays, n_y, n_x):
    keys = _POINT_KEYS if psf_type == 'point' else _LINE_KEYS
    out = {k: arrays[i].reshape(1, n_y, n_x) for i, k in enumerate(keys)}
    if psf_type == 'point':
        out['descan_sted'] = out['sted']          # same array object, ref:249
    return out


def generate_psfs(shape, excitation_brightness, depletion_brightness, blur_sigma,
                  psf_type='point', output_dir=None, verbose=True):
    """ref:168-176 signature, ref:347-363 return dict of (1, ny, nx) float64 arrays."""
    if psf_type not in _TYPES:
        raise UnboundLocalError("psf_type must be 'point' or 'line'")   # the reference dies with this
    nz, ny, nx = shape
    if nz != 1:
        raise NotImplementedError('device PSF generation handles shape (1, ny, nx)')
    narr = 5 if psf_type == 'point' else 7
    arrays = np.empty((narr, ny, nx), dtype=np.float64)
    scalars = np.zeros(10)
    check(lib.rl_psf_generate(_ctx().handle, _TYPES[psf_type], ny, nx, float(excitation_brightness),
                              float(depletion_brightness), float(blur_sigma), 0, ptr(arrays), None, ptr(scalars)))
    if psf_type == 'line' and verbose:
        print(" Ideal line rescan ratio: %0.5f" % (scalars[1]))
        print(" Neareset integer:", int(scalars[0]))
        print(" Calculating rescan psf...", end='')
        print(" ...done.")
    psfs = _as_dict(psf_type, arrays, ny, nx)
    if output_dir is not None:
        extras = _line_extras(ny, nx, excitation_brightness, depletion_brightness, blur_sigma, int(scalars[0])) if psf_type == 'line' else None
        _dump_psfs(psfs, psf_type, output_dir, extras)
    return psfs


def _line_extras(ny, nx, excitation_brightness, depletion_brightness, blur_sigma, ratio):
    """The two intermediates of the 'line' branch that the reference also writes (ref:339-341): the emission PSF and
    the unscaled rescan ring (1, ny, ratio * nx)."""
    emission, unscaled = np.empty((1, ny, nx)), np.empty((1, ny, ratio * nx))
    check(lib.rl_psf_generate_line_extras(_ctx().handle, ny, nx, float(excitation_brightness), float(depletion_brightness),
                                          float(blur_sigma), int(ratio), ptr(emission), ptr(unscaled)))
    return {'emission_psf.tif': emission, 'sted_psf_line_rescan_unscaled.tif': unscaled}


def _dump_psfs(psfs, psf_type, output_dir, extras=None):
    """Optional TIFF dump, ref:311-346: the reference's nine (line) / five (point) files under its file names."""
    from . import np_tif
    if not os.path.exists(output_dir):
        os.mkdir(output_dir)
    sfx = '_psf_point.tif' if psf_type == 'point' else '_psf_line.tif'
    for key in _POINT_KEYS:
        np_tif.array_to_tif(psfs[key], os.path.join(output_dir, key + sfx))
    if psf_type == 'line':
        for name in ('emission_psf.tif', 'sted_psf_line_rescan_unscaled.tif'):
            np_tif.array_to_tif((extras or {})[name], os.path.join(output_dir, name))
        np_tif.array_to_tif(psfs['rescan_sted'], os.path.join(output_dir, 'sted_psf_line_rescan.tif'))
        np_tif.array_to_tif(psfs['descan_sted'], os.path.join(output_dir, 'sted_psf_line_descan.tif'))


def _report(psf_type, excitation_brightness, depletion_brightness,
            steps_per_excitation_psf_width, pulses_per_position, want_arrays):
    blur_sigma = steps_per_excitation_psf_width / _FWHM                     # ref:91
    n = 1 + 2 * int(np.round(5 * blur_sigma))                               # ref:92
    narr = 5 if psf_type == 'point' else 7
    arrays = np.empty((narr, n, n), dtype=np.float64) if want_arrays else None
    rep = np.zeros(8)
    check(lib.rl_psf_report(_ctx().handle, _TYPES[psf_type], float(excitation_brightness),
                            float(depletion_brightness), float(steps_per_excitation_psf_width),
                            float(pulses_per_position), ptr(arrays) if want_arrays else None, ptr(rep)))
    assert int(rep[5]) == n
    assert rep[7] == 1.0          # ref:105-106,120: each PSF peaks on its central row
    return rep, arrays, n, blur_sigma


def psf_report(psf_type, excitation_brightness, depletion_brightness,
               steps_per_excitation_psf_width, pulses_per_position,
               verbose=True, output_dir=None):
    """ref:75-166.  Same arguments, same dict keys (point: ref:150-156, line:
    ref:158-166)."""
    if psf_type not in _TYPES:
        raise UnboundLocalError("psf_type must be 'point' or 'line'")
    rep, arrays, n, blur_sigma = _report(psf_type, excitation_brightness, depletion_brightness,
                                         steps_per_excitation_psf_width, pulses_per_position, True)
    psfs = _as_dict(psf_type, arrays, n, n)
    if output_dir is not None:
        extras = (_line_extras(n, n, excitation_brightness, depletion_brightness, blur_sigma, int(rep[6]))
                  if psf_type == 'line' else None)
        _dump_psfs(psfs, psf_type, output_dir, extras)
    if verbose:
        if psf_type == 'line':
            print(" Neareset integer:", int(rep[6]))
        ex_sigma, _ = get_width(psfs['excitation'][0, n // 2, :])
        print("PSF type:", psf_type)
        print("Excitation psf width: %0.3f" % (ex_sigma * _FWHM), "pixels FWHM")
        print("STED psf width: %0.3f" % (blur_sigma / rep[0] * _FWHM), "pixels FWHM")
        print("STED improvement in excitation PSF width: %0.3f" % (rep[0]))
        if psf_type == 'line':
            print("Rescan STED psf width: %0.3f" % (blur_sigma / rep[1] * _FWHM))
            print("Rescan STED improvement in PSF width: %0.3f" % (rep[1]))
        print("Excitation dose: %0.3f" % (rep[2]), "half-saturations")
        print("Depletion dose: %0.3f" % (rep[3]), "half-saturations")
        print("Expected emissions per molecule: %0.4f\n" % (rep[4]))
    out = {}
    if psf_type == 'line':
        out['resolution_improvement_rescanned'] = rep[1]
    out['resolution_improvement_descanned'] = rep[0]
    out['excitation_dose'] = rep[2]
    out['depletion_dose'] = rep[3]
    out['expected_emission'] = rep[4]
    out['pulses_per_position'] = pulses_per_position
    out['psfs'] = psfs
    return out


def psf_report_batch(parameter_sets, with_psfs=False):
    """psf_report (ref:75-166) for a list of parameter sets in ONE device pass (rl_psf_report_batch): the
    sweeps of line_sted_figure_1.py:33-48 (480 sets) and line_sted_figure_a1.py.  parameter_sets: iterable
    of (psf_type, excitation_brightness, depletion_brightness, steps_per_excitation_psf_width,
    pulses_per_position) tuples or dicts with those keys.  Returns one psf_report dict per set, each bit
    for bit what psf_report returns ('psfs' only when with_psfs)."""
    import ctypes
    keys = ('psf_type', 'excitation_brightness', 'depletion_brightness', 'steps_per_excitation_psf_width',
            'pulses_per_position')
    sets = [tuple(p[k] for k in keys) if isinstance(p, dict) else tuple(p) for p in parameter_sets]
    for p in sets:
        if p[0] not in _TYPES:
            raise UnboundLocalError("psf_type must be 'point' or 'line'")
    n_sets = len(sets)
    if n_sets == 0:
        return []
    params = np.array([[_TYPES[p[0]]] + [float(v) for v in p[1:]] for p in sets], dtype=np.float64)
    rep = np.zeros((n_sets, 8))
    arrays, ptrs = [], None
    if with_psfs:
        dp = ctypes.POINTER(ctypes.c_double)
        ptrs = (dp * n_sets)()
        for i, p in enumerate(sets):
            n = 1 + 2 * int(np.round(5 * (p[3] / _FWHM)))
            arrays.append(np.empty((5 if p[0] == 'point' else 7, n, n), dtype=np.float64))
            ptrs[i] = arrays[i].ctypes.data_as(dp)
    check(lib.rl_psf_report_batch(_ctx().handle, n_sets, ptr(params), ptr(rep), ptrs))
    out = []
    for i, p in enumerate(sets):
        assert rep[i, 7] == 1.0          # ref:105-106,120: each PSF peaks on its central row
        r = {}
        if p[0] == 'line':
            r['resolution_improvement_rescanned'] = rep[i, 1]
        r['resolution_improvement_descanned'] = rep[i, 0]
        r['excitation_dose'] = rep[i, 2]
        r['depletion_dose'] = rep[i, 3]
        r['expected_emission'] = rep[i, 4]
        r['pulses_per_position'] = p[4]
        if with_psfs:
            n = int(rep[i, 5])
            r['psfs'] = _as_dict(p[0], arrays[i], n, n)
        out.append(r)
    return out


# ---------------------------------------------------------------------------
# tune_psf (ref:365-476).  The reference minimises with scipy.optimize.
# minimize_scalar's default method: a golden-section bracket search started at
# (0, 1) followed by Brent's parabolic/golden minimiser (tol 1.48e-8).  Both are
# restated here as one small class.
# ---------------------------------------------------------------------------
class _ScalarMinimizer:
    GOLD, TINY, GROW = 1.618034, 1e-21, 110.0
    CGOLD, MINTOL = 0.3819660, 1.0e-11

    def __init__(self, func, tol=1.48e-8, maxiter=500):
        self.func, self.tol, self.maxiter = func, tol, maxiter
        self.calls = 0

    def f(self, x):
        self.calls += 1
        return self.func(x)

    def bracket(self, xa=0.0, xb=1.0):
        f = self.f
        fa, fb = f(xa), f(xb)
        if fa < fb:
            xa, xb, fa, fb = xb, xa, fb, fa
        xc = xb + self.GOLD * (xb - xa)
        fc = f(xc)
        rounds = 0
        while fc < fb:
            p = (xb - xa) * (fb - fc)
            q = (xb - xc) * (fb - fa)
            d = q - p
            denom = 2.0 * self.TINY if abs(d) < self.TINY else 2.0 * d
            w = xb - ((xb - xc) * q - (xb - xa) * p) / denom
            wlim = xb + self.GROW * (xc - xb)
            if rounds > 1000:
                raise RuntimeError('No valid bracket was found')
            rounds += 1
            if (w - xc) * (xb - w) > 0.0:              # parabola minimum between b and c
                fw = f(w)
                if fw < fc:
                    return xb, w, xc, fb, fw, fc
                if fw > fb:
                    return xa, xb, w, fa, fb, fw
                w = xc + self.GOLD * (xc - xb)
                fw = f(w)
            elif (w - wlim) * (wlim - xc) >= 0.0:      # beyond the growth limit
                w = wlim
                fw = f(w)
            elif (w - wlim) * (xc - w) > 0.0:          # between c and the limit
                fw = f(w)
                if fw < fc:
                    xb, xc, fb, fc = xc, w, fc, fw
                    w = xc + self.GOLD * (xc - xb)
                    fw = f(w)
            else:
                w = xc + self.GOLD * (xc - xb)
                fw = f(w)
            xa, xb, xc, fa, fb, fc = xb, xc, w, fb, fc, fw
        return xa, xb, xc, fa, fb, fc

    def minimize(self):
        xa, xb, xc, fa, fb, fc = self.bracket()
        x = w = v = xb
        fx = fw = fv = fb
        lo, hi = min(xa, xc), max(xa, xc)
        step_before, step = 0.0, 0.0
        for _ in range(self.maxiter):
            tol1 = self.tol * abs(x) + self.MINTOL
            tol2 = 2.0 * tol1
            mid = 0.5 * (lo + hi)
            if abs(x - mid) < tol2 - 0.5 * (hi - lo):
                break
            use_golden = True
            if abs(step_before) > tol1:
                r = (x - w) * (fx - fv)
                q = (x - v) * (fx - fw)
                p = (x - v) * q - (x - w) * r
                q = 2.0 * (q - r)
                if q > 0.0:
                    p = -p
                q = abs(q)
                prev = step_before
                step_before = step
                if p > q * (lo - x) and p < q * (hi - x) and abs(p) < abs(0.5 * q * prev):
                    step = p / q
                    u = x + step
                    if (u - lo) < tol2 or (hi - u) < tol2:
                        step = tol1 if mid - x >= 0 else -tol1
                    use_golden = False
            if use_golden:
                step_before = (lo - x) if x >= mid else (hi - x)
                step = self.CGOLD * step_before
            if abs(step) < tol1:
                u = x + tol1 if step >= 0 else x - tol1
            else:
                u = x + step
            fu = self.f(u)
            if fu > fx:
                if u < x:
                    lo = u
                else:
                    hi = u
                if fu <= fw or w == x:
                    v, fv, w, fw = w, fw, u, fu
                elif fu <= fv or v == x or v == w:
                    v, fv = u, fu
            else:
                if u >= x:
                    lo = x
                else:
                    hi = x
                v, fv, w, fw, x, fx = w, fw, x, fx, u, fu
        return x


def tune_psf(psf_type, scan_type, desired_resolution_improvement,
             desired_emissions_per_molecule, max_excitation_brightness=0.5,
             steps_per_improved_psf_width=3, relative_error=1e-6,
             verbose_results=False, verbose_iterations=False):
    """ref:365-476: find excitation brightness, depletion brightness and pulse
    count that give the requested resolution improvement and emissions."""
    assert (psf_type, scan_type) in (('point', 'descanned'),
                                     ('line', 'descanned'),
                                     ('line', 'rescanned'))
    assert float(desired_resolution_improvement) == desired_resolution_improvement
    assert float(desired_emissions_per_molecule) == desired_emissions_per_molecule
    assert float(max_excitation_brightness) == max_excitation_brightness
    assert float(steps_per_improved_psf_width) == steps_per_improved_psf_width
    assert float(relative_error) == relative_error
    steps = steps_per_improved_psf_width * desired_resolution_improvement      # ref:406
    args = {'psf_type': psf_type,
            'excitation_brightness': max_excitation_brightness,
            'depletion_brightness': 1,
            'steps_per_excitation_psf_width': steps,
            'pulses_per_position': 1,
            'verbose': False,
            'output_dir': None}
    which = 0 if scan_type == 'descanned' else 1

    def scalars():          # psf_report without moving the PSF arrays off the device
        rep, _, _, _ = _report(psf_type, args['excitation_brightness'], args['depletion_brightness'],
                               steps, args['pulses_per_position'], False)
        return rep

    num_iterations = 0
    while True:
        num_iterations += 1
        if num_iterations >= 10:                                                # ref:419-421
            print("Max. iterations exceeded; giving up")
            break

        def resolution_miss(depletion_brightness):                             # ref:424-428
            args['depletion_brightness'] = abs(depletion_brightness)
            return (scalars()[which] - desired_resolution_improvement) ** 2
        args['depletion_brightness'] = abs(_ScalarMinimizer(resolution_miss).minimize())
        if verbose_iterations:
            print("Depletion brightness:", args['depletion_brightness'])
        args['excitation_brightness'] = max_excitation_brightness              # ref:434-438
        args['pulses_per_position'] = 1
        args['pulses_per_position'] = np.ceil(desired_emissions_per_molecule / scalars()[4])
        if verbose_iterations:
            print(args['pulses_per_position'], "pulses.")

        def emission_miss(excitation_brightness):                              # ref:443-447
            args['excitation_brightness'] = abs(excitation_brightness)
            return (scalars()[4] - desired_emissions_per_molecule) ** 2
        args['excitation_brightness'] = abs(_ScalarMinimizer(emission_miss).minimize())
        if verbose_iterations:
            print("Excitation brightness:", args['excitation_brightness'])
        relative_resolution_error = ((scalars()[which] - desired_resolution_improvement) /
                                     desired_resolution_improvement)           # signed, ref:457-461
        if relative_resolution_error < relative_error:
            break
    results = psf_report(**args)                                               # ref:451
    if verbose_results:
        print("PSF tuning complete, after", num_iterations, "iterations.")
        print(" Inputs:")
        for k in sorted(args.keys()):
            print('  ', k, ': ', args[k], sep='')
        print(" Outputs:")
        for k in sorted(results.keys()):
            if k == 'psfs':
                print('  ', k, ': ', sorted(results[k].keys()), sep='')
            else:
                print('  ', k, ': ', results[k], sep='')
        print()
    results.update(args)                                                       # ref:475
    return results


# ---------------------------------------------------------------------------
# The figure-2 harness around the module: PSF rotation and the
# point / line operating-point pair, line_sted_figure_2.py:183-248, 264-272.
# ---------------------------------------------------------------------------
def rotate(x, degrees):
    """line_sted_figure_2.py:264-272: (1, ny, nx) PSF rotated in its plane; exact
    for 0 and 90 degrees, cubic-spline + clip to [0, 1.1 max] otherwise (device)."""
    if degrees == 0:
        return x
    if degrees == 90:
        return np.rot90(np.squeeze(x)).reshape(x.shape)
    x = as_f64(x)
    out = np.empty_like(x)
    check(lib.rl_rotate_psf_stack(_ctx().handle, ptr(x), ptr(out), x.shape[0], x.shape[1], x.shape[2], float(degrees)))
    return out


def psf_comparison_pair(point_resolution_improvement, line_resolution_improvement,
                        point_emissions_per_molecule, line_emissions_per_molecule,
                        line_scan_type, line_num_orientations,
                        max_excitation_brightness=0.25, steps_per_improved_psf_width=4,
                        steps_per_excitation_psf_width=25):
    """line_sted_figure_2.py:183-248: tune a point-STED and a line-STED operating
    point of equal dose, sample both PSFs at the display resolution, normalise them to
    their emission level and fan the line PSF out over its scan orientations."""
    point = tune_psf('point', 'descanned', float(point_resolution_improvement),
                     float(point_emissions_per_molecule),
                     max_excitation_brightness=max_excitation_brightness,
                     steps_per_improved_psf_width=float(steps_per_improved_psf_width))
    line = tune_psf('line', line_scan_type, float(line_resolution_improvement),
                    float(line_emissions_per_molecule),
                    max_excitation_brightness=max_excitation_brightness,
                    steps_per_improved_psf_width=float(steps_per_improved_psf_width))
    fine_point = psf_report('point', point['excitation_brightness'], point['depletion_brightness'],
                            steps_per_excitation_psf_width, point['pulses_per_position'], verbose=False)
    fine_line = psf_report('line', line['excitation_brightness'], line['depletion_brightness'],
                           steps_per_excitation_psf_width, line['pulses_per_position'], verbose=False)
    point_sted_psf = [point['expected_emission'] *
                      (fine_point['psfs']['descan_sted'] / fine_point['psfs']['descan_sted'].sum())]
    assert line['pulses_per_position'] >= line_num_orientations          # script :239
    which = {'descanned': 'descan_sted', 'rescanned': 'rescan_sted'}[line_scan_type]
    base = fine_line['psfs'][which] / fine_line['psfs'][which].sum()
    line_sted_psfs = [1 / line_num_orientations * line['expected_emission'] * rotate(base, angle)
                      for angle in np.arange(0, 180, 180 / line_num_orientations)]
    return {'point_sted_psf': point_sted_psf, 'line_sted_psfs': line_sted_psfs,
            'point': point, 'line': line}


# The twelve hand-tuned comparisons of the figure-2 script (line_sted_figure_2.py:77-162):
# name -> (point R, line R, point emissions, line emissions, line scan type, orientations, max excitation)
FIGURE_2_OPERATING_POINTS = {
    '1p0x_ld': (0.99, 0.99, 4, 4, 'descanned', 1, 0.01),
    '1p0x_lr': (0.99, 1.38282445, 4, 4, 'rescanned', 2, 0.01),
    '1p5x_ld': (1.5, 2.68125, 4, 2.825, 'descanned', 3, 0.25),
    '1p5x_lr': (1.5, 2.95425, 4, 2.618, 'rescanned', 3, 0.25),
    '2p0x_ld': (2, 4.04057, 4, 3.007, 'descanned', 4, 0.25),
    '2p0x_lr': (2, 4.07614, 4, 3.0227, 'rescanned', 4, 0.25),
    '2p5x_ld': (2.5, 5.13325, 4, 3.792, 'descanned', 6, 0.25),
    '2p5x_lr': (2.5, 5.15129, 4, 3.8, 'rescanned', 6, 0.25),
    '3p0x_ld': (3, 5.94563, 4, 5.034, 'descanned', 8, 0.25),
    '3p0x_lr': (3, 5.95587, 4, 5.0385, 'rescanned', 8, 0.25),
    '4p0x_ld': (4, 7.8386627, 4, 7.371, 'descanned', 10, 0.25),
    '4p0x_lr': (4, 7.840982, 4, 7.37195, 'rescanned', 10, 0.25),
}


def figure_2_psfs(names=None):
    """The PSF sets of calculate_psfs (line_sted_figure_2.py:66-181) for the named
    comparisons (default: all twelve): {name + '_point_sted' / '_line_N_angles_sted': [psfs]}."""
    out, comparisons = {}, {}
    for name in (names or FIGURE_2_OPERATING_POINTS):
        pr, lr, pe, le, scan, nori, maxexc = FIGURE_2_OPERATING_POINTS[name]
        c = psf_comparison_pair(pr, lr, pe, le, scan, nori, max_excitation_brightness=maxexc)
        comparisons[name] = c
        out[name + '_point_sted'] = c['point_sted_psf']
        out[name + '_line_%i_angles_sted' % len(c['line_sted_psfs'])] = c['line_sted_psfs']
    return out, comparisons
