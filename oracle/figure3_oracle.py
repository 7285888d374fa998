"""CPU restatement (numpy, float64) of the scan-position-by-scan-position imaging simulator of
figure_generation/line_sted_figure_3.py.  TEST INFRASTRUCTURE: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Pinned: tests/golden/g11_fig3.npz holds what the reference's own `simulate_imaging` hands to its
figure code (tests/golden/make_golden_fig3.py runs the reference's function definitions -- never its
module-level main() -- with a recording `generate_figure`); tests/test_oracle_golden.py checks this
restatement against it, and the scipy.ndimage restatements below against scipy itself.

Restated third-party arithmetic (scipy.ndimage, not in /root/reference; versions as recorded in the
golden file): `shift`, `rotate(mode='nearest', reshape=False)`, `zoom` with order-3 splines and
`gaussian_filter(truncate=...)` (the latter in line_sted_oracle.py).
"""
import numpy as np

from . import line_sted_oracle as orc

IMAGING_TYPES = ('descan_point', 'nondescan_multipoint', 'descan_line', 'rescan_line')


# --------------------------------------------------------------------------------------------
# scipy.ndimage.shift(x, (0, sy, sx)) as line_sted_figure_3.py:389-392 uses it: integer shifts only
# (the scan positions are integer multiples of the integer step, :102,112-114,121-124,135-137).  An
# interpolating cubic spline reproduces its samples, so the result is the input moved by (sy, sx)
# with zeros moved in (mode='constant', cval 0); the clip to [0, 1.1 max] is kept.
# --------------------------------------------------------------------------------------------
def shift_int(x, sy, sx):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    ny, nx = x.shape[-2:]
    sy, sx = int(sy), int(sx)
    if abs(sy) >= ny or abs(sx) >= nx:
        return out
    dst_y = slice(max(sy, 0), ny + min(sy, 0))
    src_y = slice(max(-sy, 0), ny + min(-sy, 0))
    dst_x = slice(max(sx, 0), nx + min(sx, 0))
    src_x = slice(max(-sx, 0), nx + min(-sx, 0))
    out[..., dst_y, dst_x] = x[..., src_y, src_x]
    return out


def shift(x, s):                                            # ref fig3:389-392
    return np.clip(shift_int(x, s[-2], s[-1]), 0, 1.1 * np.max(x))


def _spline_eval(coef, y, x, clamp=False):
    """Cubic B-spline with coefficients `coef` at (y, x); neighbour indices beyond the edges are
    mirrored (scipy's modes 'constant' / 'mirror') or clamped (mode 'nearest')."""
    ny, nx = coef.shape
    fy, fx = np.floor(y), np.floor(x)
    wy, wx = orc._bspline3_weights(y - fy), orc._bspline3_weights(x - fx)
    out = np.zeros(np.broadcast(y, x).shape)
    for i in range(4):
        iy = (fy - 1 + i).astype(np.int64)
        yy = np.clip(iy, 0, ny - 1) if clamp else orc._mirror_index(iy, ny)
        for j in range(4):
            ix = (fx - 1 + j).astype(np.int64)
            xx = np.clip(ix, 0, nx - 1) if clamp else orc._mirror_index(ix, nx)
            out = out + wy[i] * wx[j] * coef[yy, xx]
    return out


NPAD = 12   # scipy.ndimage._interpolation._prepad_for_spline_filter for mode 'nearest'


def _spline_prefilter_axis_reflect(c, axis):
    """Cubic B-spline coefficients along `axis` under half-sample symmetric ('reflect') boundary
    conditions: what scipy.ndimage.spline_filter1d applies for mode='nearest' (the edge-padded array of
    rotate(mode='nearest') is filtered with it).  Exact-sum initialisation of the causal pass."""
    c = np.moveaxis(np.array(c, dtype=np.float64), axis, 0).copy()
    n = c.shape[0]
    if n == 1:
        return np.moveaxis(c, 0, axis)
    z = np.sqrt(3.0) - 2.0
    c *= (1 - z) * (1 - 1 / z)
    zn = z ** n
    first = c[0].copy()
    acc = c[0] + zn * c[n - 1]
    zi = z
    for i in range(1, n):
        acc = acc + zi * (c[i] + zn * c[n - 1 - i])
        zi *= z
    c[0] = acc * (z / (1 - zn * zn)) + first
    for i in range(1, n):
        c[i] += z * c[i - 1]
    c[n - 1] = c[n - 1] * (z / (z - 1))
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    return np.moveaxis(c, 0, axis)


def rotate_plane_nearest(plane, degrees):
    """scipy.ndimage.rotate(plane, degrees, reshape=False, order=3, mode='nearest') of a 2-D array: the
    input is padded by 12 edge samples, prefiltered (reflect conditions on the padded array), and the
    spline of the padded array is evaluated at the (unclamped) source coordinates with the indices of
    the four neighbour samples clamped to the padded extent."""
    a = np.asarray(plane, dtype=np.float64)
    ny, nx = a.shape
    th = np.deg2rad(degrees)
    c, s = np.cos(th), np.sin(th)
    cy, cx = (ny - 1) / 2, (nx - 1) / 2
    off_y, off_x = cy - (c * cy + s * cx), cx - (-s * cy + c * cx)
    oy, ox = np.meshgrid(np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing='ij')
    padded = np.pad(a, NPAD, mode='edge')
    coef = _spline_prefilter_axis_reflect(_spline_prefilter_axis_reflect(padded, 0), 1)
    y = c * oy + s * ox + off_y + NPAD
    x = -s * oy + c * ox + off_x + NPAD
    return _spline_eval(coef, y, x, clamp=True)


def rotate(x, angle_degrees):                               # ref fig3:382-391
    if angle_degrees == 0:
        return np.array(x, dtype=np.float64, copy=True)
    x = np.asarray(x, dtype=np.float64)
    r = np.stack([rotate_plane_nearest(p, angle_degrees) for p in x])
    return np.clip(r, 0, 1.1 * x.max())


def zoom_y(plane, factor):
    """scipy.ndimage.zoom(plane, (factor, 1)) with its defaults (order 3, mode 'constant', prefilter,
    grid_mode False): output rows o sample the spline at o * (ny - 1) / (out - 1)."""
    a = np.asarray(plane, dtype=np.float64)
    ny, nx = a.shape
    out_ny = int(round(ny * factor))
    coef = orc._spline_prefilter_axis(orc._spline_prefilter_axis(a, 0), 1)
    z = (ny - 1) / (out_ny - 1) if out_ny > 1 else 1.0
    y = np.arange(out_ny, dtype=np.float64) * z
    inside = (y >= 0) & (y <= ny - 1)
    yy, xx = np.meshgrid(np.clip(y, 0, ny - 1), np.arange(nx, dtype=np.float64), indexing='ij')
    out = _spline_eval(coef, yy, xx)
    return np.where(inside[:, None], out, 0.0)


def scale_y(x, scaling_factor):                             # ref fig3:394-409
    assert x.ndim == 3 and x.shape[0] == 1 and x.shape[1] > 1
    scaled = zoom_y(x[0], scaling_factor)
    y_dif = x.shape[-2] - scaled.shape[-2]
    return np.pad(scaled, ((y_dif // 2, y_dif - y_dif // 2), (0, 0)), 'constant').reshape(x.shape)


def scan_setup(obj_shape, imaging_type, psf_width, R, pad):
    """Scan geometry of simulate_imaging (ref fig3:98-137): sigma, step, excitation separation, the
    scan positions, the padded shape."""
    _, n_y, n_x = obj_shape
    psf_sigma = psf_width / (2 * np.sqrt(2 * np.log(2)))
    step = int(np.round(psf_width / (4 * R)))
    exc_sep = None
    if imaging_type in ('descan_line', 'rescan_line'):
        positions = [(int(y), 0) for y in np.arange(-n_y // 2, n_y // 2 + 1, step)]
    elif imaging_type == 'descan_point':
        positions = [(int(y), int(x)) for y in np.arange(-n_y // 2, n_y // 2 + 1, step)
                     for x in np.arange(-n_x // 2, n_x // 2 + 1, step)]
    else:
        exc_sep = int(step * np.round(psf_width * 1.4 / step))
        positions = [(int(y), int(x)) for y in np.arange(0, exc_sep, step) for x in np.arange(0, exc_sep, step)]
    return psf_sigma, step, exc_sep, positions


def centered_excitation(shape, imaging_type, psf_sigma, R, pad, exc_sep):   # ref fig3:109-139
    exc = np.zeros(shape)
    if imaging_type in ('descan_line', 'rescan_line'):
        exc[0, shape[1] // 2, :] = 1
        sigma = (0, psf_sigma / R, 0)
    elif imaging_type == 'descan_point':
        exc[0, shape[1] // 2, shape[2] // 2] = 1
        sigma = (0, psf_sigma / R, psf_sigma / R)
    else:
        exc[0, pad:-pad:exc_sep, pad:-pad:exc_sep] = 1
        sigma = (0, psf_sigma / R, psf_sigma / R)
    return orc.gaussian_filter(exc, sigma, truncate=8)


def simulate_imaging(obj, imaging_type, psf_width, R, num_orientations, pulses_per_position, pad,
                     generate_figure=None):
    """ref fig3:76-273 without the file output: the loop over (pass, orientation, scan position).
    `generate_figure` receives what the reference's figure code receives, for the frames the reference
    renders.  Returns {'maxima': (exc, glow, inst, cum, reconst, new_sig), 'reconstructions': {rot:
    final reconstruction (padded)}, 'pulses_delivered', 'camera_exposures', 'frames': count}."""
    assert obj.ndim == 3 and obj.shape[0] == 1 and imaging_type in IMAGING_TYPES
    assert psf_width >= 1 and R >= 1 and pad > 0 and int(pad) == pad
    psf_sigma, step, exc_sep, scan_positions = scan_setup(obj.shape, imaging_type, psf_width, R, pad)
    _, n_y, n_x = obj.shape
    obj = np.pad(np.asarray(obj, dtype=np.float64), ((0, 0), (pad, pad), (pad, pad)), 'constant')
    if imaging_type in ('descan_point', 'nondescan_multipoint'):
        num_orientations = 1
    centered_exc = centered_excitation(obj.shape, imaging_type, psf_sigma, R, pad, exc_sep)
    max_exc = centered_exc[0, pad:-pad, pad:-pad].max()
    max_glow = max_inst_sig = max_cum_sig = max_reconst = max_new_sig = 0
    finals, frames = {}, 0
    for which_run in ('find_maxima', 'generate_figures'):
        camera_exposures, pulses_delivered = 0, 0
        for rot in np.arange(0, 180, 180 / num_orientations)[::-1]:
            if which_run == 'find_maxima' and rot > 0:
                continue
            rot_obj = rotate(obj, rot)
            cum_detector_sig = np.zeros(obj.shape)
            reconstruction = np.zeros(obj.shape)
            for which_pos, (shift_y, shift_x) in enumerate(scan_positions):
                pulses_delivered += pulses_per_position
                last_reconstruction = reconstruction.copy()
                exc = shift(centered_exc, (0, shift_y, shift_x))
                glow = rot_obj * exc
                descanned_glow = shift(glow, (0, -shift_y, -shift_x))
                if imaging_type in ('descan_line', 'descan_point'):
                    inst_detector_sig = orc.gaussian_filter(descanned_glow, psf_sigma)
                    cum_detector_sig = inst_detector_sig
                    y0 = shift_y + n_y // 2 + pad
                    if imaging_type == 'descan_line':
                        reconstruction[0, y0:y0 + step, :] = inst_detector_sig.sum(axis=1, keepdims=True)
                        camera_exposures += 1
                    else:
                        x0 = shift_x + n_x // 2 + pad
                        reconstruction[0, y0:y0 + step, x0:x0 + step] = inst_detector_sig.sum()
                        camera_exposures = 'N/A'
                elif imaging_type == 'nondescan_multipoint':
                    inst_detector_sig = orc.gaussian_filter(glow, psf_sigma)
                    cum_detector_sig = inst_detector_sig
                    for y_sp in range(pad + shift_y, pad + shift_y + n_y, exc_sep):
                        for x_sp in range(pad + shift_x, pad + shift_x + n_x, exc_sep):
                            region = inst_detector_sig[0, max(y_sp - exc_sep // 3, 0):y_sp + exc_sep // 3,
                                                       max(x_sp - exc_sep // 3, 0):x_sp + exc_sep // 3]
                            reconstruction[0, y_sp - step // 2:y_sp - step // 2 + step,
                                           x_sp - step // 2:x_sp - step // 2 + step] = region.sum()
                    camera_exposures += 1
                else:
                    scaled = scale_y(orc.gaussian_filter(descanned_glow, psf_sigma), 1 / (R ** 2 + 1))
                    inst_detector_sig = shift(scaled, (0, shift_y, shift_x))
                    cum_detector_sig += inst_detector_sig
                    if (shift_y, shift_x) == scan_positions[-1]:
                        reconstruction = cum_detector_sig
                        camera_exposures += 1
                new_signal = reconstruction - last_reconstruction
                if which_run == 'find_maxima':
                    max_glow = max(glow.max(), max_glow)
                    max_inst_sig = max(inst_detector_sig.max(), max_inst_sig)
                    max_cum_sig = max(cum_detector_sig.max(), max_cum_sig)
                    max_reconst = max(reconstruction.max(), max_reconst)
                    max_new_sig = max(new_signal.max(), max_new_sig)
                else:
                    num_to_skip = max(int(np.round(len(scan_positions) / 150)), 1)
                    if which_pos % num_to_skip != 0 and which_pos != len(scan_positions) - 1:
                        continue
                    frames += 1
                    if generate_figure is not None:
                        rot_exc, rot_glow = rotate(exc, -rot), rotate(glow, -rot)
                        c = (0, slice(pad, -pad), slice(pad, -pad))
                        generate_figure(rot, which_pos, obj[c] / obj.max(), rot_exc[c] / max_exc, rot_glow[c] / max_glow,
                                        inst_detector_sig[c] / max_inst_sig, cum_detector_sig[c] / max_cum_sig,
                                        new_signal[c] / max_new_sig, reconstruction[c] / max_reconst,
                                        pulses_delivered, camera_exposures)
            if which_run == 'generate_figures':
                finals[float(rot)] = reconstruction.copy()
    return {'maxima': (max_exc, max_glow, max_inst_sig, max_cum_sig, max_reconst, max_new_sig),
            'reconstructions': finals, 'pulses_delivered': pulses_delivered, 'camera_exposures': camera_exposures,
            'frames': frames, 'positions': scan_positions, 'step': step, 'exc_sep': exc_sep}
