"""Mirror of the imaging simulator of figure_generation/line_sted_figure_3.py.

`simulate_imaging` keeps the reference's signature (:76-85) and semantics (:86-273): four imaging
types (descan point, non-descanned multipoint, descan line, rescan line), the scan-position loop, the
two passes (display maxima first, frames second).  The scan loop of one orientation -- thousands of
independent scan positions -- runs batched on the device in float64 (`rl_fig3_scan`,
csrc/fig3_kernels.hip); rotations are the device's (`rl_rotate_image`); what is left here is the cheap
bookkeeping the reference interleaves with it (where a scan position's value lands in the
reconstruction, which frames are rendered, the display normalisation).

The reference renders each frame with matplotlib and assembles GIF/MP4 files with ImageMagick and
ffmpeg (:274-379): presentation, not built.  Instead `generate_figure` is a callable that receives
exactly the arguments the reference's `generate_figure` receives (:264-273) -- pass the reference's own
function to get its figures -- and the function returns what it computed.
"""
import ctypes
import os

import numpy as np

from . import _lib

IMAGING_TYPES = ('descan_point', 'nondescan_multipoint', 'descan_line', 'rescan_line')


class _Params(ctypes.Structure):      # rl_fig3_params of include/rlsted.h
    _fields_ = [('imaging_type', ctypes.c_int), ('ny', ctypes.c_int), ('nx', ctypes.c_int), ('n_y', ctypes.c_int),
                ('n_x', ctypes.c_int), ('pad', ctypes.c_int), ('step', ctypes.c_int), ('exc_sep', ctypes.c_int),
                ('psf_sigma', ctypes.c_double), ('rescan_scale', ctypes.c_double)]


def _ctx():
    return _lib.Context.get(0)


def rotate(x, angle_degrees):
    """:382-391 -- scipy.ndimage.rotate(axes=(1, 2), mode='nearest', reshape=False), clipped to [0, 1.1 max]."""
    if angle_degrees == 0:
        return x.copy()
    x = _lib.as_f64(x)
    out = np.empty_like(x)
    for z in range(x.shape[0]):
        _lib.check(_lib.lib.rl_rotate_image(_ctx().handle, _lib.ptr(x[z]), _lib.ptr(out[z]), x.shape[1], x.shape[2],
                                            float(angle_degrees), 0))
    return np.clip(out, 0, 1.1 * x.max())


def shift(x, shift):
    """:393-396 for the integer shifts the scan uses: an interpolating spline reproduces its samples,
    so the array moves by whole pixels and zeros move in; clipped to [0, 1.1 max] like the reference."""
    sy, sx = int(shift[-2]), int(shift[-1])
    assert sy == shift[-2] and sx == shift[-1] and all(s == 0 for s in shift[:-2]), 'integer in-plane shifts only'
    out = np.zeros_like(x, dtype=np.float64)
    ny, nx = x.shape[-2:]
    if abs(sy) < ny and abs(sx) < nx:
        out[..., max(sy, 0):ny + min(sy, 0), max(sx, 0):nx + min(sx, 0)] = \
            x[..., max(-sy, 0):ny + min(-sy, 0), max(-sx, 0):nx + min(-sx, 0)]
    return np.clip(out, 0, 1.1 * x.max())


def _gaussian_filter(a, sigma3, truncate):
    a = _lib.as_f64(a)
    out = np.empty_like(a)
    s3 = np.asarray(sigma3, dtype=np.float64)
    _lib.check(_lib.lib.rl_gaussian_filter(_ctx().handle, _lib.ptr(a), _lib.ptr(out), a.shape[0], a.shape[1], a.shape[2],
                                           _lib.ptr(s3), float(truncate)))
    return out


def simulate_imaging(obj, imaging_type, psf_width, R, num_orientations, pulses_per_position, pad,
                     comparison_name='', generate_figure=None, verbose=False):
    """line_sted_figure_3.py:76-273.  Returns a dict: 'maxima' (max_exc, max_glow, max_inst_sig,
    max_cum_sig, max_reconst, max_new_sig :140-141,252-256), 'reconstructions' {orientation in degrees:
    final padded reconstruction}, 'filenames' (the frame names the reference would write, last frame
    repeated as at :259-261), 'pulses_delivered', 'camera_exposures', 'scan_positions'."""
    output_filename = imaging_type + '_'
    if num_orientations > 1:
        output_filename += '%02iangles_' % num_orientations
    output_filename += comparison_name
    if verbose:
        print("\nSimulating:", output_filename)
    obj = np.asarray(obj)
    assert len(obj.shape) == 3 and obj.shape[0] == 1
    assert imaging_type in IMAGING_TYPES
    assert psf_width >= 1
    psf_sigma = psf_width / (2 * np.sqrt(2 * np.log(2)))
    assert R >= 1
    step = int(np.round(psf_width / (4 * R)))                   # 4 scan positions per STED PSF (:102)
    assert num_orientations >= 1 and int(num_orientations) == num_orientations
    assert pad > 0 and int(pad) == pad
    pad = int(pad)
    _, n_y, n_x = obj.shape
    obj = np.pad(obj.astype(np.float64), ((0, 0), (pad, pad), (pad, pad)), 'constant')
    _, ny, nx = obj.shape
    centered_exc = np.zeros(obj.shape)
    exc_sep = 0
    if imaging_type in ('descan_line', 'rescan_line'):
        centered_exc[0, ny // 2, :] = 1
        sted_sigma = (0, psf_sigma / R, 0)
        scan_positions = [(int(y), 0) for y in np.arange(-n_y // 2, n_y // 2 + 1, step)]
    elif imaging_type == 'descan_point':
        centered_exc[0, ny // 2, nx // 2] = 1
        sted_sigma = (0, psf_sigma / R, psf_sigma / R)
        scan_positions = [(int(y), int(x)) for y in np.arange(-n_y // 2, n_y // 2 + 1, step)
                          for x in np.arange(-n_x // 2, n_x // 2 + 1, step)]
        num_orientations = 1
    else:
        exc_sep = int(step * np.round(psf_width * 1.4 / step))
        centered_exc[0, pad:-pad:exc_sep, pad:-pad:exc_sep] = 1
        sted_sigma = (0, psf_sigma / R, psf_sigma / R)
        scan_positions = [(int(y), int(x)) for y in np.arange(0, exc_sep, step) for x in np.arange(0, exc_sep, step)]
        num_orientations = 1
    centered_exc = _gaussian_filter(centered_exc, sted_sigma, truncate=8)      # :139
    max_exc = centered_exc[0, pad:-pad, pad:-pad].max()
    n_pos = len(scan_positions)
    positions = np.ascontiguousarray(scan_positions, dtype=np.int32)
    num_to_skip = max(int(np.round(n_pos / 150)), 1)                            # :258
    display = np.array([p for p in range(n_pos) if p % num_to_skip == 0 or p == n_pos - 1], dtype=np.int32)
    prm = _Params(IMAGING_TYPES.index(imaging_type), ny, nx, n_y, n_x, pad, step, exc_sep, psf_sigma, 1 / (R ** 2 + 1))
    regions = (-(-n_y // exc_sep), -(-n_x // exc_sep)) if exc_sep else (0, 0)
    n_val = nx if imaging_type == 'descan_line' else regions[0] * regions[1]

    def scan(rot_obj, want_frames):
        """rl_fig3_scan for one orientation: per-position scalars / reconstruction values, the detector
        images of the displayed positions, the final rescan image."""
        sc = np.empty((n_pos, 4))
        vals = np.empty((n_pos, n_val)) if n_val else None
        shown = display if want_frames else display[:0]
        frames = np.empty((len(shown), 2, ny, nx)) if len(shown) else None
        cum = np.empty((ny, nx)) if imaging_type == 'rescan_line' else None
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.check(_lib.lib.rl_fig3_scan(
            _ctx().handle, ctypes.cast(ctypes.byref(prm), ctypes.c_void_p), _lib.ptr(_lib.as_f64(rot_obj[0])),
            _lib.ptr(centered_exc[0]), positions.ctypes.data_as(ip), n_pos,
            shown.ctypes.data_as(ip) if len(shown) else None, len(shown), _lib.ptr(sc),
            _lib.ptr(vals) if vals is not None else None, _lib.ptr(frames) if frames is not None else None,
            _lib.ptr(cum) if cum is not None else None))
        return sc, vals, frames, cum

    def write_block(reconstruction, which_pos, sc, vals):
        """Where scan position which_pos lands in the reconstruction (:186-222)."""
        shift_y, shift_x = scan_positions[which_pos]
        if imaging_type == 'descan_line':
            y0 = shift_y + n_y // 2 + pad
            reconstruction[0, y0:y0 + step, :] = vals[which_pos][None, :]
        elif imaging_type == 'descan_point':
            y0, x0 = shift_y + n_y // 2 + pad, shift_x + n_x // 2 + pad
            reconstruction[0, y0:y0 + step, x0:x0 + step] = sc[which_pos, 3]
        elif imaging_type == 'nondescan_multipoint':
            k = 0
            for y_sp in range(pad + shift_y, pad + shift_y + n_y, exc_sep):
                for x_sp in range(pad + shift_x, pad + shift_x + n_x, exc_sep):
                    reconstruction[0, y_sp - step // 2:y_sp - step // 2 + step,
                                   x_sp - step // 2:x_sp - step // 2 + step] = vals[which_pos, k]
                    k += 1

    max_glow = max_inst_sig = max_cum_sig = max_reconst = max_new_sig = 0
    filenames, finals = [], {}
    camera_exposures = pulses_delivered = 0
    for which_run in ('find_maxima', 'generate_figures'):
        camera_exposures, pulses_delivered = 0, 0
        for rot in np.arange(0, 180, 180 / num_orientations)[::-1]:
            if which_run == 'find_maxima' and rot > 0:
                continue
            if verbose:
                print("Orientation:", rot, "degrees")
            rot_obj = rotate(obj, rot)
            sc, vals, frames, cum_final = scan(rot_obj, which_run == 'generate_figures' and generate_figure is not None)
            reconstruction = np.zeros(obj.shape)
            shown = 0
            for which_pos, (shift_y, shift_x) in enumerate(scan_positions):
                pulses_delivered += pulses_per_position
                last_reconstruction = reconstruction.copy() if which_run == 'find_maxima' or which_pos in display else None
                if imaging_type == 'rescan_line':
                    if which_pos == n_pos - 1:                 # the sensor is read out once per orientation (:235-239)
                        reconstruction = cum_final[None].copy()
                        camera_exposures += 1
                else:
                    write_block(reconstruction, which_pos, sc, vals)
                    camera_exposures = 'N/A' if imaging_type == 'descan_point' else camera_exposures + 1
                if which_run == 'find_maxima':
                    new_signal = reconstruction - last_reconstruction
                    max_glow = max(sc[which_pos, 0], max_glow)
                    max_inst_sig = max(sc[which_pos, 1], max_inst_sig)
                    max_cum_sig = max(sc[which_pos, 2], max_cum_sig)
                    max_reconst = max(reconstruction.max(), max_reconst)
                    max_new_sig = max(new_signal.max(), max_new_sig)
                    continue
                if which_pos % num_to_skip != 0 and which_pos != n_pos - 1:
                    continue
                filenames.append(os.path.join(os.getcwd(), os.pardir, os.pardir, 'big_images', 'Figure_3_temp',
                                              imaging_type + '_%03ideg_' % rot + comparison_name + '_%06i.svg' % which_pos))
                if which_pos == n_pos - 1:
                    filenames.extend([filenames[-1]] * 10)
                if generate_figure is not None:
                    exc = shift(centered_exc, (0, shift_y, shift_x))
                    glow = rot_obj * exc
                    rot_exc, rot_glow = rotate(exc, -rot), rotate(glow, -rot)
                    new_signal = reconstruction - last_reconstruction
                    c = (0, slice(pad, -pad), slice(pad, -pad))
                    generate_figure(filenames[-1], obj[c] / obj.max(), rot_exc[c] / max_exc, rot_glow[c] / max_glow,
                                    frames[shown, 0][c[1:]] / max_inst_sig, frames[shown, 1][c[1:]] / max_cum_sig,
                                    new_signal[c] / max_new_sig, reconstruction[c] / max_reconst,
                                    pulses_delivered, camera_exposures)
                shown += 1
            if which_run == 'generate_figures':
                finals[float(rot)] = reconstruction
    return {'maxima': (max_exc, max_glow, max_inst_sig, max_cum_sig, max_reconst, max_new_sig),
            'reconstructions': finals, 'filenames': filenames, 'output_filename': output_filename,
            'pulses_delivered': pulses_delivered, 'camera_exposures': camera_exposures,
            'scan_positions': scan_positions}
