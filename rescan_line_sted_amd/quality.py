"""Reconstruction-quality metrics of the figure-2 harness and of
Deconvolver.record_iteration, on the device (csrc/quality_kernels.hip, float64):

  fourier_error(estimate, true_object)        line_sted_figure_2.py:353-355
  ft_error_history(estimates, true_object)    line_sted_tools.py:539-547
  map_coordinates(image, coordinates)         scipy defaults, as used at :381-384
  error_vs_spatial_frequency(...)             line_sted_figure_2.py:362-390

("ref2:NNN" = line numbers in figure_generation/line_sted_figure_2.py.)
"""
import numpy as np

from ._lib import lib, check, ptr, as_f64
from .psf import _ctx, gaussian_filter


def _fft2_magnitude(x, scale, log1p):
    x = as_f64(x)
    squeeze = x.ndim == 2
    if squeeze:
        x = x[None]
    if x.ndim != 3:
        raise ValueError('expected a 2-D image or a 3-D stack of images')
    out = np.empty_like(x)
    check(lib.rl_fft2_magnitude(_ctx().handle, ptr(x), x.shape[0], x.shape[1], x.shape[2],
                                float(scale), int(log1p), ptr(out)))
    return out[0] if squeeze else out


def fourier_error(estimate, true_object):
    """ref2:353-355: abs(fftshift(fftn(x - true_object))) / prod(true_object.shape), 2-D arrays
    (or stacks of them: one 2-D transform per leading index)."""
    d = as_f64(np.asarray(estimate, dtype=np.float64) - np.asarray(true_object, dtype=np.float64))
    return _fft2_magnitude(d, 1.0 / (d.shape[-2] * d.shape[-1]), False)


def ft_error_history(estimates, true_object):
    """line_sted_tools.py:539-547: log(1 + abs(fftshift(fft2(estimate_k - true_object)))) for a
    stack of saved estimates (k, ny, nx); a 2-D input is treated as one estimate."""
    d = np.asarray(estimates, dtype=np.float64) - np.asarray(true_object, dtype=np.float64)
    if d.ndim == 2:
        d = d.reshape(1, d.shape[0], d.shape[1])
    return _fft2_magnitude(as_f64(d), 1.0, True)


def map_coordinates(image, coordinates):
    """scipy.ndimage.map_coordinates(image, coordinates) for a 2-D image with scipy's defaults
    (order=3, mode='constant', cval=0.0, prefilter=True); coordinates = (2, ...) array-like."""
    image = as_f64(image)
    if image.ndim != 2:
        raise ValueError('expected a 2-D image')
    c = np.asarray(coordinates, dtype=np.float64)
    if c.shape[0] != 2:
        raise ValueError('coordinates must have shape (2, ...)')
    ys, xs = as_f64(c[0].ravel()), as_f64(c[1].ravel())
    out = np.empty(ys.size)
    check(lib.rl_spline_sample(_ctx().handle, ptr(image), image.shape[0], image.shape[1], ptr(ys), ptr(xs),
                               int(ys.size), ptr(out)))
    return out.reshape(c.shape[1:])


def error_vs_spatial_frequency(estimate, true_object, angle_degrees=0.0, radius=0.3, samples=1000,
                               smooth=True):
    """ref2:362-390: the Fourier error along a line through the centre of the shifted spectrum
    at `angle_degrees`, half length `radius` (fraction of the image size), `samples` points,
    cubic-spline interpolated and (smooth=True) Gaussian filtered with sigma = samples / 80.
    The figure plots angle 0 ("best") for point and line STED and 90/num_angles ("worst")."""
    fe = fourier_error(estimate, true_object)
    if fe.ndim != 2:
        raise ValueError('expected 2-D estimate and object')
    n_x, n_y = fe.shape                                    # ref2:363 names the axes this way round
    ang = angle_degrees * 2 * np.pi / 360
    x0, x1 = (0.5 + radius * np.array((-np.cos(ang), np.cos(ang)))) * n_x
    y0, y1 = (0.5 + radius * np.array((-np.sin(ang), np.sin(ang)))) * n_y
    xy = np.vstack((np.linspace(x0, x1, samples), np.linspace(y0, y1, samples)))
    z = map_coordinates(np.ascontiguousarray(np.transpose(fe)), xy)             # ref2:381
    if not smooth:
        return z
    return gaussian_filter(z.reshape(1, 1, -1), (0, 0, samples / 80))[0, 0]     # ref2:387
