"""Drop-in mirror of the reference module figure_generation/line_sted_tools.py.

Same public names, arguments, return-dict keys and error behaviour as the
reference (citations "ref:NNN" are line numbers in that file); the numerical
work runs in hand-written HIP kernels on an MI355X through librlsted.so.

    psf_report, generate_psfs, tune_psf, get_width     -> psf.py (device PSF kernels)
    Deconvolver (H, H_t, create_data_from_object, iterate, record_*)
    logarithmic_progress

plus batch-oriented helpers that the reference lacks (`simulate`,
`deconvolve`), which process many independent frames per launch.

Precision: the reference computes in float64.  `Deconvolver(dtype=...)` picks
the device arithmetic: 'f64' (default for this class: bit-level agreement with
the reference to ~1e-12) or 'f32' (the BASELINE throughput mode, <=1e-5
normwise at 20 iterations).  Environment variable RLSTED_DTYPE overrides the
default.
"""
import os
import time

import numpy as np

from . import _lib
from ._lib import DeconvPlan, RNG_NONE, RNG_PHILOX

_DEFAULT_DTYPE = os.environ.get('RLSTED_DTYPE', 'f64')
_DEFAULT_DEVICE = int(os.environ.get('RLSTED_DEVICE', '0'))


def _np_tif():
    from . import np_tif
    return np_tif


class Deconvolver:
    """ref:478-594.  One PSF list ("views"), measurements as Python lists of
    (nz, ny, nx) float64 arrays, estimate updated in place by iterate()."""

    def __init__(self, psfs, output_prefix=None, verbose=True, dtype=None,
                 device=None, rng='numpy'):
        """'psfs' is a list of numpy arrays, one for each PSF (ref:479-494).
        dtype/device/rng are extensions: device arithmetic type, GPU index and
        the Poisson generator ('numpy' = the reference's np.random.poisson on
        the host, 'philox' = counter-based generator on the device)."""
        self.psfs = list(psfs)
        if output_prefix is None:
            output_prefix = os.getcwd()
        if not os.path.exists(os.path.dirname(output_prefix)):
            os.mkdir(os.path.dirname(output_prefix))
        self.output_prefix = output_prefix
        self.verbose = verbose
        self.num_iterations = 0
        self.saved_iterations = []
        self.estimate_history = []
        self.dtype = dtype or _DEFAULT_DTYPE
        self.device = _DEFAULT_DEVICE if device is None else device
        self.rng = rng
        for p in self.psfs:
            if np.ndim(p) != 3:
                raise NotImplementedError('PSFs must be 3-D arrays (pz, py, px); got %s' % (np.shape(p),))
        self._plan = None
        self._aux_plans = {}            # H / H_t on shapes other than the data's (the RL state stays untouched)
        self._estimate = None
        self._estimate_stale = False    # the device holds a newer estimate than self._estimate
        self._estimate_push = False     # self._estimate must go to the device before the next iteration
        return None

    # ---- device plan management -------------------------------------------
    def _plan_psfs(self, nz):
        """The 2-D PSFs of a device plan for data of nz slices.  The reference's fftconvolve is n-dimensional (ref:574,586):
        out[z] = sum_k conv2d(x[z + c - k], psf[k]), c = (pz - 1) // 2 ('same' along z too).  Every PSF its scripts build is
        (1, n, n) -- axis 0 is then a batch of independent slices, which is what a plan's frames are.  A PSF with depth
        couples the slices; on ONE slice only its plane k = c meets the data, so the plan takes that plane.  More than one
        slice with such a PSF is not built."""
        out = []
        for p in self.psfs:
            pz = np.shape(p)[0]
            if pz == 1:
                out.append(p)
            elif nz == 1:
                c = (pz - 1) // 2
                out.append(np.asarray(p)[c:c + 1])
            else:
                raise NotImplementedError(
                    'a PSF with depth (shape %s) couples the %d z slices of the data: the device path convolves slice by slice '
                    '(PSFs of shape (1, py, px), or single-slice data)' % (np.shape(p), nz))
        return out

    def _plan_for(self, shape):
        """The plan that holds the data (object, measurement, estimate)."""
        nz, ny, nx = shape
        p = self._plan
        if p is None or (p.B, p.ny, p.nx) != (nz, ny, nx):
            if p is not None and self._estimate_stale:      # keep what the old plan computed
                self._estimate, self._estimate_stale = p.estimate(), False
            self._plan = DeconvPlan(self._plan_psfs(nz), nz, ny, nx, dtype=self.dtype, device=self.device)
            self._measurement_on_device = False             # the host copy is pushed again when needed
            self._estimate_push = self._estimate is not None and np.shape(self._estimate) == (nz, ny, nx)
            if hasattr(self, 'H_t_normalization'):
                del self.H_t_normalization
        return self._plan

    def _operator_plan(self, shape):
        """H / H_t accept arrays of any shape in the reference (ref:567-594) without touching the
        deconvolver's state: other shapes than the data's get a plan of their own."""
        nz, ny, nx = shape
        p = self._plan
        if p is None:
            return self._plan_for(shape)
        if (p.B, p.ny, p.nx) == (nz, ny, nx):
            return p
        key = (nz, ny, nx)
        if key not in self._aux_plans:
            self._aux_plans = {key: DeconvPlan(self._plan_psfs(nz), nz, ny, nx, dtype=self.dtype, device=self.device)}
        return self._aux_plans[key]

    # ---- data ---------------------------------------------------------------
    def create_data_from_object(self, obj, total_brightness=None, random_seed=None):
        assert len(obj.shape) == 3                      # ref:502
        assert obj.dtype == np.float64                  # ref:503
        if self._estimate_stale and self._plan is not None:   # fetch before the plan's measurement changes
            self._fetch_estimate()
        plan = self._plan_for(obj.shape)
        plan.set_object(obj, None if total_brightness is None
                        else self._brightness(obj, total_brightness))
        self.true_object = plan.object()
        noiseless = plan.noiseless()                    # (nz, V, ny, nx)
        self.noiseless_measurement = [np.ascontiguousarray(noiseless[:, v]) for v in range(plan.V)]
        if self.rng == 'numpy':
            if random_seed is not None:
                np.random.seed(random_seed)             # ref:508-509 (global state, like the reference)
            self.noisy_measurement = [np.random.poisson(m) + 1e-9     # ref:510
                                      for m in self.noiseless_measurement]
            plan.set_measurement(np.stack(self.noisy_measurement, axis=1))
        elif self.rng == 'philox':
            plan.simulate(seed=0 if random_seed is None else int(random_seed), rng=RNG_PHILOX)
            noisy = plan.measurement()
            self.noisy_measurement = [np.ascontiguousarray(noisy[:, v]) for v in range(plan.V)]
        else:
            raise ValueError("rng must be 'numpy' or 'philox'")
        self._measurement_on_device = True
        if self.num_iterations > 0 and self._estimate is not None:   # the reference keeps self.estimate (ref:521)
            self._estimate_push = np.shape(self._estimate) == (plan.B, plan.ny, plan.nx)
        return None

    @staticmethod
    def _brightness(obj, total_brightness):
        """The reference scales the whole (nz, ny, nx) stack by one factor
        total_brightness / obj.sum() (ref:505-506); the plan scales per frame to
        a per-frame target, so hand it each frame's share of the total."""
        total = obj.sum()
        return np.array([total_brightness * obj[z].sum() / total for z in range(obj.shape[0])])

    def load_data_from_tif(self, filename):
        """ref:514-518.  The reference asserts `shape == 3`, which can never hold
        (dead code there); the intent -- a 3-D stack -- is what is checked here."""
        data = _np_tif().tif_to_array(filename) + 1e-9
        assert len(data.shape) == 3
        assert data.min() >= 0
        self.noisy_measurement = data
        self._measurement_on_device = False
        return None

    # ---- Richardson-Lucy -----------------------------------------------------
    def _push_measurement(self):
        m = self.noisy_measurement
        if isinstance(m, (list, tuple)):
            m = [np.asarray(v) for v in m]
        else:
            # A stack loaded from noisy_measurement.tif: record_data writes the views one after the
            # other along axis 0 (ref:562-564), so with V PSFs axis 0 is (view, slice).
            m, V = np.asarray(m), len(self.psfs)
            if m.shape[0] % V != 0:
                raise ValueError('measurement stack of %d slices for %d PSFs' % (m.shape[0], V))
            m = list(m.reshape((V, m.shape[0] // V) + m.shape[1:]))
        keep = None
        if self.num_iterations > 0 and (self._estimate_stale or self._estimate is not None):
            keep = self.estimate                            # new data, same estimate (ref:521)
        plan = self._plan_for(m[0].shape)
        plan.set_measurement(np.stack(m, axis=1))
        self._measurement_on_device = True
        if keep is not None and np.shape(keep) == (plan.B, plan.ny, plan.nx):
            self._estimate, self._estimate_push = keep, True
        return plan

    def _sync_estimate(self, plan):
        if self._estimate_push:
            plan.set_estimate(self._estimate)
            self._estimate_push = False

    def iterate(self):
        """ref:520-531: estimate *= H_t(measurement / H(estimate))."""
        if not getattr(self, '_measurement_on_device', False):
            self._push_measurement()
        plan = self._plan
        if self.num_iterations == 0:                      # ref:521-522: always from ones (a NEW array there)
            plan.reset_estimate()
            self._estimate_push = False
            self._estimate = None
        self._sync_estimate(plan)
        self.num_iterations += 1
        plan.iterate(1)
        self._estimate_stale = True
        return None

    def iterate_many(self, k):
        """k iterations without returning to Python in between (extension)."""
        if not getattr(self, '_measurement_on_device', False):
            self._push_measurement()
        if self.num_iterations == 0:
            self._plan.reset_estimate()
            self._estimate_push = False
            self._estimate = None
        self._sync_estimate(self._plan)
        self.num_iterations += k
        self._plan.iterate(k)
        self._estimate_stale = True

    def _fetch_estimate(self):
        """The reference updates ONE array in place (`self.estimate *= ...`, ref:531); so does the mirror: the
        host copy is refreshed into the array handed out before (a fresh 2 MB-per-frame array costs more in
        page faults than its PCIe transfer)."""
        plan, buf = self._plan, self._estimate
        if not (isinstance(buf, np.ndarray) and buf.dtype == np.float64 and buf.flags.c_contiguous
                and buf.flags.writeable and buf.shape == (plan.B, plan.ny, plan.nx)):
            buf = None
        self._estimate = plan.estimate(out=buf)
        self._estimate_stale = False
        self._warn_unresolved(plan)

    def _warn_unresolved(self, plan):
        """Once per Deconvolver, when the estimate is fetched: the iterations met predictions H(estimate) <= 0 (a dark region whose
        prediction is below eps * the frame's maximum).  The reference divides by the clamped zero there -- inf, then nan; the
        kernels kept the pixels neutral and every value finite."""
        if getattr(self, '_unresolved_warned', False):
            return
        n = plan.unresolved()
        if n:
            import warnings
            self._unresolved_warned = True
            warnings.warn('%d row segments of the Richardson-Lucy iterations met a prediction H(estimate) <= 0: the reference would '
                          'have divided by zero (inf, nan); these pixels were kept neutral.%s' % (
                              n, ' The float32 plan (RLSTED_DTYPE=f32) cannot resolve the predictions of dark regions through its transforms: '
                              'use the float64 default for such data, or -- PSFs up to about 15 x 15 -- the direct stencil (RLSTED_DIRECT=2).'
                              if self.dtype == 'f32' else ''), RuntimeWarning, stacklevel=3)

    @property
    def estimate(self):
        if self._estimate_stale:
            self._fetch_estimate()
        if self._estimate is None:
            raise AttributeError('estimate')          # like the reference before iterate()
        return self._estimate

    @estimate.setter
    def estimate(self, value):
        """A plain attribute in the reference: after the first iteration `d.estimate = x` makes the next
        iterate() continue from x (the first iterate() always starts from ones, ref:521-522)."""
        self._estimate = np.array(value, dtype=np.float64)
        self._estimate_stale = False
        self._estimate_push = True

    # ---- operators -------------------------------------------------------------
    def H(self, x):
        """ref:567-577: list, one blurred (clamped >= 0) image stack per PSF."""
        x = np.asarray(x, dtype=np.float64)
        plan = self._operator_plan(x.shape)
        out = plan.forward(x)
        return [np.ascontiguousarray(out[:, v]) for v in range(plan.V)]

    def H_t(self, y, normalize=True):
        """ref:579-594."""
        y = [np.asarray(v, dtype=np.float64) for v in y]
        plan = self._operator_plan(y[0].shape)
        if normalize and not hasattr(self, 'H_t_normalization'):
            self.H_t_normalization = plan.normalization().reshape(1, plan.ny, plan.nx) * np.ones((plan.B, 1, 1))
        return plan.adjoint(np.stack(y, axis=1), normalize=normalize)

    # ---- recording (host I/O, ref:533-565) ---------------------------------
    def record_iteration(self, save_tifs=True):
        self.saved_iterations.append(self.num_iterations)
        self.estimate_history.append(self.estimate.copy())
        if save_tifs:
            np_tif = _np_tif()
            eh = np.squeeze(np.concatenate(self.estimate_history, axis=0))
            np_tif.array_to_tif(eh, self.output_prefix + 'estimate_history.tif')

            from .quality import ft_error_history       # ref:539-547, float64 on the device
            np_tif.array_to_tif(ft_error_history(eh, self.true_object),
                                self.output_prefix + 'estimate_FT_error_history.tif')
        return None

    def record_data(self):
        np_tif = _np_tif()
        if hasattr(self, 'psfs'):
            psfs = np.squeeze(np.concatenate(self.psfs, axis=0))
            np_tif.array_to_tif(psfs, self.output_prefix + 'psfs.tif')
        if hasattr(self, 'true_object'):
            np_tif.array_to_tif(self.true_object, self.output_prefix + 'object.tif')
        if hasattr(self, 'noiseless_measurement'):
            nm = np.squeeze(np.concatenate(self.noiseless_measurement, axis=0))
            np_tif.array_to_tif(nm, self.output_prefix + 'noiseless_measurement.tif')
        if hasattr(self, 'noisy_measurement'):
            nm = np.squeeze(np.concatenate(self.noisy_measurement, axis=0))
            np_tif.array_to_tif(nm, self.output_prefix + 'noisy_measurement.tif')
        return None


# ---------------------------------------------------------------------------
# Batch API (extension): many independent frames per launch.  One "frame" is
# one (object, seed) pair imaged through the plan's PSF set.
# ---------------------------------------------------------------------------
def simulate(objects, psfs, total_brightness=None, seed=0, dtype='f32', device=None,
             rng='philox', plan=None):
    """create_data_from_object for a batch: objects (B, ny, nx) float array.
    Returns (plan, noiseless (B,V,ny,nx), noisy (B,V,ny,nx))."""
    objects = np.asarray(objects, dtype=np.float64)
    B, ny, nx = objects.shape
    if plan is None:
        plan = DeconvPlan(psfs, B, ny, nx, dtype=dtype,
                          device=_DEFAULT_DEVICE if device is None else device)
    plan.set_object(objects, total_brightness)
    if rng == 'philox':
        plan.simulate(seed=seed, rng=RNG_PHILOX)
    elif rng == 'none':
        plan.simulate(seed=seed, rng=RNG_NONE)
    elif rng == 'numpy':
        if seed is not None:
            np.random.seed(seed)
        plan.set_measurement(np.random.poisson(plan.noiseless()) + 1e-9)
    else:
        raise ValueError("rng must be 'philox', 'numpy' or 'none'")
    return plan, plan.noiseless(), plan.measurement()


def deconvolve(measurement, psfs, iterations, dtype='f32', device=None, plan=None):
    """K Richardson-Lucy iterations on a batch: measurement (B, V, ny, nx).
    Returns the estimates (B, ny, nx)."""
    measurement = np.asarray(measurement, dtype=np.float64)
    B, V, ny, nx = measurement.shape
    if plan is None:
        plan = DeconvPlan(psfs, B, ny, nx, dtype=dtype,
                          device=_DEFAULT_DEVICE if device is None else device)
    plan.set_measurement(measurement)
    plan.reset_estimate()
    plan.iterate(iterations)
    return plan.estimate()


def _save_points(n):
    """Indices flagged by logarithmic_progress for an n-long iterable: the powers
    of two p with p + 1 < n, and the final index (ref:618-623)."""
    points, p = set(), 1
    while p + 1 < n:
        points.add(p)
        p *= 2
    points.add(n - 1)
    return points


class _ProgressPrinter:
    """Textual progress display of ref:624-651: silent for the first 1.5 s, then a
    65-column star bar that restarts (with a rate line) at every save point."""
    RULER = ("Progress:\n|0%" + " " * 13 + "|" + " " * 15 + "|50%" +
             " " * 12 + "|" + " " * 15 + "|100%")

    def __init__(self, n, points):
        self.n, self.points = n, points
        self.t0 = time.perf_counter()
        self.ruler_shown = False
        self.stars = 0

    def step(self, i):
        elapsed = time.perf_counter() - self.t0
        if elapsed <= 1.5:
            return
        if i in self.points:
            rate = i / elapsed
            print("Iteration %d/%d %0.1fs elapsed, ~%0.1fs remaining, %0.1f iter/s"
                  % (i, self.n - 1, elapsed, (self.n - i) / rate, rate))
            self.ruler_shown, self.stars = False, 0
        if not self.ruler_shown:
            print(self.RULER)
            self.ruler_shown = True
        while self.stars / 65 < i / self.n:
            print("*", end='')
            self.stars += 1
        if (i + 1) in self.points:
            print()


def logarithmic_progress(iterable, verbose=True):
    """ref:596-651.  Generator of (item, flag) pairs; flag is True when the
    item's index is a power of two or the last one -- the moments the reference
    saves results.  An empty iterable yields nothing.  With verbose=True a progress bar appears after 1.5 s."""
    n = len(iterable)
    if n == 0:
        return iterable
    points = _save_points(n)
    printer = _ProgressPrinter(n, points) if verbose else None
    for i, item in enumerate(iterable):
        yield (item, i in points)
        if printer is not None:
            printer.step(i)


def __getattr__(name):
    # PSF generation lives in psf.py; resolved lazily so that importing the
    # Deconvolver does not require it.
    if name in ('psf_report', 'generate_psfs', 'tune_psf', 'get_width'):
        from . import psf
        return getattr(psf, name)
    raise AttributeError(name)
