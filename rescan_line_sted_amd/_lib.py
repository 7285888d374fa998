"""ctypes binding of librlsted.so (C ABI: include/rlsted.h).

The library is the product: if it is missing or fails to load, importing this
module raises -- there is no CPU fallback for the device path.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RLSTED_LIB') or os.path.join(_HERE, '_lib', 'librlsted.so')   # RLSTED_LIB: a build variant (development)

RL_F32, RL_F64 = 0, 1
RNG_NONE, RNG_PHILOX = 0, 1
DTYPES = {'f32': RL_F32, 'float32': RL_F32, 'f64': RL_F64, 'float64': RL_F64}

# name -> (restype, argtypes); exactly the symbols include/rlsted.h declares
_c = ctypes
_vp, _i, _dp = _c.c_void_p, _c.c_int, _c.POINTER(_c.c_double)
PROTOTYPES = {
    'rl_last_error': (_c.c_char_p, []),
    'rl_version': (_i, []),
    'rl_device_count': (_i, [_c.POINTER(_i)]),
    'rl_ctx_create': (_i, [_i, _c.POINTER(_vp)]),
    'rl_ctx_destroy': (_i, [_vp]),
    'rl_ctx_synchronize': (_i, [_vp]),
    'rl_fft_length_for': (_i, [_i]),
    'rl_deconv_create': (_i, [_vp, _dp, _i, _i, _i, _i, _i, _i, _i, _c.POINTER(_vp)]),
    'rl_deconv_destroy': (_i, [_vp]),
    'rl_deconv_info': (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_c.c_size_t)]),
    'rl_deconv_set_object': (_i, [_vp, _dp, _dp]),
    'rl_deconv_simulate': (_i, [_vp, _i, _c.c_uint64]),
    'rl_deconv_simulate_keyed': (_i, [_vp, _i, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint32)]),
    'rl_deconv_set_measurement': (_i, [_vp, _dp]),
    'rl_deconv_iterate': (_i, [_vp, _i]),
    'rl_deconv_reset_estimate': (_i, [_vp]),
    'rl_deconv_set_estimate': (_i, [_vp, _dp]),
    'rl_deconv_get_object': (_i, [_vp, _dp]),
    'rl_deconv_get_noiseless': (_i, [_vp, _dp]),
    'rl_deconv_get_measurement': (_i, [_vp, _dp]),
    'rl_deconv_get_estimate': (_i, [_vp, _dp]),
    'rl_deconv_get_normalization': (_i, [_vp, _dp]),
    'rl_forward': (_i, [_vp, _dp, _dp]),
    'rl_adjoint': (_i, [_vp, _dp, _dp, _i]),
    'rl_deconv_last_ms': (_i, [_vp, _dp, _dp]),
    'rl_deconv_bench_cycles': (_i, [_vp, _i, _i, _i, _c.c_uint64, _dp]),
    'rl_deconv_time_kernels': (_i, [_vp, _i, _dp]),
    'rl_deconv_time_cycle': (_i, [_vp, _i, _i, _c.c_uint64, _dp, _dp, _dp]),
    'rl_deconv_device_ptr': (_i, [_vp, _i, _c.POINTER(_vp), _c.POINTER(_c.c_size_t), _c.POINTER(_i)]),
    'rl_host_alloc': (_i, [_c.c_size_t, _c.POINTER(_vp)]),
    'rl_host_free': (_i, [_vp]),
    'rl_deconv_strategy': (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    'rl_deconv_unresolved': (_i, [_vp, _c.POINTER(_c.c_uint64), _i]),
    'rl_deconv_dims': (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    'rl_batch_run': (_i, [_vp, _vp, _i, _i, _i, _dp]),
    'rl_batch_submit': (_i, [_vp, _vp, _i, _i, _i, _vp, _i]),
    'rl_device_alloc': (_i, [_vp, _c.c_size_t, _c.POINTER(_vp)]),
    'rl_device_free': (_i, [_vp, _vp]),
    'rl_device_download': (_i, [_vp, _vp, _i, _c.c_size_t, _dp]),
    'rl_device_upload': (_i, [_vp, _vp, _i, _c.c_size_t, _dp]),
    'rl_comm_gather_device': (_i, [_vp, _vp, _c.POINTER(_c.c_size_t), _i, _i, _vp]),
    'rl_comm_bcast_host': (_i, [_vp, _dp, _c.c_size_t, _i]),
    'rl_comm_unique_id': (_i, [_vp]),
    'rl_comm_create': (_i, [_vp, _i, _i, _vp, _c.POINTER(_vp)]),
    'rl_comm_destroy': (_i, [_vp]),
    'rl_comm_info': (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i)]),
    'rl_comm_barrier': (_i, [_vp]),
    'rl_comm_allreduce_max': (_i, [_vp, _dp]),
    'rl_gather': (_i, [_vp, _vp, _i, _i, _c.POINTER(_i), _dp]),
    'rl_gather_device': (_i, [_vp, _vp, _i, _i, _c.POINTER(_i), _c.POINTER(_vp), _c.POINTER(_c.c_size_t), _c.POINTER(_i)]),
    'rl_comm_gather_host': (_i, [_vp, _dp, _c.POINTER(_c.c_size_t), _i, _dp]),
    'rl_rotate_image': (_i, [_vp, _dp, _dp, _i, _i, _c.c_double, _i]),
    'rl_fig3_scan': (_i, [_vp, _vp, _dp, _dp, _c.POINTER(_i), _i, _c.POINTER(_i), _i, _dp, _dp, _dp, _dp]),
    'rl_gauss_fit': (_i, [_dp, _i, _dp, _c.POINTER(_i)]),
    'rl_gaussian_filter': (_i, [_vp, _dp, _dp, _i, _i, _i, _dp, _c.c_double]),
    'rl_psf_generate': (_i, [_vp, _i, _i, _i, _c.c_double, _c.c_double, _c.c_double, _i, _dp, _dp, _dp]),
    'rl_psf_generate_line_extras': (_i, [_vp, _i, _i, _c.c_double, _c.c_double, _c.c_double, _i, _dp, _dp]),
    'rl_rotate_psf': (_i, [_vp, _dp, _dp, _i, _i, _c.c_double]),
    'rl_rotate_psf_stack': (_i, [_vp, _dp, _dp, _i, _i, _i, _c.c_double]),
    'rl_psf_report': (_i, [_vp, _i, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _dp, _dp]),
    'rl_psf_report_batch': (_i, [_vp, _i, _dp, _dp, _c.POINTER(_dp)]),
    'rl_fft2_magnitude': (_i, [_vp, _dp, _i, _i, _i, _c.c_double, _i, _dp]),
    'rl_spline_sample': (_i, [_vp, _dp, _i, _i, _dp, _dp, _i, _dp]),
}


class RlstedError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'librlsted.so is not built (%s). Run `python -m rescan_line_sted_amd._build` '
            '(needs hipcc); there is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI drifted
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise RlstedError('librlsted error %d: %s' % (rc, lib.rl_last_error().decode()))


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return a.ctypes.data_as(_dp)


class _Pinned:
    """One rl_host_alloc block, exposed to numpy through __array_interface__ (numpy keeps this object as the
    base of every view, so the block is freed when the last view dies)."""

    def __init__(self, nbytes):
        self.p = _vp()
        check(lib.rl_host_alloc(nbytes, ctypes.byref(self.p)))
        self.__array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (self.p.value, False), 'version': 3}

    def __del__(self):
        if getattr(self, 'p', None) and lib is not None:
            lib.rl_host_free(self.p)
            self.p = None


def pinned_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (rl_host_alloc): pass it as the object / measurement, or as
    `out=` of the getters, and it crosses PCIe by direct DMA."""
    shape = tuple(int(x) for x in np.atleast_1d(shape))
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    return np.asarray(_Pinned(max(n, dt.itemsize)))[:n].view(dt).reshape(shape)


def device_count():
    n = _i(0)
    rc = lib.rl_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


class Context:
    """rl_ctx: a stream and the twiddle tables of one GPU.  One per GPU for ordinary use (stream 0); a sweep deals its
    plans to a few more of them (`stream` 1, 2, ...) so that the launches of independent plans overlap on the device."""
    _cache = {}

    def __init__(self, device=0):
        self.handle = _vp()
        check(lib.rl_ctx_create(device, ctypes.byref(self.handle)))
        self.device = device

    @classmethod
    def get(cls, device=0, stream=0):
        if (device, stream) not in cls._cache:
            cls._cache[(device, stream)] = Context(device)
        return cls._cache[(device, stream)]

    def synchronize(self):
        check(lib.rl_ctx_synchronize(self.handle))


def common_psf_shape(psfs):
    """The reference convolves with each PSF on its own (fftconvolve(x, psf, mode='same'), ref:573,585), so the
    PSFs of one Deconvolver may differ in shape; a plan holds one (V, py, px) stack.  Every PSF is embedded in
    zeros of a common odd shape with its 'same'-mode centre (n - 1) // 2 on the common centre: the convolution
    is unchanged."""
    if all(p.shape == psfs[0].shape for p in psfs):
        return psfs
    half = [0, 0]
    for p in psfs:
        for ax in (0, 1):
            n = p.shape[1 + ax]
            c = (n - 1) // 2
            half[ax] = max(half[ax], c, n - 1 - c)
    out = []
    for p in psfs:
        q = np.zeros((1, 2 * half[0] + 1, 2 * half[1] + 1))
        oy, ox = half[0] - (p.shape[1] - 1) // 2, half[1] - (p.shape[2] - 1) // 2
        q[0, oy:oy + p.shape[1], ox:ox + p.shape[2]] = p[0]
        out.append(q)
    return out


class DeconvPlan:
    """rl_deconv: `batch` frames sharing one PSF set and one image shape."""

    def __init__(self, psfs, batch, ny, nx, dtype='f32', device=0, stream=0):
        psfs = [as_f64(p) for p in psfs]
        for p in psfs:
            if p.ndim != 3 or p.shape[0] != 1:
                raise ValueError('PSFs must have shape (1, py, px); got %s' % [q.shape for q in psfs])
        psfs = common_psf_shape(psfs)
        self.ctx = Context.get(device, stream)
        self.psf_stack = as_f64(np.concatenate(psfs, axis=0))
        self.V, self.py, self.px = self.psf_stack.shape
        self.B, self.ny, self.nx = int(batch), int(ny), int(nx)
        self.dtype = dtype
        self.handle = _vp()
        check(lib.rl_deconv_create(self.ctx.handle, ptr(self.psf_stack), self.V, self.py, self.px,
                                   self.B, self.ny, self.nx, DTYPES[dtype], ctypes.byref(self.handle)))

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h:
            lib.rl_deconv_destroy(h)
            self.handle = None

    def info(self):
        ly, lx, pitch, nbytes = _i(), _i(), _i(), _c.c_size_t()
        check(lib.rl_deconv_info(self.handle, ctypes.byref(ly), ctypes.byref(lx), ctypes.byref(pitch),
                                 ctypes.byref(nbytes)))
        return {'ly': ly.value, 'lx': lx.value, 'pitch': pitch.value, 'device_bytes': nbytes.value}

    def strategy(self):
        a, b, c, d = _i(), _i(), _i(), _i()
        check(lib.rl_deconv_strategy(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
        return {'separable': a.value == 1, 'direct_stencil': a.value == 2, 'real_psf_spectrum': bool(b.value),
                'split_column_pass': bool(c.value), 'frame_pairs': bool(d.value)}

    def set_object(self, obj, total_brightness=None):
        obj = as_f64(obj).reshape(self.B, self.ny, self.nx)
        tb = None
        if total_brightness is not None:
            tb = as_f64(np.broadcast_to(np.asarray(total_brightness, dtype=np.float64), (self.B,)))
        check(lib.rl_deconv_set_object(self.handle, ptr(obj), ptr(tb) if tb is not None else None))

    def simulate(self, seed=0, rng=RNG_PHILOX):
        check(lib.rl_deconv_simulate(self.handle, rng, _c.c_uint64(seed)))

    def unresolved(self, reset=False):
        """Lanes of the ratio launches that met a prediction H(estimate) <= 0 inside the image since the plan was created (or the
        counter last reset): 0 on data the plan's arithmetic resolves.  An f32 plan that counts -- sparse emitters on a black
        background, a PSF narrower than the gaps -- stays finite but is no longer within 1e-5 of the float64 result
        (include/rlsted.h rl_deconv_unresolved)."""
        n = _c.c_uint64(0)
        check(lib.rl_deconv_unresolved(self.handle, _c.byref(n), 1 if reset else 0))
        return int(n.value)

    def set_measurement(self, noisy):
        noisy = as_f64(noisy).reshape(self.B, self.V, self.ny, self.nx)
        check(lib.rl_deconv_set_measurement(self.handle, ptr(noisy)))

    def iterate(self, k=1):
        check(lib.rl_deconv_iterate(self.handle, int(k)))

    def reset_estimate(self):
        check(lib.rl_deconv_reset_estimate(self.handle))

    def set_estimate(self, est):
        est = as_f64(est).reshape(self.B, self.ny, self.nx)
        check(lib.rl_deconv_set_estimate(self.handle, ptr(est)))

    def _get(self, fn, shape, out=None):
        if out is None:
            out = np.empty(shape, dtype=np.float64)
        elif out.dtype != np.float64 or not out.flags.c_contiguous or out.size != int(np.prod(shape)):
            raise ValueError('out must be a C-contiguous float64 array of %d elements' % int(np.prod(shape)))
        check(fn(self.handle, ptr(out)))
        return out if out.shape == tuple(shape) else out.reshape(shape)

    def object(self, out=None):
        return self._get(lib.rl_deconv_get_object, (self.B, self.ny, self.nx), out)

    def noiseless(self, out=None):
        return self._get(lib.rl_deconv_get_noiseless, (self.B, self.V, self.ny, self.nx), out)

    def measurement(self, out=None):
        return self._get(lib.rl_deconv_get_measurement, (self.B, self.V, self.ny, self.nx), out)

    def estimate(self, out=None):
        """out: a float64 array to fill (e.g. pinned_empty(...)) instead of a fresh allocation."""
        return self._get(lib.rl_deconv_get_estimate, (self.B, self.ny, self.nx), out)

    def normalization(self):
        return self._get(lib.rl_deconv_get_normalization, (self.ny, self.nx))

    def forward(self, x):
        x = as_f64(x).reshape(self.B, self.ny, self.nx)
        out = np.empty((self.B, self.V, self.ny, self.nx), dtype=np.float64)
        check(lib.rl_forward(self.handle, ptr(x), ptr(out)))
        return out

    def adjoint(self, y, normalize=True):
        y = as_f64(y).reshape(self.B, self.V, self.ny, self.nx)
        out = np.empty((self.B, self.ny, self.nx), dtype=np.float64)
        check(lib.rl_adjoint(self.handle, ptr(y), ptr(out), 1 if normalize else 0))
        return out

    def last_ms(self):
        a, b = _c.c_double(), _c.c_double()
        check(lib.rl_deconv_last_ms(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return {'iterate_ms': a.value, 'simulate_ms': b.value}

    def simulate_keyed(self, seeds, image_ids, rng=RNG_PHILOX):
        """Poisson noise with a Philox key per frame: frame f draws with seed seeds[f] and image
        index image_ids[f] * n_psf + view, whatever batch it sits in."""
        seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.B,)))
        ids = np.ascontiguousarray(np.broadcast_to(np.asarray(image_ids, dtype=np.uint32), (self.B,)))
        check(lib.rl_deconv_simulate_keyed(self.handle, rng, seeds.ctypes.data_as(_c.POINTER(_c.c_uint64)),
                                           ids.ctypes.data_as(_c.POINTER(_c.c_uint32))))

    class _Task(_c.Structure):      # rl_task of include/rlsted.h
        _fields_ = [('object', _dp), ('total_brightness', _c.c_double), ('seed', _c.c_uint64), ('image_id', _c.c_uint32)]

    def batch_run(self, objects, total_brightness, seeds, image_ids, iterations, rng=RNG_PHILOX, fetch=True):
        """rl_batch_run: one simulate + `iterations` x iterate cycle per task, any number of tasks, in
        chunks of the plan's batch.  objects: (n, ny, nx); seeds / image_ids / total_brightness: scalars
        or n values.  Returns the (n, ny, nx) estimates (fetch=False: None, the last chunk stays on the
        device)."""
        objects = as_f64(objects).reshape(-1, self.ny, self.nx)
        n = objects.shape[0]
        tb = np.broadcast_to(np.asarray(0.0 if total_brightness is None else total_brightness, dtype=np.float64), (n,))
        seeds = np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (n,))
        ids = np.broadcast_to(np.asarray(image_ids, dtype=np.uint32), (n,))
        tasks = (self._Task * n)()
        for i in range(n):
            tasks[i].object = objects[i].ctypes.data_as(_dp)
            tasks[i].total_brightness = float(tb[i])
            tasks[i].seed = int(seeds[i])
            tasks[i].image_id = int(ids[i])
        out = np.empty((n, self.ny, self.nx), dtype=np.float64) if fetch else None
        check(lib.rl_batch_run(self.handle, _c.cast(tasks, _vp), n, int(iterations), rng, ptr(out) if fetch else None))
        return out

    def batch_submit(self, objects, total_brightness, seeds, image_ids, iterations, dev_out, out_dtype='f32', rng=RNG_PHILOX):
        """rl_batch_submit: the same cycle per task, ENQUEUED -- returns once the objects are staged; the estimates go to device
        memory at `dev_out` (a ctypes.c_void_p / address: n tasks x ny x nx elements of out_dtype, unpadded; None: nowhere).
        Synchronise the context before reading them."""
        n = len(objects)
        tb = np.broadcast_to(np.asarray(0.0 if total_brightness is None else total_brightness, dtype=np.float64), (n,))
        seeds = np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (n,))
        ids = np.broadcast_to(np.asarray(image_ids, dtype=np.uint32), (n,))
        objs = [as_f64(o).reshape(self.ny, self.nx) for o in objects]
        tasks = (self._Task * n)()
        for i in range(n):
            tasks[i].object = objs[i].ctypes.data_as(_dp)
            tasks[i].total_brightness = float(tb[i])
            tasks[i].seed = int(seeds[i])
            tasks[i].image_id = int(ids[i])
        check(lib.rl_batch_submit(self.handle, _c.cast(tasks, _vp), n, int(iterations), rng, dev_out, DTYPES[out_dtype]))

    def bench_cycles(self, k, reps, rng=RNG_PHILOX, seed=0):
        ms = _c.c_double()
        check(lib.rl_deconv_bench_cycles(self.handle, int(k), int(reps), rng, _c.c_uint64(seed), ctypes.byref(ms)))
        return ms.value

    BUFFERS = {'estimate': 0, 'measurement': 1, 'noiseless': 2, 'object': 3}

    def device_array(self, which='estimate'):
        """Zero-copy view of a plan buffer as an object with __cuda_array_interface__
        (anything that speaks that protocol can wrap it); the plan keeps ownership."""
        from .sharding import DeviceArray
        idx = self.BUFFERS[which]
        p, n, dt = _vp(), _c.c_size_t(), _i()
        check(lib.rl_deconv_device_ptr(self.handle, idx, ctypes.byref(p), ctypes.byref(n), ctypes.byref(dt)))
        shape = (self.B, self.ny, self.nx) if idx in (0, 3) else (self.B, self.V, self.ny, self.nx)
        self.ctx.synchronize()
        return DeviceArray(p.value, shape, '<f4' if dt.value == RL_F32 else '<f8', self)

    KERNEL_NAMES = ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE', 'rowpass_FWD', 'poisson')

    CYCLE_KERNELS = ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE', 'rowpass_FWD', 'rowpass_INV', 'poisson')

    def time_cycle(self, k, rng=RNG_PHILOX, seed=0):
        """rl_deconv_time_cycle: {kernel: (average ms per launch, launches)} of one whole cycle measured with
        HIP events on the streams the launches go to, and the frames an RL launch covers."""
        avg, cnt, fpl = np.zeros(8), np.zeros(8), _c.c_double()
        check(lib.rl_deconv_time_cycle(self.handle, int(k), rng, _c.c_uint64(seed), ptr(avg), ptr(cnt), ctypes.byref(fpl)))
        return {n: (float(a), int(c)) for n, a, c in zip(self.CYCLE_KERNELS, avg, cnt) if c > 0}, int(fpl.value)

    def time_kernels(self, reps=20):
        out = np.zeros(7, dtype=np.float64)
        check(lib.rl_deconv_time_kernels(self.handle, int(reps), ptr(out)))
        kt = dict(zip(self.KERNEL_NAMES, out[:6].tolist()))
        kt['frames_per_rl_launch'] = int(out[6])      # the RL loop works through the batch in slices
        return kt

