// psf_kernels.hip -- gfx950 kernels for PSF generation, float64.
//
// Reference: figure_generation/line_sted_tools.py generate_psfs (:168-363).
// PSFs are tiny ((1,n,n), n <= ~133) so these kernels are latency bound; the
// point is that the whole psf_report pipeline stays on the device between the
// two host-side Gaussian fits, with float64 arithmetic in the reference's
// operation order (exact-equality asserts :105-106,120 depend on symmetry).
#include <hip/hip_runtime.h>
#include "psf_kernels.hpp"

namespace rl {

// half-sample symmetric extension (scipy 'reflect'): d c b a | a b c d | d c b a
__device__ __forceinline__ int reflect_index(int i, int n) {
    const int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return m >= n ? period - 1 - m : m;
}

// scipy.ndimage correlate1d, symmetric-kernel branch (call sites :185-213,260,280):
// out[i] = x[i]*w[r] + sum_{j=-r}^{-1} (x[i+j] + x[i-j]) * w[r+j], outermost pair first.
__global__ void k_blur_axis(const double* __restrict__ in, double* __restrict__ out, int nz, int ny, int nx, int axis,
                            const double* __restrict__ w, int radius) {
    const int total = nz * ny * nx;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int x = e % nx, y = (e / nx) % ny, z = e / (nx * ny);
    const int len = axis == 0 ? nz : (axis == 1 ? ny : nx);
    const int pos = axis == 0 ? z : (axis == 1 ? y : x);
    const int stride = axis == 0 ? ny * nx : (axis == 1 ? nx : 1);
    const double* line = in + (e - pos * stride);
    double acc = line[pos * stride] * w[radius];
    for (int j = -radius; j < 0; ++j) {
        const double a = line[reflect_index(pos + j, len) * stride];
        const double b = line[reflect_index(pos - j, len) * stride];
        acc += (a + b) * w[radius + j];
    }
    out[e] = acc;
}

// delta at the centre pixel (kind 0, :183-184) or a one-pixel vertical line (kind 1, :189-190)
__global__ void k_delta(double* out, int ny, int nx, int kind) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const int x = e % nx, y = e / nx;
    out[e] = (x == nx / 2 && (kind == 1 || y == ny / 2)) ? 1.0 : 0.0;
}

// Single-workgroup reduction with wavefront shuffles; op 0 = max, 1 = sum.
// `stride`/`count` select a strided slice (a whole array or one row).
__global__ void __launch_bounds__(1024) k_reduce(const double* __restrict__ in, int count, int stride, int op,
                                                 double* __restrict__ out) {
    __shared__ double part[16];
    double v = op == 0 ? -1.0e308 : 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const double x = in[(size_t)i * stride];
        v = op == 0 ? (x > v ? x : v) : v + x;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(v, off, 64);
        v = op == 0 ? (o > v ? o : v) : v + o;
    }
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (wave == 0) {
        const int nw = (blockDim.x + 63) / 64;
        v = lane < nw ? part[lane] : (op == 0 ? -1.0e308 : 0.0);
        for (int off = 8; off > 0; off >>= 1) {
            const double o = __shfl_down(v, off, 64);
            v = op == 0 ? (o > v ? o : v) : v + o;
        }
        if (lane == 0) *out = v;
    }
}

// exc = g * (exc_b / max g); dep_raw = outer/max(outer) - g/max(g)      (:187,204-206)
__global__ void k_psf_stage1(const double* __restrict__ g, const double* __restrict__ outer, const double* __restrict__ maxes,
                             double exc_b, double* __restrict__ exc, double* __restrict__ dep_raw, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double mg = maxes[0], mo = maxes[1];
    exc[e] = g[e] * (exc_b / mg);
    dep_raw[e] = (outer[e] / mo) - (g[e] / mg);
}

// dep = dep_raw * (dep_b / max dep_raw); fractions; sted                 (:207,226-240)
__global__ void k_psf_stage2(const double* __restrict__ exc, double* __restrict__ dep, const double* __restrict__ maxes,
                             double dep_b, double* __restrict__ exc_frac, double* __restrict__ dep_frac,
                             double* __restrict__ sted, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double d = dep[e] * (dep_b / maxes[2]);
    dep[e] = d;
    const double ef = 1.0 - exp2(-exc[e] / 1.0);
    const double df = exp2(-d / 1.0);
    exc_frac[e] = ef;
    dep_frac[e] = df;
    sted[e] = ef * df;
}

// ---- rescan simulation, :258-310 -------------------------------------------
// The object is a centred delta, so at scan position sp the glow is the single
// value a = sted_row[(2cx - sp) mod nx] at the centre pixel and the blurred glow
// is ((b0[sp] * ry[y]) * rx[x]) with b0 the axis-0 pass of the 3-axis blur.
__device__ __forceinline__ int wrap(int i, int n) {
    int m = i % n;
    return m < 0 ? m + n : m;
}

__global__ void k_rescan_b0(const double* __restrict__ sted_row, const double* __restrict__ w, int radius, int n,
                            double* __restrict__ b0) {
    const int sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp >= n) return;
    const double a = sted_row[wrap(2 * (n / 2) - sp, n)];
    double acc = a * w[radius];                      // axis 0 has length 1: every tap reads the same value
    for (int j = -radius; j < 0; ++j) acc += (a + a) * w[radius + j];
    b0[sp] = acc;
}

// blurred_sp[y][x] for the glow at scan position sp.  ry / rx are the responses of
// the 1-D blur (with its reflect boundary) to a unit impulse at the centre
// row / column; without boundary contact they are just the Gaussian weights.
__device__ __forceinline__ double blurred(const double* b0, const double* ry, const double* rx, int sp, int y, int x) {
    return (b0[sp] * ry[y]) * rx[x];
}

// descanned_signal_cumu[y][sp] = sum_x roll(blurred_sp, c - sp)[y][x]       (:285-288)
__global__ void k_descan(const double* __restrict__ b0, const double* __restrict__ ry, const double* __restrict__ rx,
                         int ny, int nx, double* __restrict__ descan) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const int sp = e % nx, y = e / nx, c = nx / 2;
    double acc = 0.0;
    for (int x = 0; x < nx; ++x) acc += blurred(b0, ry, rx, sp, y, wrap(x - (c - sp), nx));
    descan[e] = acc;
}

// rescanned_signal_cumu[y][X], X < ratio*nx: sum over sp (in order) of the
// instantaneous image rolled to sp*ratio - c in the ratio*nx ring           (:291-298)
__global__ void k_rescan_cumu(const double* __restrict__ b0, const double* __restrict__ ry, const double* __restrict__ rx,
                              int ny, int nx, int ratio, double* __restrict__ cumu) {
    const int rn = ratio * nx;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * rn) return;
    const int X = e % rn, y = e / rn, c = nx / 2;
    double acc = 0.0;
    for (int sp = 0; sp < nx; ++sp) {
        const int xp = wrap(X - (sp * ratio - c), rn);       // column of the un-rolled instantaneous image
        if (xp < nx) acc += blurred(b0, ry, rx, sp, y, wrap(xp - (c - sp), nx));
    }
    cumu[e] = acc;
}

// rescan[y][j] = sum_q roll(cumu, ratio/2)[y][j*ratio + q]                   (:301-309)
__global__ void k_rescan_bin(const double* __restrict__ cumu, int ny, int nx, int ratio, double* __restrict__ rescan) {
    const int rn = ratio * nx;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const int j = e % nx, y = e / nx;
    double acc = 0.0;
    for (int q = 0; q < ratio; ++q) acc += cumu[(size_t)y * rn + wrap(j * ratio + q - ratio / 2, rn)];
    rescan[e] = acc;
}

// ---- cubic B-spline rotation of a PSF, line_sted_figure_2.py:264-272 ------------------
// scipy.ndimage.rotate(order=3, reshape=False, mode='constant'): recursive prefilter
// with mirror boundaries along both axes, then the spline evaluated at the rotated
// coordinates (source coordinates outside [0, n-1] give 0), clipped to [0, 1.1 max].
// One thread filters one line (n <= a few hundred): the recursion is serial by nature.
__global__ void k_spline_prefilter(double* __restrict__ a, int ny, int nx, int axis) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = axis == 0 ? ny : nx, lines = axis == 0 ? nx : ny;
    if (line >= lines || n < 2) return;
    double* c = a + (axis == 0 ? line : (size_t)line * nx);
    const int st = axis == 0 ? nx : 1;
    const double z = -0.26794919243112270647255365849413;          // sqrt(3) - 2
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    for (int i = 0; i < n; ++i) c[(size_t)i * st] *= gain;
    const double zn1 = pow(z, (double)(n - 1));
    double c0 = c[0] + zn1 * c[(size_t)(n - 1) * st], zi = z;
    for (int i = 1; i < n - 1; ++i) {
        c0 += zi * (c[(size_t)i * st] + zn1 * c[(size_t)(n - 1 - i) * st]);
        zi *= z;
    }
    c[0] = c0 / (1.0 - zn1 * zn1);
    for (int i = 1; i < n; ++i) c[(size_t)i * st] += z * c[(size_t)(i - 1) * st];
    c[(size_t)(n - 1) * st] = (z * c[(size_t)(n - 2) * st] + c[(size_t)(n - 1) * st]) * z / (z * z - 1.0);
    for (int i = n - 2; i >= 0; --i) c[(size_t)i * st] = z * (c[(size_t)(i + 1) * st] - c[(size_t)i * st]);
}

__device__ __forceinline__ int mirror_index(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = (i < 0 ? -i : i) % p;
    return i >= n ? p - i : i;
}

__global__ void k_spline_rotate(const double* __restrict__ coef, double* __restrict__ out, int ny, int nx, double c,
                                double s, const double* __restrict__ vmax) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const int ox = e % nx, oy = e / nx;
    const double cy = (ny - 1) * 0.5, cx = (nx - 1) * 0.5;
    double y = c * oy + s * ox + (cy - (c * cy + s * cx));
    double x = -s * oy + c * ox + (cx - (-s * cy + c * cx));
    const double tol = 1e-10;
    double r = 0.0;
    if (y >= -tol && y <= ny - 1 + tol && x >= -tol && x <= nx - 1 + tol) {
        y = fmin(fmax(y, 0.0), (double)(ny - 1));
        x = fmin(fmax(x, 0.0), (double)(nx - 1));
        const double fy = floor(y), fx = floor(x), ty = y - fy, tx = x - fx;
        const double wy[4] = {(1 - ty) * (1 - ty) * (1 - ty) / 6, (3 * ty * ty * ty - 6 * ty * ty + 4) / 6,
                              (-3 * ty * ty * ty + 3 * ty * ty + 3 * ty + 1) / 6, ty * ty * ty / 6};
        const double wx[4] = {(1 - tx) * (1 - tx) * (1 - tx) / 6, (3 * tx * tx * tx - 6 * tx * tx + 4) / 6,
                              (-3 * tx * tx * tx + 3 * tx * tx + 3 * tx + 1) / 6, tx * tx * tx / 6};
        for (int i = 0; i < 4; ++i) {
            const int yy = mirror_index((int)fy - 1 + i, ny);
            for (int j = 0; j < 4; ++j) r += wy[i] * wx[j] * coef[(size_t)yy * nx + mirror_index((int)fx - 1 + j, nx)];
        }
    }
    const double hi = 1.1 * vmax[0];                                  // np.clip(rotated, 0, 1.1 * x.max())
    out[e] = r < 0.0 ? 0.0 : (r > hi ? hi : r);
}


// ---- batched PSF pipeline: the same per-element arithmetic and the same reduction order as the single-set
// kernels above, blockIdx.y (or the job index) selecting the parameter set -- results are bit for bit those
// of rl_psf_report called set by set.
__global__ void __launch_bounds__(1024) k_reduce_jobs(const double* __restrict__ base, double* __restrict__ out_base,
                                                      const ReduceJob* __restrict__ jobs) {
    __shared__ double part[16];
    const ReduceJob jb = jobs[blockIdx.x];
    const double* in = base + jb.off;
    const int op = jb.op;
    double v = op == 0 ? -1.0e308 : 0.0;
    for (int i = threadIdx.x; i < jb.count; i += blockDim.x) {
        const double x = in[(size_t)i * jb.stride];
        v = op == 0 ? (x > v ? x : v) : v + x;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(v, off, 64);
        v = op == 0 ? (o > v ? o : v) : v + o;
    }
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (wave == 0) {
        const int nw = (blockDim.x + 63) / 64;
        v = lane < nw ? part[lane] : (op == 0 ? -1.0e308 : 0.0);
        for (int off = 8; off > 0; off >>= 1) {
            const double o = __shfl_down(v, off, 64);
            v = op == 0 ? (o > v ? o : v) : v + o;
        }
        if (lane == 0) out_base[jb.out] = v;
    }
}

__global__ void k_batch_stage1(double* __restrict__ base, const PsfSetDesc* __restrict__ sets) {
    const PsfSetDesc d = sets[blockIdx.y];
    const int n = d.n * d.n, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double* g = base + d.g;
    const double* outer = g + n;
    const double mg = base[d.gmax], mo = base[d.gmax + 1];
    double* exc = base + d.arrays;
    exc[e] = g[e] * (d.exc_b / mg);
    exc[n + e] = (outer[e] / mo) - (g[e] / mg);      // dep_raw
}
__global__ void k_batch_stage2(double* __restrict__ base, const PsfSetDesc* __restrict__ sets) {
    const PsfSetDesc d = sets[blockIdx.y];
    const int n = d.n * d.n, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double* exc = base + d.arrays;
    double *dep = exc + n, *excf = dep + n, *depf = excf + n, *sted = depf + n;
    const double dv = dep[e] * (d.dep_b / base[d.scal + 2]);
    dep[e] = dv;
    const double ef = 1.0 - exp2(-exc[e] / 1.0);
    const double df = exp2(-dv / 1.0);
    excf[e] = ef;
    depf[e] = df;
    sted[e] = ef * df;
}
__global__ void k_batch_rescan_b0(double* __restrict__ base, const PsfSetDesc* __restrict__ sets) {
    const PsfSetDesc d = sets[blockIdx.y];
    const int sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.type != 1 || sp >= d.n) return;
    const int n = d.n, radius = d.radius;
    const double* sted_row = base + d.arrays + 4 * (size_t)n * n + (size_t)(n / 2) * n;
    const double* w = base + d.w;
    const double a = sted_row[wrap(2 * (n / 2) - sp, n)];
    double acc = a * w[radius];
    for (int j = -radius; j < 0; ++j) acc += (a + a) * w[radius + j];
    base[d.b0 + sp] = acc;
}
__global__ void k_batch_descan(double* __restrict__ base, const PsfSetDesc* __restrict__ sets) {
    const PsfSetDesc d = sets[blockIdx.y];
    const int n = d.n, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.type != 1 || e >= n * n) return;
    const double *b0 = base + d.b0, *ry = base + d.ryx, *rx = ry + n;
    const int sp = e % n, y = e / n, c = n / 2;
    double acc = 0.0;
    for (int x = 0; x < n; ++x) acc += blurred(b0, ry, rx, sp, y, wrap(x - (c - sp), n));
    base[d.arrays + 5 * (size_t)n * n + e] = acc;
}
__global__ void k_batch_rescan_cumu(double* __restrict__ base, const PsfSetDesc* __restrict__ sets) {
    const PsfSetDesc d = sets[blockIdx.y];
    const int n = d.n, rn = d.ratio * n, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.type != 1 || e >= n * rn) return;
    const double *b0 = base + d.b0, *ry = base + d.ryx, *rx = ry + n;
    const int X = e % rn, y = e / rn, c = n / 2;
    double acc = 0.0;
    for (int sp = 0; sp < n; ++sp) {
        const int xp = wrap(X - (sp * d.ratio - c), rn);
        if (xp < n) acc += blurred(b0, ry, rx, sp, y, wrap(xp - (c - sp), n));
    }
    base[d.cumu + e] = acc;
}
__global__ void k_batch_rescan_bin(double* __restrict__ base, const PsfSetDesc* __restrict__ sets) {
    const PsfSetDesc d = sets[blockIdx.y];
    const int n = d.n, rn = d.ratio * n, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.type != 1 || e >= n * n) return;
    const double* cumu = base + d.cumu;
    const int j = e % n, y = e / n;
    double acc = 0.0;
    for (int q = 0; q < d.ratio; ++q) acc += cumu[(size_t)y * rn + wrap(j * d.ratio + q - d.ratio / 2, rn)];
    base[d.arrays + 6 * (size_t)n * n + e] = acc;
}

static inline unsigned nblk(int n) { return (unsigned)((n + 255) / 256); }

hipError_t psf_blur_axis(const double* in, double* out, int nz, int ny, int nx, int axis, const double* w, int radius,
                         hipStream_t s) {
    k_blur_axis<<<nblk(nz * ny * nx), 256, 0, s>>>(in, out, nz, ny, nx, axis, w, radius);
    return hipGetLastError();
}
hipError_t psf_delta(double* out, int ny, int nx, int kind, hipStream_t s) {
    k_delta<<<nblk(ny * nx), 256, 0, s>>>(out, ny, nx, kind);
    return hipGetLastError();
}
hipError_t psf_reduce(const double* in, int count, int stride, int op, double* out, hipStream_t s) {
    k_reduce<<<1, 1024, 0, s>>>(in, count, stride, op, out);
    return hipGetLastError();
}
hipError_t psf_stage1(const double* g, const double* outer, const double* maxes, double exc_b, double* exc,
                      double* dep_raw, int n, hipStream_t s) {
    k_psf_stage1<<<nblk(n), 256, 0, s>>>(g, outer, maxes, exc_b, exc, dep_raw, n);
    return hipGetLastError();
}
hipError_t psf_stage2(const double* exc, double* dep, const double* maxes, double dep_b, double* exc_frac,
                      double* dep_frac, double* sted, int n, hipStream_t s) {
    k_psf_stage2<<<nblk(n), 256, 0, s>>>(exc, dep, maxes, dep_b, exc_frac, dep_frac, sted, n);
    return hipGetLastError();
}
hipError_t psf_rescan(const double* sted_row, const double* w, int radius, const double* ry, const double* rx, int ny,
                      int nx, int ratio, double* b0, double* cumu, double* descan, double* rescan, hipStream_t s) {
    k_rescan_b0<<<nblk(nx), 256, 0, s>>>(sted_row, w, radius, nx, b0);
    k_descan<<<nblk(ny * nx), 256, 0, s>>>(b0, ry, rx, ny, nx, descan);
    k_rescan_cumu<<<nblk(ny * ratio * nx), 256, 0, s>>>(b0, ry, rx, ny, nx, ratio, cumu);
    k_rescan_bin<<<nblk(ny * nx), 256, 0, s>>>(cumu, ny, nx, ratio, rescan);
    return hipGetLastError();
}

// in: [ny][nx] device; work: [ny][nx] device scratch (receives the spline coefficients);
// vmax: device scalar scratch
// max_count: the values (from `in` on) whose maximum sets the clip level -- the plane's ny * nx, or a whole stack's (fig2:271 clips
// at 1.1 * the ARRAY's maximum); 0: *vmax is already there
hipError_t psf_spline_rotate(const double* in, double* work, double* out, double* vmax, int ny, int nx,
                             double degrees, hipStream_t s, int max_count) {
    hipError_t e = hipMemcpyAsync(work, in, (size_t)ny * nx * sizeof(double), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    if (max_count > 0) k_reduce<<<1, 1024, 0, s>>>(in, max_count, 1, 0, vmax);
    k_spline_prefilter<<<nblk(nx), 256, 0, s>>>(work, ny, nx, 0);
    k_spline_prefilter<<<nblk(ny), 256, 0, s>>>(work, ny, nx, 1);
    const double th = degrees * 0.017453292519943295769236907684886;
    k_spline_rotate<<<nblk(ny * nx), 256, 0, s>>>(work, out, ny, nx, cos(th), sin(th), vmax);
    return hipGetLastError();
}


hipError_t psf_reduce_jobs(const double* base, double* out_base, const ReduceJob* jobs, int n_jobs, hipStream_t s) {
    if (n_jobs > 0) k_reduce_jobs<<<n_jobs, 1024, 0, s>>>(base, out_base, jobs);
    return hipGetLastError();
}
hipError_t psf_batch_stage1(double* base, const PsfSetDesc* sets, int n_sets, int max_n, hipStream_t s) {
    k_batch_stage1<<<dim3(nblk(max_n * max_n), n_sets), 256, 0, s>>>(base, sets);
    return hipGetLastError();
}
hipError_t psf_batch_stage2(double* base, const PsfSetDesc* sets, int n_sets, int max_n, hipStream_t s) {
    k_batch_stage2<<<dim3(nblk(max_n * max_n), n_sets), 256, 0, s>>>(base, sets);
    return hipGetLastError();
}
hipError_t psf_batch_rescan(double* base, const PsfSetDesc* sets, int n_sets, int max_n, int max_ratio, hipStream_t s) {
    k_batch_rescan_b0<<<dim3(nblk(max_n), n_sets), 256, 0, s>>>(base, sets);
    k_batch_descan<<<dim3(nblk(max_n * max_n), n_sets), 256, 0, s>>>(base, sets);
    k_batch_rescan_cumu<<<dim3(nblk(max_n * max_n * max_ratio), n_sets), 256, 0, s>>>(base, sets);
    k_batch_rescan_bin<<<dim3(nblk(max_n * max_n), n_sets), 256, 0, s>>>(base, sets);
    return hipGetLastError();
}

hipError_t psf_spline_prefilter(double* a, int ny, int nx, hipStream_t s) {
    k_spline_prefilter<<<nblk(nx), 256, 0, s>>>(a, ny, nx, 0);
    k_spline_prefilter<<<nblk(ny), 256, 0, s>>>(a, ny, nx, 1);
    return hipGetLastError();
}

}  // namespace rl
