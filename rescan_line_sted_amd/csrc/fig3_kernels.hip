// fig3_kernels.hip -- the scan-position loop of line_sted_figure_3.simulate_imaging (:76-273) on the
// device, float64, batched over scan positions ("thousands of independent
// scan positions per frame").  C ABI: rl_fig3_scan, rl_rotate_image (include/rlsted.h).
//
// Per scan position s = (sy, sx) the reference does (line numbers of line_sted_figure_3.py):
//   exc = shift(centered_exc, s); glow = rot_obj * exc; descanned = shift(glow, -s)        :174-176
//   descan_*:   inst = gaussian_filter(descanned, psf_sigma); sums of inst -> reconstruction  :182-204
//   multipoint: inst = gaussian_filter(glow, psf_sigma); region sums -> reconstruction        :205-222
//   rescan:     inst = shift(scale_y(gaussian_filter(descanned), 1/(R^2+1)), s); cum += inst  :223-239
// The shifts are integer (the scan positions are multiples of the integer step), and an interpolating
// spline reproduces its samples: shift == move + zero fill, so descanned[y][x] = rot_obj[y+sy][x+sx] *
// centered_exc[y][x].  scale_y is scipy.ndimage.zoom(order 3) along y: mirror prefilter + cubic
// B-spline evaluation at o (ny-1)/(out-1), zero padded back to ny rows (:394-409).  Everything else is
// the reference's operation order (Gaussian: centre tap first, tap pairs from the outside in).
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "ctx.hpp"

namespace {

__device__ __forceinline__ int reflect_index(int i, int n) {   // scipy 'reflect': d c b a | a b c d | d c b a
    const int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return m >= n ? period - 1 - m : m;
}
__device__ __forceinline__ int mirror_index(int i, int n) {    // scipy 'mirror': d c b | a b c d | c b a
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = (i < 0 ? -i : i) % p;
    return i >= n ? p - i : i;
}

// descanned glow (mode 0) or glow (mode 1: multipoint, nothing is descanned) of a chunk of positions
__global__ void k_glow(const double* __restrict__ rot_obj, const double* __restrict__ exc, const int* __restrict__ pos,
                       int n_pos, int ny, int nx, int mode, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)ny * nx;
    if (e >= n * n_pos) return;
    const int p = (int)(e / n), y = (int)((e % n) / nx), x = (int)(e % nx);
    const int sy = pos[2 * p], sx = pos[2 * p + 1];
    double v = 0.0;
    if (mode == 0) {
        const int qy = y + sy, qx = x + sx;
        if (qy >= 0 && qy < ny && qx >= 0 && qx < nx) v = rot_obj[(size_t)qy * nx + qx] * exc[(size_t)y * nx + x];
    } else {
        const int qy = y - sy, qx = x - sx;
        if (qy >= 0 && qy < ny && qx >= 0 && qx < nx) v = rot_obj[(size_t)y * nx + x] * exc[(size_t)qy * nx + qx];
    }
    out[e] = v;
}

// scipy.ndimage correlate1d (symmetric kernel) along y (axis 1) or x (axis 2) of [n_img][ny][nx]
__global__ void k_blur(const double* __restrict__ in, double* __restrict__ out, int n_img, int ny, int nx, int along_x,
                       const double* __restrict__ w, int radius) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n_img * ny * nx) return;
    const int x = (int)(e % nx), y = (int)((e / nx) % ny);
    const int len = along_x ? nx : ny, at = along_x ? x : y, st = along_x ? 1 : nx;
    const double* line = in + (e - (size_t)at * st);
    double acc = line[(size_t)at * st] * w[radius];
    for (int j = -radius; j < 0; ++j)
        acc += (line[(size_t)reflect_index(at + j, len) * st] + line[(size_t)reflect_index(at - j, len) * st]) * w[radius + j];
    out[e] = acc;
}

// per image: max and sum (one workgroup per image)
__global__ void __launch_bounds__(256) k_image_max_sum(const double* __restrict__ in, size_t n, double* __restrict__ vmax,
                                                       double* __restrict__ vsum, int stride_out) {
    __shared__ double smax[4], ssum[4];
    const double* img = in + (size_t)blockIdx.x * n;
    double m = -1.0e308, s = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = img[i];
        m = v > m ? v : m;
        s += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_down(m, off, 64), os = __shfl_down(s, off, 64);
        m = om > m ? om : m;
        s += os;
    }
    if (threadIdx.x % 64 == 0) {
        smax[threadIdx.x / 64] = m;
        ssum[threadIdx.x / 64] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            m = smax[k] > m ? smax[k] : m;
            s += ssum[k];
        }
        if (vmax) vmax[(size_t)blockIdx.x * stride_out] = m;
        if (vsum) vsum[(size_t)blockIdx.x * stride_out] = s;
    }
}

// descan_line: out[p][x] = sum over y of inst[p][y][x]   (inst.sum(axis=1), :190-192)
__global__ void k_column_sums(const double* __restrict__ in, int n_img, int ny, int nx, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n_img * nx) return;
    const int p = (int)(e / nx), x = (int)(e % nx);
    const double* img = in + (size_t)p * ny * nx + x;
    double s = 0.0;
    for (int y = 0; y < ny; ++y) s += img[(size_t)y * nx];
    out[e] = s;
}

// multipoint: region sums around every excitation spot (:209-220); regions [ry][rx] per position
__global__ void k_region_sums(const double* __restrict__ in, const int* __restrict__ pos, int n_img, int ny, int nx,
                              int pad, int exc_sep, int ry, int rx, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n_img * ry * rx) return;
    const int p = (int)(e / (ry * rx)), iy = (int)((e / rx) % ry), ix = (int)(e % rx);
    const int y_sp = pad + pos[2 * p] + iy * exc_sep, x_sp = pad + pos[2 * p + 1] + ix * exc_sep;
    const int h = exc_sep / 3;
    const int y0 = max(y_sp - h, 0), y1 = min(y_sp + h, ny), x0 = max(x_sp - h, 0), x1 = min(x_sp + h, nx);
    const double* img = in + (size_t)p * ny * nx;
    double s = 0.0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += img[(size_t)y * nx + x];
    out[e] = s;
}

// cubic B-spline prefilter along y (mirror conditions, exact-sum initialisation) of [n_img][ny][nx], in place;
// one thread per column
__global__ void k_prefilter_y(double* __restrict__ a, int n_img, int ny, int nx) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n_img * nx || ny < 2) return;
    double* c = a + (e / nx) * (size_t)ny * nx + (e % nx);
    const size_t st = nx;
    const double z = -0.26794919243112270647255365849413;   // sqrt(3) - 2
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    for (int i = 0; i < ny; ++i) c[i * st] *= gain;
    const double zn1 = pow(z, (double)(ny - 1));
    double c0 = c[0] + zn1 * c[(ny - 1) * st], zi = z;
    for (int i = 1; i < ny - 1; ++i) {
        c0 += zi * (c[i * st] + zn1 * c[(ny - 1 - i) * st]);
        zi *= z;
    }
    c[0] = c0 / (1.0 - zn1 * zn1);
    for (int i = 1; i < ny; ++i) c[i * st] += z * c[(i - 1) * st];
    c[(ny - 1) * st] = (z * c[(ny - 2) * st] + c[(ny - 1) * st]) * z / (z * z - 1.0);
    for (int i = ny - 2; i >= 0; --i) c[i * st] = z * (c[(i + 1) * st] - c[i * st]);
}

__device__ __forceinline__ void bspline3(double t, double* w) {
    w[0] = (1 - t) * (1 - t) * (1 - t) / 6;
    w[1] = (3 * t * t * t - 6 * t * t + 4) / 6;
    w[2] = (-3 * t * t * t + 3 * t * t + 3 * t + 1) / 6;
    w[3] = t * t * t / 6;
}

// rescan: inst[p] = clip(shift(pad(zoom_y(blurred[p])), s), 0, .)  (:226-232): output row Y holds zoomed row
// o = Y - sy - top when 0 <= o < out_ny, column X - sx of the source; negatives (spline ringing) clip to 0.
__global__ void k_rescan_inst(const double* __restrict__ coef, const int* __restrict__ pos, int n_img, int ny, int nx,
                              int out_ny, int top, double zoom, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)ny * nx;
    if (e >= n * n_img) return;
    const int p = (int)(e / n), Y = (int)((e % n) / nx), X = (int)(e % nx);
    const int o = Y - pos[2 * p] - top, sx = X - pos[2 * p + 1];
    double v = 0.0;
    // rows of the padded, scaled image come from Y - sy in [0, ny): the shift moves zeros in beyond that
    if (o >= 0 && o < out_ny && Y - pos[2 * p] >= 0 && Y - pos[2 * p] < ny && sx >= 0 && sx < nx) {
        const double yy = (double)o * zoom;
        if (yy >= 0.0 && yy <= (double)(ny - 1)) {
            const double fy = floor(yy);
            double w[4];
            bspline3(yy - fy, w);
            const double* c = coef + (size_t)p * n + sx;
            for (int i = 0; i < 4; ++i) v += w[i] * c[(size_t)mirror_index((int)fy - 1 + i, ny) * nx];
        }
    }
    out[e] = v < 0.0 ? 0.0 : v;
}

// running sum over the positions of a chunk: io[p] <- cum + io[0] + ... + io[p]; cum <- its last value
__global__ void k_prefix_accumulate(double* __restrict__ io, double* __restrict__ cum, int n_img, size_t n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double c = cum[e];
    for (int p = 0; p < n_img; ++p) {
        c += io[(size_t)p * n + e];
        io[(size_t)p * n + e] = c;
    }
    cum[e] = c;
}

// ---- scipy.ndimage.rotate(order 3, reshape=False) of one plane ----------------------------------
// mode 1 ('nearest'): 12 edge samples of padding, prefilter with half-sample symmetric ('reflect')
// conditions on the padded array, spline evaluated at the unclamped source coordinate with the four
// neighbour indices clamped to the padded extent.
__global__ void k_pad_edge(const double* __restrict__ in, int ny, int nx, int npad, double* __restrict__ out) {
    const int py = ny + 2 * npad, px = nx + 2 * npad;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= py * px) return;
    const int y = min(max(e / px - npad, 0), ny - 1), x = min(max(e % px - npad, 0), nx - 1);
    out[e] = in[(size_t)y * nx + x];
}
__global__ void k_prefilter_reflect(double* __restrict__ a, int ny, int nx, int axis) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = axis == 0 ? ny : nx, lines = axis == 0 ? nx : ny;
    if (line >= lines || n < 2) return;
    double* c = a + (axis == 0 ? line : (size_t)line * nx);
    const size_t st = axis == 0 ? nx : 1;
    const double z = -0.26794919243112270647255365849413;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    for (int i = 0; i < n; ++i) c[i * st] *= gain;
    const double zn = pow(z, (double)n);
    const double first = c[0];
    double acc = c[0] + zn * c[(n - 1) * st], zi = z;
    for (int i = 1; i < n; ++i) {
        acc += zi * (c[i * st] + zn * c[(n - 1 - i) * st]);
        zi *= z;
    }
    c[0] = acc * (z / (1.0 - zn * zn)) + first;
    for (int i = 1; i < n; ++i) c[i * st] += z * c[(i - 1) * st];
    c[(n - 1) * st] = c[(n - 1) * st] * (z / (z - 1.0));
    for (int i = n - 2; i >= 0; --i) c[i * st] = z * (c[(i + 1) * st] - c[i * st]);
}
__global__ void k_rotate_nearest(const double* __restrict__ coef, int ny, int nx, int npad, double c, double s, double hi,
                                 int clip, double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const int ox = e % nx, oy = e / nx;
    const int py = ny + 2 * npad, px = nx + 2 * npad;
    const double cy = (ny - 1) * 0.5, cx = (nx - 1) * 0.5;
    const double y = c * oy + s * ox + (cy - (c * cy + s * cx)) + npad;
    const double x = -s * oy + c * ox + (cx - (-s * cy + c * cx)) + npad;
    const double fy = floor(y), fx = floor(x);
    double wy[4], wx[4];
    bspline3(y - fy, wy);
    bspline3(x - fx, wx);
    double r = 0.0;
    for (int i = 0; i < 4; ++i) {
        const int yy = min(max((int)fy - 1 + i, 0), py - 1);
        for (int j = 0; j < 4; ++j) r += wy[i] * wx[j] * coef[(size_t)yy * px + min(max((int)fx - 1 + j, 0), px - 1)];
    }
    out[e] = clip ? (r < 0.0 ? 0.0 : (r > hi ? hi : r)) : r;
}

inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

std::vector<double> gaussian_weights(double sigma, double truncate, int* radius) {
    const int r = (int)(truncate * sigma + 0.5);
    std::vector<double> w(2 * r + 1);
    const double k = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = -r; i <= r; ++i) sum += (w[i + r] = std::exp(k * (double)(i * i)));
    for (double& v : w) v /= sum;
    *radius = r;
    return w;
}

}  // namespace

extern "C" {

int rl_rotate_image(rl_ctx* ctx, const double* in, double* out, int ny, int nx, double degrees, int clip) {
    if (!ctx || !in || !out) return fail(RL_ERR_INVALID, "NULL argument");
    if (ny < 1 || nx < 1) return fail(RL_ERR_INVALID, "non-positive shape");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int npad = 12, py = ny + 2 * npad, px = nx + 2 * npad;
    const size_t n = (size_t)ny * nx, np_ = (size_t)py * px;
    double* work = nullptr;
    RL_TRY(ctx->psf_workspace(2 * n + np_ + 8, &work));
    double *d_in = work, *d_out = work + n, *d_pad = work + 2 * n;
    HIP_TRY(hipMemcpyAsync(d_in, in, n * 8, hipMemcpyHostToDevice, s));
    double hi = 0.0;
    if (clip) {   // np.clip(rotated, 0, 1.1 * x.max())
        for (size_t i = 0; i < n; ++i) hi = in[i] > hi || i == 0 ? in[i] : hi;
        hi *= 1.1;
    }
    k_pad_edge<<<blocks(np_), 256, 0, s>>>(d_in, ny, nx, npad, d_pad);
    k_prefilter_reflect<<<blocks(px), 256, 0, s>>>(d_pad, py, px, 0);
    k_prefilter_reflect<<<blocks(py), 256, 0, s>>>(d_pad, py, px, 1);
    const double th = degrees * 0.017453292519943295769236907684886;
    k_rotate_nearest<<<blocks(n), 256, 0, s>>>(d_pad, ny, nx, npad, std::cos(th), std::sin(th), hi, clip, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RL_OK;
}

int rl_fig3_scan(rl_ctx* ctx, const rl_fig3_params* p, const double* rot_obj, const double* centered_exc,
                 const int* positions, int n_pos, const int* display, int n_display, double* pos_scalars,
                 double* pos_values, double* display_out, double* cum_final) {
    if (!ctx || !p || !rot_obj || !centered_exc || !positions || !pos_scalars) return fail(RL_ERR_INVALID, "NULL argument");
    if (p->imaging_type < 0 || p->imaging_type > 3) return fail(RL_ERR_INVALID, "imaging_type must be 0..3");
    if (p->ny < 2 || p->nx < 1 || n_pos < 1 || n_display < 0 || !(p->psf_sigma > 1e-15)) return fail(RL_ERR_INVALID, "bad shape / sigma");
    if (n_display > 0 && (!display || !display_out)) return fail(RL_ERR_INVALID, "display list without output");
    const int type = p->imaging_type, ny = p->ny, nx = p->nx;
    const bool rescan = type == 3, multipoint = type == 1, line = type == 2;
    if (rescan && !cum_final) return fail(RL_ERR_INVALID, "rescan_line needs cum_final");
    if ((line || multipoint) && !pos_values) return fail(RL_ERR_INVALID, "pos_values is NULL");
    if (multipoint && (p->exc_sep < 1 || p->pad < 1)) return fail(RL_ERR_INVALID, "multipoint needs exc_sep and pad");
    const int ry = multipoint ? (p->n_y + p->exc_sep - 1) / p->exc_sep : 0, rx = multipoint ? (p->n_x + p->exc_sep - 1) / p->exc_sep : 0;
    const int n_val = line ? nx : (multipoint ? ry * rx : 0);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)ny * nx;
    int radius = 0;
    const std::vector<double> hw = gaussian_weights(p->psf_sigma, 4.0, &radius);
    // chunk of positions: three image stacks of at most ~1.5 GB together
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_pos, ((size_t)64 << 20) / n));
    double* work = nullptr;
    const size_t need = 3 * n + (size_t)chunk * (3 * n + 4 + n_val) + hw.size() + (size_t)chunk + 64;
    RL_TRY(ctx->psf_workspace(need, &work));
    double *d_obj = work, *d_exc = d_obj + n, *d_cum = d_exc + n;
    double *a = d_cum + n, *b = a + (size_t)chunk * n, *c = b + (size_t)chunk * n;
    double *d_sc = c + (size_t)chunk * n, *d_val = d_sc + (size_t)chunk * 4, *d_w = d_val + (size_t)chunk * n_val;
    int* d_pos = reinterpret_cast<int*>(d_w + hw.size());   // 2 * chunk ints fit in chunk doubles
    HIP_TRY(hipMemcpyAsync(d_obj, rot_obj, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_exc, centered_exc, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_w, hw.data(), hw.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(d_cum, 0, n * 8, s));
    // scale_y geometry (:394-409): scipy.ndimage.zoom rounds the output height
    const int out_ny = rescan ? (int)std::lround(std::nearbyint((double)ny * p->rescan_scale)) : 0;
    const int top = rescan ? (ny - out_ny) / 2 : 0;
    const double zoom = out_ny > 1 ? (double)(ny - 1) / (double)(out_ny - 1) : 1.0;
    if (rescan && (out_ny < 1 || out_ny > ny)) return fail(RL_ERR_INVALID, "rescan_scale must shrink the image");
    std::vector<double> sc;
    int next_display = 0;
    for (int p0 = 0; p0 < n_pos; p0 += chunk) {
        const int np_ = std::min(chunk, n_pos - p0);
        const size_t total = (size_t)np_ * n;
        HIP_TRY(hipMemcpyAsync(d_pos, positions + 2 * (size_t)p0, (size_t)np_ * 2 * sizeof(int), hipMemcpyHostToDevice, s));
        k_glow<<<blocks(total), 256, 0, s>>>(d_obj, d_exc, d_pos, np_, ny, nx, multipoint ? 1 : 0, a);
        k_image_max_sum<<<np_, 256, 0, s>>>(a, n, d_sc + 0, nullptr, 4);               // glow.max()  (:252)
        k_blur<<<blocks(total), 256, 0, s>>>(a, b, np_, ny, nx, 0, d_w, radius);        // gaussian_filter axis 1, then 2
        k_blur<<<blocks(total), 256, 0, s>>>(b, a, np_, ny, nx, 1, d_w, radius);        // (axis 0 has one sample: identity)
        double* inst = a;
        if (rescan) {
            k_prefilter_y<<<blocks((size_t)np_ * nx), 256, 0, s>>>(a, np_, ny, nx);
            k_rescan_inst<<<blocks(total), 256, 0, s>>>(a, d_pos, np_, ny, nx, out_ny, top, zoom, b);
            inst = b;
        }
        k_image_max_sum<<<np_, 256, 0, s>>>(inst, n, d_sc + 1, d_sc + 3, 4);            // inst.max(), inst.sum()
        if (line) k_column_sums<<<blocks((size_t)np_ * nx), 256, 0, s>>>(inst, np_, ny, nx, d_val);
        if (multipoint) k_region_sums<<<blocks((size_t)np_ * n_val), 256, 0, s>>>(inst, d_pos, np_, ny, nx, p->pad, p->exc_sep, ry, rx, d_val);
        double* cum = inst;   // cum_detector_sig is inst_detector_sig except for rescan (:185,207,234)
        if (rescan) {
            HIP_TRY(hipMemcpyAsync(c, inst, total * 8, hipMemcpyDeviceToDevice, s));
            k_prefix_accumulate<<<blocks(n), 256, 0, s>>>(c, d_cum, np_, n);
            cum = c;
        }
        k_image_max_sum<<<np_, 256, 0, s>>>(cum, n, d_sc + 2, nullptr, 4);              // cum_detector_sig.max()
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(pos_scalars + 4 * (size_t)p0, d_sc, (size_t)np_ * 4 * 8, hipMemcpyDeviceToHost, s));
        if (n_val) HIP_TRY(hipMemcpyAsync(pos_values + (size_t)p0 * n_val, d_val, (size_t)np_ * n_val * 8, hipMemcpyDeviceToHost, s));
        for (; next_display < n_display && display[next_display] < p0 + np_; ++next_display) {
            const int d = display[next_display];
            if (d < p0 || (next_display > 0 && d <= display[next_display - 1])) return fail(RL_ERR_INVALID, "display indices must ascend");
            HIP_TRY(hipMemcpyAsync(display_out + (size_t)next_display * 2 * n, inst + (size_t)(d - p0) * n, n * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(display_out + (size_t)next_display * 2 * n + n, cum + (size_t)(d - p0) * n, n * 8, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));   // the chunk buffers are reused
    }
    if (next_display != n_display) return fail(RL_ERR_INVALID, "display index beyond the last position");
    if (rescan) {
        HIP_TRY(hipMemcpyAsync(cum_final, d_cum, n * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return RL_OK;
}

}  // extern "C"
