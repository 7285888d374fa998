// quality_kernels.hip -- reconstruction-quality metrics of the figure-2 harness and of
// Deconvolver.record_iteration on the device (float64):
//   * |fftshift(fft2(x))| * scale, optionally log(1 + .)   (line_sted_tools.py:539-547,
//     line_sted_figure_2.py:353-355) for images of ANY size: a direct two-stage DFT on the
//     exact image grid -- the convolution path's transforms exist only for the padded
//     lengths 64..4608, and the metric is evaluated a handful of times per run;
//   * scipy.ndimage.map_coordinates(order=3, mode='constant') at a list of points
//     (line_sted_figure_2.py:381-384): cubic B-spline coefficients by the same mirror
//     prefilter as the PSF rotation (psf_kernels.hip), then a 4x4 neighbourhood per point.
#include <hip/hip_runtime.h>
#include "quality_kernels.hpp"

namespace rl {

// stage 1: S[img][y][k] = sum_x X[img][y][x] * W_nx[(k x) mod nx],  k in [0, nx/2]
__global__ void k_dft_rows_real(const double* __restrict__ x, const double2* __restrict__ w, double2* __restrict__ s1,
                                int ny, int nx, int hx) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, img = blockIdx.z;
    if (k >= hx) return;
    const double* row = x + ((size_t)img * ny + y) * nx;
    double re = 0.0, im = 0.0;
    int idx = 0;                       // (k * j) mod nx, kept incrementally
    for (int j = 0; j < nx; ++j) {
        const double2 t = w[idx];
        const double v = row[j];
        re += v * t.x;
        im += v * t.y;
        idx += k;
        if (idx >= nx) idx -= nx;
    }
    s1[((size_t)img * ny + y) * hx + k] = make_double2(re, im);
}

// stage 2 + magnitude + fftshift: F[ky][k] = sum_y S[y][k] * W_ny[(ky y) mod ny];
// out[(ky + ny/2) % ny][(k + nx/2) % nx] = f(|F| * scale), and the Hermitian partner
// F[-ky][-k] = conj(F[ky][k]) fills the other half of the columns.
__global__ void k_dft_cols_mag(const double2* __restrict__ s1, const double2* __restrict__ w, double* __restrict__ out,
                               int ny, int nx, int hx, double scale, int log1p_) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int ky = blockIdx.y, img = blockIdx.z;
    if (k >= hx) return;
    const double2* col = s1 + (size_t)img * ny * hx + k;
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int y = 0; y < ny; ++y) {
        const double2 t = w[idx], s = col[(size_t)y * hx];
        re += s.x * t.x - s.y * t.y;
        im += s.x * t.y + s.y * t.x;
        idx += ky;
        if (idx >= ny) idx -= ny;
    }
    double m = sqrt(re * re + im * im) * scale;
    if (log1p_) m = log(1.0 + m);
    double* o = out + (size_t)img * ny * nx;
    o[(size_t)((ky + ny / 2) % ny) * nx + (k + nx / 2) % nx] = m;
    const int mk = (nx - k) % nx;
    if (mk >= hx) {                    // columns not covered by the half spectrum
        const int mky = (ny - ky) % ny;
        o[(size_t)((mky + ny / 2) % ny) * nx + (mk + nx / 2) % nx] = m;
    }
}

__device__ __forceinline__ int q_mirror(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = (i < 0 ? -i : i) % p;
    return i >= n ? p - i : i;
}

// one thread per sample point; coef = prefiltered image
__global__ void k_spline_sample(const double* __restrict__ coef, int ny, int nx, const double* __restrict__ ys,
                                const double* __restrict__ xs, int n, double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double y = ys[e], x = xs[e], r = 0.0;
    if (y >= 0.0 && y <= (double)(ny - 1) && x >= 0.0 && x <= (double)(nx - 1)) {
        const double fy = floor(y), fx = floor(x), ty = y - fy, tx = x - fx;
        const double wy[4] = {(1 - ty) * (1 - ty) * (1 - ty) / 6, (3 * ty * ty * ty - 6 * ty * ty + 4) / 6,
                              (-3 * ty * ty * ty + 3 * ty * ty + 3 * ty + 1) / 6, ty * ty * ty / 6};
        const double wx[4] = {(1 - tx) * (1 - tx) * (1 - tx) / 6, (3 * tx * tx * tx - 6 * tx * tx + 4) / 6,
                              (-3 * tx * tx * tx + 3 * tx * tx + 3 * tx + 1) / 6, tx * tx * tx / 6};
        for (int i = 0; i < 4; ++i) {
            const int yy = q_mirror((int)fy - 1 + i, ny);
            for (int j = 0; j < 4; ++j) r += wy[i] * wx[j] * coef[(size_t)yy * nx + q_mirror((int)fx - 1 + j, nx)];
        }
    }
    out[e] = r;
}

hipError_t quality_fft2_magnitude(const double* x, const void* wx, const void* wy, void* s1, double* out, int nimg,
                                  int ny, int nx, double scale, int log1p_, hipStream_t s) {
    const int hx = nx / 2 + 1;
    const dim3 grid((hx + 127) / 128, ny, nimg);
    k_dft_rows_real<<<grid, 128, 0, s>>>(x, (const double2*)wx, (double2*)s1, ny, nx, hx);
    k_dft_cols_mag<<<grid, 128, 0, s>>>((const double2*)s1, (const double2*)wy, out, ny, nx, hx, scale, log1p_);
    return hipGetLastError();
}

hipError_t quality_spline_sample(const double* coef, int ny, int nx, const double* ys, const double* xs, int n,
                                 double* out, hipStream_t s) {
    k_spline_sample<<<(n + 255) / 256, 256, 0, s>>>(coef, ny, nx, ys, xs, n, out);
    return hipGetLastError();
}

}  // namespace rl
