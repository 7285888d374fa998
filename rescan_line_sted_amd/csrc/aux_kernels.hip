// aux_kernels.hip -- small gfx950 kernels around the FFT path: constant fill,
// PSF spectrum (direct DFT of the small PSF support, float64), Poisson noise.
#include <hip/hip_runtime.h>
#include "aux_kernels.hpp"
#include "philox_poisson.hpp"

namespace rl {

template <typename T>
__global__ void k_fill(T* p, size_t n, T value) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = value;
}

// Stage 1: S1[v][a][kx] = sum_b psf[v][a][b] * W_Lx[(kx * ((b - cx) mod Lx)) mod Lx]
__global__ void k_psf_dft_rows(const double* __restrict__ psf, const double2* __restrict__ wx, double2* __restrict__ s1,
                               int py, int px, int lx, int kx) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y, v = blockIdx.z;
    if (k >= kx) return;
    const int cx = (px - 1) / 2;
    const double* row = psf + ((size_t)v * py + a) * px;
    double re = 0.0, im = 0.0;
    for (int b = 0; b < px; ++b) {
        const int off = ((b - cx) % lx + lx) % lx;
        const double2 w = wx[(int)(((long long)k * off) % lx)];
        re += row[b] * w.x;
        im += row[b] * w.y;
    }
    s1[((size_t)v * py + a) * kx + k] = make_double2(re, im);
}

// Stage 2: psf_hat[v][ky][kx] = scale * sum_a S1[v][a][kx] * W_Ly[(ky * ((a - cy) mod Ly)) mod Ly]
template <typename T>
__global__ void k_psf_dft_cols(const double2* __restrict__ s1, const double2* __restrict__ wy, cx<T>* __restrict__ out,
                               int py, int ly, int kx, int pitch, double scale, int transposed) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int ky = blockIdx.y, v = blockIdx.z;
    if (k >= pitch) return;
    const int cy = (py - 1) / 2;
    double re = 0.0, im = 0.0;
    if (k < kx) {
        for (int a = 0; a < py; ++a) {
            const int off = ((a - cy) % ly + ly) % ly;
            const double2 w = wy[(int)(((long long)ky * off) % ly)];
            const double2 s = s1[((size_t)v * py + a) * kx + k];
            re += s.x * w.x - s.y * w.y;
            im += s.x * w.y + s.y * w.x;
        }
    }
    if (transposed) {   // [view][kx][ly]: contiguous along ky for wave-private column transforms
        if (k < kx) out[((size_t)v * kx + k) * ly + ky] = mk<T>((T)(re * scale), (T)(im * scale));
    } else {
        out[((size_t)v * ly + ky) * pitch + k] = mk<T>((T)(re * scale), (T)(im * scale));
    }
}

// noisy = Poisson(noiseless) + 1e-9   (line_sted_tools.py:510)
template <typename T>
__global__ void k_poisson(const T* __restrict__ noiseless, T* __restrict__ noisy, unsigned n_pix, unsigned n_img,
                          unsigned long long seed, int rng_kind) {
    const size_t total = (size_t)n_pix * n_img;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned img = (unsigned)(i / n_pix), pix = (unsigned)(i % n_pix);
        const double lam = (double)noiseless[i];
        const double k = rng_kind == 1 ? philox_poisson(lam, seed, img, pix) : lam;
        noisy[i] = (T)(k + 1e-9);
    }
}

static unsigned blocks_for(size_t n, unsigned block) {
    size_t b = (n + block - 1) / block;
    return (unsigned)(b > 2048 ? 2048 : (b ? b : 1));   // grid-stride beyond 2048 workgroups
}

hipError_t aux_fill(int dtype, void* p, size_t n, double value, hipStream_t s) {
    if (dtype == DT_F32) k_fill<float><<<blocks_for(n, 256), 256, 0, s>>>((float*)p, n, (float)value);
    else k_fill<double><<<blocks_for(n, 256), 256, 0, s>>>((double*)p, n, value);
    return hipGetLastError();
}

hipError_t aux_psf_spectrum(int dtype, const double* psf_dev, const void* wx_dev, const void* wy_dev, void* s1_dev,
                            void* out, int n_psf, int py, int px, int ly, int lx, int kx, int pitch, int transposed,
                            hipStream_t s) {
    k_psf_dft_rows<<<dim3((kx + 127) / 128, py, n_psf), 128, 0, s>>>(psf_dev, (const double2*)wx_dev, (double2*)s1_dev, py, px, lx, kx);
    const double scale = 1.0 / ((double)ly * (double)lx);
    if (dtype == DT_F32)
        k_psf_dft_cols<float><<<dim3((pitch + 127) / 128, ly, n_psf), 128, 0, s>>>((const double2*)s1_dev, (const double2*)wy_dev, (cx<float>*)out, py, ly, kx, pitch, scale, transposed);
    else
        k_psf_dft_cols<double><<<dim3((pitch + 127) / 128, ly, n_psf), 128, 0, s>>>((const double2*)s1_dev, (const double2*)wy_dev, (cx<double>*)out, py, ly, kx, pitch, scale, transposed);
    return hipGetLastError();
}

hipError_t aux_poisson(int dtype, const void* noiseless, void* noisy, unsigned n_pix, unsigned n_img,
                       unsigned long long seed, int rng_kind, hipStream_t s) {
    const size_t total = (size_t)n_pix * n_img;
    if (dtype == DT_F32)
        k_poisson<float><<<blocks_for(total, 256), 256, 0, s>>>((const float*)noiseless, (float*)noisy, n_pix, n_img, seed, rng_kind);
    else
        k_poisson<double><<<blocks_for(total, 256), 256, 0, s>>>((const double*)noiseless, (double*)noisy, n_pix, n_img, seed, rng_kind);
    return hipGetLastError();
}

}  // namespace rl
