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

// noisy = Poisson(noiseless) + 1e-9   (line_sted_tools.py:510), in two launches so that
// the rare slow path (log / log-gamma acceptance test, further attempts) does not run
// with 13 % of the lanes active on every wave:
//   k_poisson_fast  every pixel: first PTRS attempt; squeeze-accepted pixels are written,
//                   the others are appended (index only) to a work list
//   k_poisson_slow  the listed pixels, densely packed, through the full sampler
// Values are a function of (seed, image, pixel) only, so the list order is irrelevant.
// Each workgroup owns a fixed segment of the work list (capacity = the pixels it
// visits), fills it through an LDS counter and publishes the fill level in counts[].
template <typename T>
__global__ void k_poisson_fast(const T* __restrict__ noiseless, T* __restrict__ noisy, unsigned n_pix, unsigned n_img,
                               unsigned long long seed, int rng_kind, unsigned* __restrict__ list, unsigned seg_cap,
                               unsigned* __restrict__ counts) {
    __shared__ unsigned fill;
    if (threadIdx.x == 0) fill = 0;
    __syncthreads();
    unsigned* seg = list + (size_t)blockIdx.x * seg_cap;
    const size_t total = (size_t)n_pix * n_img;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const double lam = (double)noiseless[i];
        if (rng_kind != 1) {
            noisy[i] = (T)(lam + 1e-9);
            continue;
        }
        double k;
        if (philox_poisson_fast(lam, seed, (unsigned)(i / n_pix), (unsigned)(i % n_pix), &k)) noisy[i] = (T)(k + 1e-9);
        else seg[atomicAdd(&fill, 1u)] = (unsigned)i;
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = fill;
}

template <typename T>
__global__ void k_poisson_slow(const T* __restrict__ noiseless, T* __restrict__ noisy, unsigned n_pix,
                               unsigned long long seed, const unsigned* __restrict__ list, unsigned seg_cap,
                               const unsigned* __restrict__ counts) {
    const unsigned* seg = list + (size_t)blockIdx.x * seg_cap;
    const unsigned n = counts[blockIdx.x];
    for (unsigned q = threadIdx.x; q < n; q += blockDim.x) {
        const unsigned i = seg[q];
        noisy[i] = (T)(philox_poisson((double)noiseless[i], seed, i / n_pix, i % n_pix) + 1e-9);
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
                       unsigned long long seed, int rng_kind, void* list_ws, hipStream_t s) {
    const size_t total = (size_t)n_pix * n_img;
    if (total >= 0xffffffffull) return hipErrorInvalidValue;      // 32-bit work-list entries
    const unsigned g = blocks_for(total, 256);
    // list_ws layout: g segments of seg_cap entries (seg_cap * g <= total + g*256), then g counters
    const unsigned seg_cap = (unsigned)((total + (size_t)g * 256 - 1) / ((size_t)g * 256)) * 256;
    unsigned* list = (unsigned*)list_ws;
    unsigned* counts = list + (size_t)seg_cap * g;
    if (dtype == DT_F32) {
        k_poisson_fast<float><<<g, 256, 0, s>>>((const float*)noiseless, (float*)noisy, n_pix, n_img, seed, rng_kind, list, seg_cap, counts);
        if (rng_kind == 1) k_poisson_slow<float><<<g, 256, 0, s>>>((const float*)noiseless, (float*)noisy, n_pix, seed, list, seg_cap, counts);
    } else {
        k_poisson_fast<double><<<g, 256, 0, s>>>((const double*)noiseless, (double*)noisy, n_pix, n_img, seed, rng_kind, list, seg_cap, counts);
        if (rng_kind == 1) k_poisson_slow<double><<<g, 256, 0, s>>>((const double*)noiseless, (double*)noisy, n_pix, seed, list, seg_cap, counts);
    }
    return hipGetLastError();
}

size_t aux_poisson_workspace_bytes(size_t total_pixels) {
    const unsigned g = blocks_for(total_pixels, 256);
    const size_t seg_cap = (total_pixels + (size_t)g * 256 - 1) / ((size_t)g * 256) * 256;
    return (seg_cap * g + g) * sizeof(unsigned);
}

}  // namespace rl
