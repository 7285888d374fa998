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

// Philox key / counter of an image of the launch: one seed and consecutive image indices from
// image0, or -- frame_seeds != nullptr -- a seed and an image id per frame (V views each), so that
// a frame's draws do not depend on which batch it was put in.
struct PoissonKeys {
    unsigned long long seed0;
    unsigned image0, V;
    const unsigned long long* frame_seeds;   // [frames] or nullptr
    const unsigned* frame_ids;               // [frames] (with frame_seeds)
    __device__ __forceinline__ unsigned long long seed(unsigned img) const { return frame_seeds ? frame_seeds[img / V] : seed0; }
    __device__ __forceinline__ unsigned image(unsigned img) const {
        return frame_seeds ? frame_ids[img / V] * V + img % V : image0 + img;
    }
};

// noisy = Poisson(noiseless) + 1e-9   (line_sted_tools.py:510).  ONE launch since round 4: a workgroup takes tiles of 2048
// pixels; every pixel gets the first PTRS attempt (philox_poisson_fast: ~87 % are squeeze-accepted and written at once), the
// others go -- index and rate -- to a list in LDS: lam >= 10 from the front, 0 < lam < 10 (multiplication method) from the back.
// The listed pixels are then finished by the same workgroup, densely packed: one PTRS attempt per round (attempt 0 repeats the
// first candidate and continues into the acceptance test the fast path skipped), the survivors repacked into the other list
// between rounds -- a wave never waits for its slowest lane's fifth attempt -- and the small rates through the multiplication
// loop.  Values are a function of (seed, image, pixel) only, so the order is irrelevant: bit for bit what rounds 1-3's two
// launches (all pixels; then a global work list of the rejected ones, which re-read their rates from global memory: 2.9x the
// bytes of read + write) produced, and what the numpy twin produces.
constexpr int kPoissonE = 8, kPoissonTile = 256 * kPoissonE;
template <typename T>
__global__ void __launch_bounds__(256) k_poisson(const T* __restrict__ noiseless, T* __restrict__ noisy, unsigned n_pix, unsigned total,
                                                 PoissonKeys keys, int rng_kind) {
    __shared__ unsigned lst_idx[2][kPoissonTile];
    __shared__ T lst_lam[2][kPoissonTile];
    __shared__ unsigned cnt[2], cnt_small;
    const unsigned tid = threadIdx.x;
    for (unsigned tile0 = blockIdx.x * (unsigned)kPoissonTile; tile0 < total; tile0 += gridDim.x * (unsigned)kPoissonTile) {
        if (tid == 0) cnt[0] = cnt_small = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kPoissonE; ++e) {
            const unsigned i = tile0 + tid + 256u * (unsigned)e;
            if (i < total) {
                const T lam_t = noiseless[i];
                const double lam = (double)lam_t;
                if (rng_kind != 1) {
                    noisy[i] = (T)(lam + 1e-9);
                } else {
                    const unsigned img = i / n_pix, pix = i - img * n_pix;   // 32-bit division
                    double k;
                    if (philox_poisson_fast(lam, keys.seed(img), keys.image(img), pix, &k)) {
                        noisy[i] = (T)(k + 1e-9);
                    } else if (lam >= 10.0) {
                        const unsigned q = atomicAdd(&cnt[0], 1u);
                        lst_idx[0][q] = i;
                        lst_lam[0][q] = lam_t;
                    } else {
                        const unsigned q = kPoissonTile - 1u - atomicAdd(&cnt_small, 1u);
                        lst_idx[0][q] = i;
                        lst_lam[0][q] = lam_t;
                    }
                }
            }
        }
        __syncthreads();
        if (rng_kind != 1) continue;          // (uniform)
        // lam >= 10: attempt `blk` of every pixel still listed
        unsigned n = cnt[0];
        const unsigned n_small = cnt_small;
        int cur = 0;
        for (unsigned blk = 0; n > 0; ++blk) {
            if (tid == 0) cnt[cur ^ 1] = 0;
            __syncthreads();
            const bool last = blk + 1 == kPoissonMaxBlocks;
            for (unsigned q = tid; q < n; q += 256u) {
                const unsigned i = lst_idx[cur][q];
                const T lam_t = lst_lam[cur][q];
                const unsigned img = i / n_pix, pix = i - img * n_pix;
                const unsigned long long seed = keys.seed(img);
                double k;
                const bool done = philox_ptrs_attempt((double)lam_t, (unsigned)seed, (unsigned)(seed >> 32), keys.image(img), pix, blk, &k);
                if (done || last) {
                    noisy[i] = (T)((done || k >= 0.0 ? k : 0.0) + 1e-9);
                } else {
                    const unsigned q2 = atomicAdd(&cnt[cur ^ 1], 1u);   // (at most n <= front region: never reaches the small rates at the back of list 0)
                    lst_idx[cur ^ 1][q2] = i;
                    lst_lam[cur ^ 1][q2] = lam_t;
                }
            }
            __syncthreads();
            n = cnt[cur ^ 1];
            cur ^= 1;
        }
        // 0 < lam < 10: multiplication method, one pixel per lane
        for (unsigned q = tid; q < n_small; q += 256u) {
            const unsigned at = kPoissonTile - 1u - q, i = lst_idx[0][at];
            const unsigned img = i / n_pix;
            noisy[i] = (T)(philox_poisson((double)lst_lam[0][at], keys.seed(img), keys.image(img), i - img * n_pix) + 1e-9);
        }
        __syncthreads();                      // the lists are free for the next tile
    }
}

// ---- host <-> plan staging: float64 host arrays are converted on the device -----------
// per-frame sums of a stack [frames][n], accumulated in float64 (wavefront shuffles).  grid (frames, chunks): workgroup (f, c) sums
// the c-th of `chunks` equal sections of frame f into out[f * chunks + c] -- chunks == 1: the frame's sum itself.  Few large frames
// (a 4096^2 float64 image is 134 MB: one workgroup read it at 19 GB/s, 7 ms) take kSumChunks sections and a second launch of the
// same kernel over the partial sums; the order of the additions is fixed either way, so a frame's sum does not depend on the batch.
template <typename T>
__global__ void __launch_bounds__(1024) k_frame_sums(const T* __restrict__ x, size_t n, double* __restrict__ out) {
    __shared__ double part[16];
    const size_t chunks = gridDim.y, len = (n + chunks - 1) / chunks, i0 = blockIdx.y * len, i1 = i0 + len < n ? i0 + len : n;
    const T* f = x + (size_t)blockIdx.x * n;
    double v = 0.0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) v += (double)f[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        v = threadIdx.x < (blockDim.x >> 6) ? part[threadIdx.x] : 0.0;
        for (int off = 8; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * chunks + blockIdx.y] = v;
    }
}
// sums[0 .. frames) <- the frames' sums; sums[frames .. frames * (1 + kSumChunks)) is scratch (aux_sums_elems)
template <typename T>
static void frame_sums(const T* x, size_t n, size_t frames, double* sums, hipStream_t s) {
    if (n >= ((size_t)1 << 20) && frames * kSumChunks <= 8192) {
        double* part = sums + frames;
        k_frame_sums<T><<<dim3((unsigned)frames, kSumChunks), 1024, 0, s>>>(x, n, part);
        k_frame_sums<double><<<dim3((unsigned)frames, 1), 64, 0, s>>>(part, (size_t)kSumChunks, sums);
    } else {
        k_frame_sums<T><<<dim3((unsigned)frames, 1), 1024, 0, s>>>(x, n, sums);
    }
}

// dst[f][i] = (T)(src[f][i] * (target[f] / sums[f]))   (target == nullptr: plain conversion)
template <typename T>
__global__ void k_scale_convert(const double* __restrict__ src, T* __restrict__ dst, size_t n, size_t frames,
                                const double* __restrict__ target, const double* __restrict__ sums) {
    const size_t total = n * frames;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / n;
        const double k = target ? target[f] / sums[f] : 1.0;
        dst[i] = (T)(src[i] * k);
    }
}

template <typename T>
__global__ void k_to_f64(const T* __restrict__ src, double* __restrict__ dst, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (double)src[i];
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

// Real parts of a complex array + max |im| and max |z| (stats[0], stats[1]; zeroed by the caller): a
// point-symmetric PSF has a real spectrum, which the column kernels can then use as a real multiplier.
// Non-negative doubles order like their bit patterns, so the maxima are integer atomics.
template <typename T>
__global__ void k_split_real(const cx<T>* __restrict__ z, size_t n, T* __restrict__ re, double* __restrict__ stats) {
    double im = 0.0, ab = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {   // grid stride
        const cx<T> v = z[i];
        re[i] = v.re;
        im = fmax(im, fabs((double)v.im));
        ab = fmax(ab, fmax(fabs((double)v.re), fabs((double)v.im)));
    }
    for (int off = 32; off > 0; off >>= 1) {
        im = fmax(im, __shfl_down(im, off, 64));
        ab = fmax(ab, __shfl_down(ab, off, 64));
    }
    if (threadIdx.x % 64 == 0) {
        atomicMax((unsigned long long*)stats, (unsigned long long)__double_as_longlong(im));
        atomicMax((unsigned long long*)stats + 1, (unsigned long long)__double_as_longlong(ab));
    }
}

hipError_t aux_split_real(int dtype, const void* z, size_t n, void* re, double* stats, hipStream_t s) {
    hipError_t e = hipMemsetAsync(stats, 0, 2 * sizeof(double), s);
    if (e != hipSuccess) return e;
    const unsigned g = blocks_for(n, 256);
    if (dtype == DT_F32) k_split_real<float><<<g, 256, 0, s>>>((const cx<float>*)z, n, (float*)re, stats);
    else k_split_real<double><<<g, 256, 0, s>>>((const cx<double>*)z, n, (double*)re, stats);
    return hipGetLastError();
}

// H_t(ones) without a transform (line_sted_tools.py:589-592): the 'same' convolution of an image of ones with a PSF is
// the sum of the PSF over the rectangle of taps that still meet the image,
//     conv(1, p)[i][j] = sum over a in [i + cy - ny + 1, i + cy], b in [j + cx - nx + 1, j + cx] (inside the PSF) of p[a][b],
// i.e. four reads of the PSF's float64 integral image I[a][b] = sum_{a' < a, b' < b} p[a'][b'] per view; each view's sum is
// clamped at 0 as the reference clamps each view's convolution (:587).  Exact to float64 rounding, where the transform path
// of an f32 plan carries ~2e-7 of white rounding noise -- an error every iteration multiplies into the estimate again.
template <typename T>
__global__ void k_box_norm(const double* __restrict__ integ, T* __restrict__ out, int V, int py, int px, int ny, int nx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= nx) return;
    const int cy = (py - 1) / 2, cx = (px - 1) / 2;
    const int a0 = max(i + cy - ny + 1, 0), a1 = min(i + cy, py - 1) + 1;
    const int b0 = max(j + cx - nx + 1, 0), b1 = min(j + cx, px - 1) + 1;
    double acc = 0.0;
    if (a1 > a0 && b1 > b0) {
        for (int v = 0; v < V; ++v) {
            const double* I = integ + (size_t)v * (py + 1) * (px + 1);
            const double s = (I[(size_t)a1 * (px + 1) + b1] - I[(size_t)a0 * (px + 1) + b1]) - (I[(size_t)a1 * (px + 1) + b0] - I[(size_t)a0 * (px + 1) + b0]);
            acc += s > 0.0 ? s : 0.0;
        }
    }
    out[(size_t)i * nx + j] = (T)acc;
}

hipError_t aux_box_norm(int dtype, const double* integral_dev, void* out, int V, int py, int px, int ny, int nx, hipStream_t s) {
    const dim3 grid((unsigned)((nx + 255) / 256), (unsigned)ny);
    if (dtype == DT_F32) k_box_norm<float><<<grid, 256, 0, s>>>(integral_dev, (float*)out, V, py, px, ny, nx);
    else k_box_norm<double><<<grid, 256, 0, s>>>(integral_dev, (double*)out, V, py, px, ny, nx);
    return hipGetLastError();
}

hipError_t aux_poisson(int dtype, const void* noiseless, void* noisy, unsigned n_pix, unsigned n_img, unsigned image0,
                       unsigned long long seed, int rng_kind, void* list_ws, hipStream_t s,
                       const unsigned long long* frame_seeds, const unsigned* frame_ids, unsigned V) {
    (void)list_ws;   // (rounds 1-3: the global work list between the two launches)
    const PoissonKeys keys{seed, image0, V ? V : 1u, frame_seeds, frame_ids};
    const size_t total = (size_t)n_pix * n_img;
    if (total >= 0xffffffffull - (size_t)kPoissonTile * 4096) return hipErrorInvalidValue;      // 32-bit pixel indices (and their tile stride)
    if (total == 0) return hipSuccess;
    const size_t tiles = (total + kPoissonTile - 1) / kPoissonTile;
    const unsigned g = (unsigned)(tiles > 4096 ? 4096 : tiles);
    if (dtype == DT_F32) k_poisson<float><<<g, 256, 0, s>>>((const float*)noiseless, (float*)noisy, n_pix, (unsigned)total, keys, rng_kind);
    else k_poisson<double><<<g, 256, 0, s>>>((const double*)noiseless, (double*)noisy, n_pix, (unsigned)total, keys, rng_kind);
    return hipGetLastError();
}

hipError_t aux_scale_convert(int dtype, const double* src, void* dst, size_t n, size_t frames, const double* target,
                             double* sums, hipStream_t s, bool want_sums) {
    if (target || want_sums) frame_sums<double>(src, n, frames, sums, s);
    const unsigned g = blocks_for(n * frames, 256);
    if (dtype == DT_F32) k_scale_convert<float><<<g, 256, 0, s>>>(src, (float*)dst, n, frames, target, sums);
    else k_scale_convert<double><<<g, 256, 0, s>>>(src, (double*)dst, n, frames, target, sums);
    return hipGetLastError();
}

template <typename T>
__global__ void k_any_negative(const T* __restrict__ x, size_t n, int* __restrict__ flag) {
    bool neg = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) neg = neg || x[i] < (T)0;
    if (__any(neg) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
hipError_t aux_any_negative(int dtype, const void* src, size_t n, int* flag, hipStream_t s) {
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e != hipSuccess || n == 0) return e;
    const unsigned g = blocks_for(n, 256);
    if (dtype == DT_F32) k_any_negative<float><<<g, 256, 0, s>>>((const float*)src, n, flag);
    else k_any_negative<double><<<g, 256, 0, s>>>((const double*)src, n, flag);
    return hipGetLastError();
}

hipError_t aux_image_sums(int dtype, const void* src, size_t n, size_t frames, double* sums, hipStream_t s) {
    if (frames == 0) return hipSuccess;
    if (dtype == DT_F32) frame_sums<float>((const float*)src, n, frames, sums, s);
    else frame_sums<double>((const double*)src, n, frames, sums, s);
    return hipGetLastError();
}

// dst[f][i] = (T)(src[idx[f]][i] * (target[f] / sums[idx[f]]))   (target == nullptr: plain conversion): frames that share an
// object (a sweep's seeds) are staged and uploaded once
template <typename T>
__global__ void k_scale_convert_indexed(const double* __restrict__ src, const unsigned* __restrict__ idx, T* __restrict__ dst, size_t n,
                                        size_t frames, const double* __restrict__ target, const double* __restrict__ sums) {
    const size_t total = n * frames;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / n, u = idx[f];
        const double k = target ? target[f] / sums[u] : 1.0;
        dst[i] = (T)(src[u * n + (i - f * n)] * k);
    }
}
hipError_t aux_scale_convert_indexed(int dtype, const double* src, const unsigned* idx, size_t n_unique, void* dst, size_t n, size_t frames,
                                     const double* target, double* sums, hipStream_t s) {
    if (target) frame_sums<double>(src, n, n_unique, sums, s);
    const unsigned g = blocks_for(n * frames, 256);
    if (dtype == DT_F32) k_scale_convert_indexed<float><<<g, 256, 0, s>>>(src, idx, (float*)dst, n, frames, target, sums);
    else k_scale_convert_indexed<double><<<g, 256, 0, s>>>(src, idx, (double*)dst, n, frames, target, sums);
    return hipGetLastError();
}

hipError_t aux_to_f64(int dtype, const void* src, double* dst, size_t total, hipStream_t s) {
    const unsigned g = blocks_for(total, 256);
    if (dtype == DT_F32) k_to_f64<float><<<g, 256, 0, s>>>((const float*)src, dst, total);
    else k_to_f64<double><<<g, 256, 0, s>>>((const double*)src, dst, total);
    return hipGetLastError();
}

template <typename S, typename D>
__global__ void k_cast(const S* __restrict__ src, D* __restrict__ dst, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) dst[i] = (D)src[i];
}
hipError_t aux_cast(int dtype_src, const void* src, int dtype_dst, void* dst, size_t total, hipStream_t s) {
    if (total == 0) return hipSuccess;
    if (dtype_src == dtype_dst) return hipMemcpyAsync(dst, src, total * (dtype_src == DT_F32 ? 4 : 8), hipMemcpyDeviceToDevice, s);
    const unsigned g = blocks_for(total, 256);
    if (dtype_src == DT_F32) k_cast<float, double><<<g, 256, 0, s>>>((const float*)src, (double*)dst, total);
    else k_cast<double, float><<<g, 256, 0, s>>>((const double*)src, (float*)dst, total);
    return hipGetLastError();
}

size_t aux_poisson_workspace_bytes(size_t total_pixels) {
    (void)total_pixels;
    return 256;      // (the sampler keeps its work lists in LDS since round 4; the callers' buffers stay for the interface)
}

}  // namespace rl
