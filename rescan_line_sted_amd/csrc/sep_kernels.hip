// sep_kernels.hip -- direct separable stencils for rank-1 PSFs (BASELINE north
// star: "direct separable stencils with LDS-staged tiles or FFT... chosen per kernel size").
//
// A rank-1 PSF p[a][b] = u[a] v[b] (the 0 / 90 degree line PSFs are: they are rot90s of the blurred line,
// line_sted_figure_2.py:266-269) turns the zero padded 'same' convolution of H / H_t (line_sted_tools.py:
// 567-594) into a row stencil with v followed by a column stencil with u: (py + px) multiply-adds per pixel
// instead of four FFT passes.  That wins for SMALL kernels only -- at the figure-2 size (107 taps a side) the
// FFT path is 3x faster -- so the plan picks this path when every view is rank 1 and py + px <= a threshold
// (rlsted.cpp).  Same semantics as the FFT path: out[i][j] = sum_ab x[i + cy - a][j + cx - b] p[a][b],
// cy = (py-1)/2, cx = (px-1)/2, zero outside the image, each view's result clamped at 0 (ref:575,587).
//
// Row pass: one workgroup = 256 outputs of one row, the row segment + halo staged in LDS.
// Column pass: one workgroup = 64 columns x 32 rows of outputs, the (32 + py - 1) x 64 tile of row-pass
// results staged in LDS; the Richardson-Lucy pointwise steps are its epilogues (ref:520-531).
#include <hip/hip_runtime.h>

#include <mutex>
#include <set>

#include <cstdlib>

#include "sep_kernels.hpp"

namespace rl {
namespace {

constexpr int kRowSeg = 256;   // outputs per workgroup in the row pass
constexpr int kColW = 64, kColH = 32;
constexpr size_t kSep2dMaxLds = 160 * 1024;

// out[img][y][x] = sum_b in[src(img)][y][x + cx - b] * v[view(img)][b]
template <typename T>
__global__ void __launch_bounds__(kRowSeg) k_sep_rows(const T* __restrict__ in, T* __restrict__ out, const T* __restrict__ taps_v,
                                                      int ny, int nx, int px, int V, int in_div) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* seg = reinterpret_cast<T*>(smem);                    // [kRowSeg + px - 1]
    const int img = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * kRowSeg, t = threadIdx.x;
    const int cx = (px - 1) / 2, view = img % V;
    const T* __restrict__ row = in + ((size_t)(img / in_div) * ny + y) * nx;
    // input index of seg[i]: x0 + i + cx - (px - 1)
    for (int i = t; i < kRowSeg + px - 1; i += kRowSeg) {
        const int xi = x0 + i + cx - (px - 1);
        seg[i] = (xi >= 0 && xi < nx) ? row[xi] : (T)0;
    }
    __syncthreads();
    const int x = x0 + t;
    if (x >= nx) return;
    const T* __restrict__ v = taps_v + (size_t)view * px;
    T acc = 0;
    for (int b = 0; b < px; ++b) acc += seg[t + (px - 1) - b] * v[b];   // x + cx - b  <->  seg[t + px - 1 - b]
    out[((size_t)img * ny + y) * nx + x] = acc;
}

enum SepMode { SEP_STORE = 0, SEP_RATIO = 1, SEP_SUM = 2, SEP_UPDATE = 3 };

// column stencil of the row-pass results + epilogue.  Images of `tmp` are [frame*V + view].
//   SEP_STORE : dst[frame*V+view] = max(conv, 0)                                   (H / noiseless)
//   SEP_RATIO : dst[frame*V+view] = aux[frame*V+view] / max(conv, 0)               (measurement / H(est); 1 where conv <= 0)
//   SEP_SUM   : dst[frame] = sum_v max(conv_v, 0) (/ norm if norm)                 (H_t, normaliser)
//   SEP_UPDATE: dst[frame] *= sum_v max(conv_v, 0) / norm                          (est *= H_t(ratio) / H_t(1))
template <typename T, int MODE>
__global__ void __launch_bounds__(256) k_sep_cols(const T* __restrict__ tmp, const T* __restrict__ taps_u, const T* __restrict__ aux,
                                                  const T* __restrict__ norm, T* __restrict__ dst, int ny, int nx, int py, int V) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);                   // [kColH + py - 1][kColW]
    const int x0 = blockIdx.x * kColW, y0 = blockIdx.y * kColH, t = threadIdx.x;
    const int c = t % kColW, g = t / kColW;                 // column in the tile, row group (4 groups of 8 rows)
    const int cy = (py - 1) / 2;
    const bool multi = MODE == SEP_SUM || MODE == SEP_UPDATE;
    const int frame = blockIdx.z;                           // multi: frame; else image frame*V + view
    const int nview = multi ? V : 1;
    T acc[kColH / 4];
#pragma unroll
    for (int k = 0; k < kColH / 4; ++k) acc[k] = 0;
    for (int vw = 0; vw < nview; ++vw) {
        const int img = multi ? frame * V + vw : frame;
        const int view = multi ? vw : frame % V;
        const T* __restrict__ src = tmp + (size_t)img * ny * nx;
        if (vw > 0) __syncthreads();
        // tile row i holds input row y0 + i + cy - (py - 1)
        for (int i = g; i < kColH + py - 1; i += 4) {
            const int yi = y0 + i + cy - (py - 1), x = x0 + c;
            tile[i * kColW + c] = (yi >= 0 && yi < ny && x < nx) ? src[(size_t)yi * nx + x] : (T)0;
        }
        __syncthreads();
        const T* __restrict__ u = taps_u + (size_t)view * py;
#pragma unroll
        for (int k = 0; k < kColH / 4; ++k) {
            const int r = g * (kColH / 4) + k;              // output row y0 + r
            T s = 0;
            for (int a = 0; a < py; ++a) s += tile[(r + (py - 1) - a) * kColW + c] * u[a];
            acc[k] += s > (T)0 ? s : (T)0;                  // each view clamped before the sum (ref:587)
        }
    }
    const int x = x0 + c;
    if (x >= nx) return;
#pragma unroll
    for (int k = 0; k < kColH / 4; ++k) {
        const int y = y0 + g * (kColH / 4) + k;
        if (y >= ny) continue;
        const size_t o = ((size_t)frame * ny + y) * nx + x;
        const size_t pix = (size_t)y * nx + x;
        if (MODE == SEP_STORE) dst[o] = acc[k];
        else if (MODE == SEP_RATIO) dst[o] = acc[k] > (T)0 ? aux[o] / acc[k] : (T)1;   // (a prediction that is not positive: neutral pixel, conv_kernels.hpp rl_ratio)
        else if (MODE == SEP_SUM) dst[o] = norm ? acc[k] / norm[pix] : acc[k];
        else dst[o] = dst[o] * (acc[k] / norm[pix]);
    }
}

// ---- both passes in one kernel: the input tile + halo staged in LDS once, row stencil LDS -> LDS, column
// stencil LDS -> registers, epilogue.  Taps arrive flipped and zero padded to a multiple of 8 (correlation
// form: out[y][x] = sum_k in[..+k] f[k]), so every thread slides a 16-register window along its 8 outputs and
// an LDS value is read once per 8 multiply-adds.
//   STORE / RATIO : in = [frames] (the tile is shared by the views), dst = [frames*V]
//   SUM / UPDATE  : in = [frames*V], dst = [frames]
template <typename T>
__device__ __forceinline__ void sep_window8(const T* __restrict__ base, int stride, const T* __restrict__ taps, int chunks, T (&acc)[8]) {
    T win[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { win[j] = base[j * stride]; acc[j] = 0; }
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) win[8 + j] = base[(8 * (c + 1) + j) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const T f = taps[c * 8 + k];                    // uniform address: an LDS broadcast
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += win[k + j] * f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) win[j] = win[8 + j];
    }
}

// DIRECT (round 4): the PSF is NOT rank 1 -- no row pass; every output sums px column windows of the input tile, taps
// uf = [V][px][8 nca]: F[l][k] = p[py-1-k][px-1-l] (flipped both ways, zero padded along k), vf unused.  py * px multiply-adds per
// pixel, all of one sign for a non-negative PSF: the RELATIVE accuracy the FFT path cannot give a dark region.
template <typename T, int MODE, int TH, bool DIRECT = false>
__global__ void __launch_bounds__(256) k_sep2d(const T* __restrict__ in, const T* __restrict__ uf, const T* __restrict__ vf,
                                               const T* __restrict__ aux, const T* __restrict__ norm, T* __restrict__ dst, int ny, int nx,
                                               int py, int px, int V) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool multi = MODE == SEP_SUM || MODE == SEP_UPDATE;
    constexpr int NI = TH / 8 * kColW / 256;               // column-pass items (8 rows of one column) per thread
    static_assert(NI >= 1, "tile height");
    const int nca = (py + 7) / 8, ncb = (px + 7) / 8;
    const int R = TH + 8 * nca, IP = (kColW + 8 * ncb) | 1, TP = kColW + 1;   // rows staged, odd pitches
    T* tin = reinterpret_cast<T*>(smem);                    // [R][IP]  input tile, element (i, j) <-> (y0 - oy + i, x0 - ox + j)
    T* tmp = tin + (size_t)R * IP;                          // [R][TP]  row-pass results (not DIRECT)
    T* ftaps = DIRECT ? tmp : tmp + (size_t)R * TP;         // [V][8 nca] then [V][8 ncb] (DIRECT: [V][px][8 nca]): LDS broadcasts instead of scalar-load latency
    if constexpr (DIRECT) {
        for (int i = threadIdx.x; i < V * px * 8 * nca; i += 256) ftaps[i] = uf[i];
    } else {
        for (int i = threadIdx.x; i < V * 8 * nca; i += 256) ftaps[i] = uf[i];
        for (int i = threadIdx.x; i < V * 8 * ncb; i += 256) ftaps[V * 8 * nca + i] = vf[i];
    }
    const int oy = py - 1 - (py - 1) / 2, ox = px - 1 - (px - 1) / 2;
    const int x0 = blockIdx.x * kColW, y0 = blockIdx.y * TH, frame = blockIdx.z, t = threadIdx.x;
    T sum[NI][8];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[n][j] = 0;
    for (int view = 0; view < V; ++view) {
        if (view > 0) __syncthreads();                      // the previous view's column pass has read tmp
        if (view == 0 || multi) {
            const T* __restrict__ src = in + (size_t)(multi ? frame * V + view : frame) * ny * nx;
            for (int i = t / kColW; i < R; i += 256 / kColW) {
                const int y = y0 - oy + i;
                const bool row_ok = y >= 0 && y < ny;
                for (int j = t % kColW; j < IP; j += kColW) {
                    const int x = x0 - ox + j;
                    tin[i * IP + j] = (row_ok && x >= 0 && x < nx) ? src[(size_t)y * nx + x] : (T)0;
                }
            }
            __syncthreads();
        }
        if constexpr (!DIRECT) {
            const T* fv = ftaps + V * 8 * nca + view * 8 * ncb;
            for (int w = t; w < R * (kColW / 8); w += 256) {    // lanes along rows: odd pitches keep LDS conflict free
                const int i = w % R, sgm = w / R;
                T acc[8];
                sep_window8(tin + i * IP + sgm * 8, 1, fv, ncb, acc);
#pragma unroll
                for (int j = 0; j < 8; ++j) tmp[i * TP + sgm * 8 + j] = acc[j];
            }
            __syncthreads();
        } else if (view == 0) {
            __syncthreads();                                    // the taps are in LDS
        }
        const T* fu = ftaps + view * (DIRECT ? px : 1) * 8 * nca;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int it = t + 256 * n, c = it % kColW, g = it / kColW;
            T acc[8];
            if constexpr (DIRECT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = 0;
                for (int l = 0; l < px; ++l) {                  // column x0 + c + l - ox of the tile, all its taps
                    T part[8];
                    sep_window8(tin + (g * 8) * IP + c + l, IP, fu + l * 8 * nca, nca, part);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += part[j];
                }
            } else {
                sep_window8(tmp + (g * 8) * TP + c, TP, fu, nca, acc);
            }
            const int x = x0 + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const T a = acc[j] > (T)0 ? acc[j] : (T)0;  // each view clamped (ref:575,587)
                if (multi) {
                    sum[n][j] += a;
                } else {
                    const int y = y0 + g * 8 + j;
                    if (x < nx && y < ny) {
                        const size_t o = (((size_t)frame * V + view) * ny + y) * nx + x;
                        dst[o] = MODE == SEP_STORE ? a : (a > (T)0 ? aux[o] / a : (T)1);   // (neutral where the prediction is not positive: rl_ratio)
                    }
                }
            }
        }
    }
    if (multi) {
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int it = t + 256 * n, c = it % kColW, g = it / kColW, x = x0 + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int y = y0 + g * 8 + j;
                if (x >= nx || y >= ny) continue;
                const size_t pix = (size_t)y * nx + x, o = (size_t)frame * ny * nx + pix;
                if (MODE == SEP_SUM) dst[o] = norm ? sum[n][j] / norm[pix] : sum[n][j];
                else dst[o] = dst[o] * (sum[n][j] / norm[pix]);
            }
        }
    }
}

template <typename T, int TH>
size_t sep2d_lds(int py, int px, int V, bool direct = false) {
    const int nca = (py + 7) / 8, ncb = (px + 7) / 8;
    if (direct) return ((size_t)(TH + 8 * nca) * ((kColW + 8 * ncb) | 1) + (size_t)V * px * 8 * nca) * sizeof(T);
    return ((size_t)(TH + 8 * nca) * (((kColW + 8 * ncb) | 1) + kColW + 1) + (size_t)V * 8 * (nca + ncb)) * sizeof(T);
}
template <typename T, int MODE, int TH, bool DIRECT = false>
hipError_t sep2d_launch(const void* in, const void* uf, const void* vf, const void* aux, const void* norm, void* dst, int frames,
                        int ny, int nx, int py, int px, int V, hipStream_t s) {
    static unsigned long long allowed_devices = 0;   // the attribute is per device: one bit per device id
    const size_t lds = sep2d_lds<T, TH>(py, px, V, DIRECT);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !(allowed_devices >> dev & 1ull)) {
        e = hipFuncSetAttribute((const void*)k_sep2d<T, MODE, TH, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSep2dMaxLds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) allowed_devices |= 1ull << dev;
    }
    const dim3 grid((unsigned)((nx + kColW - 1) / kColW), (unsigned)((ny + TH - 1) / TH), (unsigned)frames);
    k_sep2d<T, MODE, TH, DIRECT><<<grid, 256, lds, s>>>((const T*)in, (const T*)uf, (const T*)vf, (const T*)aux, (const T*)norm, (T*)dst, ny, nx, py, px, V);
    return hipGetLastError();
}
template <typename T, int TH, bool DIRECT = false>
hipError_t sep2d_t(int mode, const void* in, const void* uf, const void* vf, const void* aux, const void* norm, void* dst, int frames,
                   int ny, int nx, int py, int px, int V, hipStream_t s) {
    switch (mode) {
        case SEP_STORE: return sep2d_launch<T, SEP_STORE, TH, DIRECT>(in, uf, vf, aux, norm, dst, frames, ny, nx, py, px, V, s);
        case SEP_RATIO: return sep2d_launch<T, SEP_RATIO, TH, DIRECT>(in, uf, vf, aux, norm, dst, frames, ny, nx, py, px, V, s);
        case SEP_SUM: return sep2d_launch<T, SEP_SUM, TH, DIRECT>(in, uf, vf, aux, norm, dst, frames, ny, nx, py, px, V, s);
        case SEP_UPDATE: return sep2d_launch<T, SEP_UPDATE, TH, DIRECT>(in, uf, vf, aux, norm, dst, frames, ny, nx, py, px, V, s);
        default: return hipErrorInvalidValue;
    }
}

constexpr size_t kSepMaxLds = 160 * 1024;   // LDS of a gfx950 compute unit

template <typename T>
hipError_t rows_t(const void* in, void* out, const void* v, int images, int ny, int nx, int px, int V, int in_div, hipStream_t s) {
    const dim3 grid((unsigned)((nx + kRowSeg - 1) / kRowSeg), (unsigned)ny, (unsigned)images);
    k_sep_rows<T><<<grid, kRowSeg, (size_t)(kRowSeg + px - 1) * sizeof(T), s>>>((const T*)in, (T*)out, (const T*)v, ny, nx, px, V, in_div);
    return hipGetLastError();
}
template <typename T>
hipError_t cols_t(int mode, const void* tmp, const void* u, const void* aux, const void* norm, void* dst, int frames_or_images,
                  int ny, int nx, int py, int V, hipStream_t s) {
    const dim3 grid((unsigned)((nx + kColW - 1) / kColW), (unsigned)((ny + kColH - 1) / kColH), (unsigned)frames_or_images);
    const size_t lds = (size_t)(kColH + py - 1) * kColW * sizeof(T);
    if (lds > kSepMaxLds) return hipErrorInvalidValue;   // (the plan does not choose the stencils for such a PSF: sep_cols_fits)
    if (lds > 65536) {   // above the default dynamic-LDS limit: raise it, once per kernel AND DEVICE (the attribute belongs to the
                         // device's code object: a process-wide flag would leave a second GPU without it; a failure is not cached)
        static std::mutex mu;
        static std::set<int> raised;
        int dev = -1;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(mu);
        if (!raised.count(dev)) {
            e = hipFuncSetAttribute((const void*)k_sep_cols<T, SEP_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSepMaxLds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_sep_cols<T, SEP_RATIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSepMaxLds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_sep_cols<T, SEP_SUM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSepMaxLds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_sep_cols<T, SEP_UPDATE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSepMaxLds);
            if (e != hipSuccess) return e;
            raised.insert(dev);
        }
    }
    switch (mode) {
        case SEP_STORE: k_sep_cols<T, SEP_STORE><<<grid, 256, lds, s>>>((const T*)tmp, (const T*)u, (const T*)aux, (const T*)norm, (T*)dst, ny, nx, py, V); break;
        case SEP_RATIO: k_sep_cols<T, SEP_RATIO><<<grid, 256, lds, s>>>((const T*)tmp, (const T*)u, (const T*)aux, (const T*)norm, (T*)dst, ny, nx, py, V); break;
        case SEP_SUM: k_sep_cols<T, SEP_SUM><<<grid, 256, lds, s>>>((const T*)tmp, (const T*)u, (const T*)aux, (const T*)norm, (T*)dst, ny, nx, py, V); break;
        case SEP_UPDATE: k_sep_cols<T, SEP_UPDATE><<<grid, 256, lds, s>>>((const T*)tmp, (const T*)u, (const T*)aux, (const T*)norm, (T*)dst, ny, nx, py, V); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

hipError_t sep_rows(int dtype, const void* in, void* out, const void* taps_v, int images, int ny, int nx, int px, int V,
                    int in_div, hipStream_t s) {
    if (images < 1) return hipSuccess;
    if (images > 65535) return hipErrorInvalidValue;   // grid.z
    return dtype == DT_F32 ? rows_t<float>(in, out, taps_v, images, ny, nx, px, V, in_div, s)
                           : rows_t<double>(in, out, taps_v, images, ny, nx, px, V, in_div, s);
}
hipError_t sep_cols(int dtype, int mode, const void* tmp, const void* taps_u, const void* aux, const void* norm, void* dst,
                    int frames_or_images, int ny, int nx, int py, int V, hipStream_t s) {
    if (frames_or_images < 1) return hipSuccess;
    if (frames_or_images > 65535) return hipErrorInvalidValue;
    return dtype == DT_F32 ? cols_t<float>(mode, tmp, taps_u, aux, norm, dst, frames_or_images, ny, nx, py, V, s)
                           : cols_t<double>(mode, tmp, taps_u, aux, norm, dst, frames_or_images, ny, nx, py, V, s);
}


// tile height of the one-kernel form: RLSTED_SEP_TH (32 or 64) for float, 32 for double
static int sep_th32() {
    static const int th = getenv("RLSTED_SEP_TH") ? atoi(getenv("RLSTED_SEP_TH")) : 32;
    return th == 64 ? 64 : 32;
}
size_t sep2d_lds_bytes(int dtype, int py, int px, int V, bool direct = false) {
    if (dtype != DT_F32) return sep2d_lds<double, 32>(py, px, V, direct);
    return sep_th32() == 64 ? sep2d_lds<float, 64>(py, px, V, direct) : sep2d_lds<float, 32>(py, px, V, direct);
}
bool sep2d_fits(int dtype, int py, int px, int V) { return sep2d_lds_bytes(dtype, py, px, V) <= kSep2dMaxLds; }
bool direct2d_fits(int dtype, int py, int px, int V) { return sep2d_lds_bytes(dtype, py, px, V, true) <= kSep2dMaxLds; }
// the two-pass form: the column pass stages (32 + py - 1) rows of 64 columns, the row pass 256 + px - 1 values
bool sep_two_pass_fits(int dtype, int py, int px) {
    const size_t es = dtype == DT_F32 ? 4 : 8;
    return (size_t)(kColH + py - 1) * kColW * es <= kSepMaxLds && (size_t)(kRowSeg + px - 1) * es <= 65536;
}
hipError_t sep2d(int dtype, int mode, const void* in, const void* taps_uf, const void* taps_vf, const void* aux, const void* norm,
                 void* dst, int frames, int ny, int nx, int py, int px, int V, hipStream_t s) {
    if (frames < 1) return hipSuccess;
    if (taps_vf == nullptr) {   // the direct 2-D stencil: taps_uf = [V][px][8 * ceil(py / 8)]
        if (frames > 65535 || !direct2d_fits(dtype, py, px, V)) return hipErrorInvalidValue;
        if (dtype != DT_F32) return sep2d_t<double, 32, true>(mode, in, taps_uf, nullptr, aux, norm, dst, frames, ny, nx, py, px, V, s);
        return sep_th32() == 64 ? sep2d_t<float, 64, true>(mode, in, taps_uf, nullptr, aux, norm, dst, frames, ny, nx, py, px, V, s)
                                : sep2d_t<float, 32, true>(mode, in, taps_uf, nullptr, aux, norm, dst, frames, ny, nx, py, px, V, s);
    }
    if (frames > 65535 || !sep2d_fits(dtype, py, px, V)) return hipErrorInvalidValue;
    if (dtype != DT_F32) return sep2d_t<double, 32>(mode, in, taps_uf, taps_vf, aux, norm, dst, frames, ny, nx, py, px, V, s);
    return sep_th32() == 64 ? sep2d_t<float, 64>(mode, in, taps_uf, taps_vf, aux, norm, dst, frames, ny, nx, py, px, V, s)
                            : sep2d_t<float, 32>(mode, in, taps_uf, taps_vf, aux, norm, dst, frames, ny, nx, py, px, V, s);
}

}  // namespace rl
