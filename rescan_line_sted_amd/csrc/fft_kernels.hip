// fft_kernels.hip -- gfx950 kernels for one FFT length (compile with
// -DRL_CFG_L=<L>).  Column pass (FFT_y * psf_hat -> IFFT_y) and the fused row
// passes (IFFT_x -> Richardson-Lucy pointwise step -> FFT_x) of the
// convolution path; bodies in conv_kernels.hpp.
#include <hip/hip_ext.h>
#include "kernel_table.hpp"
#include "conv_kernels.hpp"
#include "fft_configs.hpp"
#include "dev_sync.hpp"
#include "fused_rl.hpp"

// waves/SIMD requested for the f32 ROW_RATIO kernel (needs <= 96 VGPRs, which it has
// within 2 registers; the other modes spill under that bound and are left alone)
#ifndef RL_ROW_MIN_WAVES
#define RL_ROW_MIN_WAVES 5
#endif
#ifndef RL_UPD_MIN_WAVES
#define RL_UPD_MIN_WAVES 1
#endif
// Off: measured neutral on time, and it costs traffic -- with whole images per XCD every XCD's L2 streams
// the full normaliser (1 MB) instead of the eighth its row groups touch (ROW_UPDATE +0.5 MB/frame).
#ifndef RL_ROW_XCD_REMAP
#define RL_ROW_XCD_REMAP 0
#endif
#ifndef RL_ROW_LEAN
#define RL_ROW_LEAN 1
#endif
// waves/SIMD requested for the f32 per-image column kernel: 6 = three 8-wave workgroups per CU
// (80 VGPRs, no spills; 3 x 51 KB LDS), whose load / transform / store phases overlap better than
// two (+1.7 % end to end).  The multi-view modes spill under that bound and keep 1.
#ifndef RL_COL_MIN_WAVES
#define RL_COL_MIN_WAVES 6
#endif
#ifndef RL_CFG_L
#error "compile with -DRL_CFG_L=<length>"
#endif

namespace rl {

// Every kernel of this file is launched through here.  When the caller has armed a pair of timing
// events (kernel_table.hpp: launch_timing), the launch records the kernel's own begin and end on
// them -- the interval a rocprofv3 kernel trace reports -- instead of stream-order timestamps.
template <typename K, typename P>
static void rl_launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, P p) {
    LaunchTiming& t = launch_timing();
    if (t.start && t.stop) {
        hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, t.start, t.stop, 0, p);
        t.start = t.stop = nullptr;   // one launch per arming
    } else {
        hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, p);
    }
}

using CF = CfgFor<RL_CFG_L>;
using Cfg = CF::Cfg;                               // row kernels
using CCfg = ColCfgFor<RL_CFG_L>::type;            // column kernels
constexpr int kC32 = CF::C32, kC64 = CF::C64, kQ32 = CF::Q32, kQ64 = CF::Q64;
#define RL_CAT_(a, b) a##b
#define RL_CAT(a, b) RL_CAT_(a, b)
#define RL_TABLE_FN RL_CAT(table_, RL_CFG_L)

// NOTE: the transform length is a template parameter of the kernels so that the
// kernels of different lengths (built in separate translation units) have
// distinct symbol names.
template <int L, int C, int MODE, typename T, bool REALP = false>
__global__ void __launch_bounds__(ColCfgFor<L>::type::T* C, (sizeof(T) == 4 && MODE == COL_PER_IMAGE && WavePrivate<typename ColCfgFor<L>::type>::value) ? RL_COL_MIN_WAVES : 1)
    k_colconv(const ColParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    using KCfg = typename ColCfgFor<L>::type;
    // XCD-aware work order (pure speed heuristic -- any placement is correct).  Workgroups are dealt
    // round-robin over the 8 XCDs, so linear ids l and l+8 share an L2; every XCD gets a contiguous
    // range of the item sequence.  The images of the launch are taken in blocks of G = p.order:
    //     for image block:  for tile pair:  for image in block:  for the two tiles of the pair
    // G = 1 is image-major (tile after tile of one image): concurrent workgroups touch neighbouring
    // 64-B segments of the same spectrum rows.  Larger G re-uses the psf_hat columns of a tile pair
    // for G images in a row: image-major streams the whole 1.33 MB psf_hat through every XCD once
    // per image, past ~3.7 MB of tile traffic in a 4 MiB L2 (half of it is fetched again).
    // G >= images is tile-major: psf_hat stays resident but concurrent workgroups scatter 64-B
    // accesses over all images.
    unsigned bx = blockIdx.x, by = blockIdx.y;
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy;
    if (total % 8 == 0) {
        const unsigned lin = by * gx + bx;
        const unsigned w = (lin % 8) * (total / 8) + lin / 8;
        // images in blocks of G = p.order: within a block, for tile pair: for image: the two tiles
        const unsigned G = p.order < 1 ? 1u : (unsigned)p.order;
        const unsigned blk = w / (gx * G), first = blk * G;
        const unsigned g = gy - first < G ? gy - first : G;          // images in this block
        const unsigned v = w - blk * gx * G;
        const unsigned paired = (gx & ~1u) * g;                       // items of full tile pairs
        if (v < paired) {
            const unsigned pr = v / (2 * g), q = v % (2 * g);
            bx = 2 * pr + (q & 1u);
            by = first + (q >> 1);
        } else {
            bx = gx - 1;
            by = first + (v - paired);
        }
    }
    if constexpr (WavePrivate<KCfg>::value)
        colconv_wave_body<KCfg, C, MODE, T, REALP>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
    else
        colconv_body<KCfg, C, T>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
}

// ---- long column transforms on the wave-private core (conv_kernels.hpp colconv_outer_body) ----
// L = 2304 = 4 x 576 and 4608 = 8 x 576, f32: fft_configs.hpp OuterCol<L>.  The f64 kernels of these lengths stay
// the workgroup-synchronous ones (4 x 9 complex doubles per lane would not fit the register file).
// MODE: COL_PER_IMAGE, or -- outer radix <= 4 -- the multi-view modes, which hold two M x 10 register sets (one 8-wave
// workgroup per CU, like the radix-8 kernel)
template <int L, int C, bool REALP, int MODE = COL_PER_IMAGE>
__global__ void __launch_bounds__(64 * C, MODE == COL_PER_IMAGE ? OuterCol<L>::MIN_WAVES : 2) k_colconv_outer(const ColParams<float> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    using OC = OuterCol<L>;
    unsigned bx = blockIdx.x, by = blockIdx.y;
    const unsigned gx = gridDim.x, total = gridDim.x * gridDim.y;
    if (total % 8 == 0) {   // XCD-contiguous, image-major work order (speed only)
        const unsigned lin = by * gx + bx;
        const unsigned w = (lin % 8) * (total / 8) + lin / 8;
        bx = w % gx;
        by = w / gx;
    }
    colconv_outer_body<typename OC::Core, OC::M, C, float, REALP, MODE>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<float>*>(smem), s);
}
template <int L>
static void fill_outer_twiddles(double* out) {
    using OC = OuterCol<L>;
    using Core = typename OC::Core;
    fill_pass_twiddles<Core>(out);
    double* dst = out + 2 * PassTw<Core, false, 0>::TOTAL;
    for (int q = 1; q < OC::M; ++q)
        for (int k = 0; k < Core::L; ++k) {
            const long double ang = -6.283185307179586476925286766559005768L * (long double)q * (long double)k / (long double)L;
            dst[2 * ((q - 1) * Core::L + k)] = (double)__builtin_cosl(ang);
            dst[2 * ((q - 1) * Core::L + k) + 1] = (double)__builtin_sinl(ang);
        }
}

// waves per SIMD requested from the register allocator (f32, wave-private lengths)
template <int L, int MODE, bool ONEV, typename T>
constexpr int row_min_waves() {
    if (sizeof(T) != 4 || !WavePrivate<typename CfgFor<L>::Cfg>::value) return 1;
    if (MODE == ROW_RATIO) return RL_ROW_MIN_WAVES;
    if (MODE == ROW_UPDATE && ONEV) return RL_UPD_MIN_WAVES;
    return 1;
}

template <int L, int Q, int MODE, bool ONEV, typename T>
__global__ void __launch_bounds__(CfgFor<L>::Cfg::T* Q, (row_min_waves<L, MODE, ONEV, T>()))
    k_rowpass(const RowParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    using KCfg = typename CfgFor<L>::Cfg;
    // XCD-consistent image placement (pure speed heuristic, as in k_colconv): workgroups are dealt
    // round-robin over the 8 XCDs, so give each XCD a contiguous range of (image, row group) items --
    // the same images the column kernel's remap gives it.  A spectrum written by one kernel is then
    // read by the next one through the same L2 (whose contents survive the kernel boundary:
    // tools/l2_probe.hip).
    unsigned bx = blockIdx.x, by = blockIdx.y;
#if RL_ROW_XCD_REMAP
    {
        const unsigned gx = gridDim.x, total = gridDim.x * gridDim.y;
        if (total % 8 == 0) {
            const unsigned lin = by * gx + bx;
            const unsigned w = (lin % 8) * (total / 8) + lin / 8;
            bx = w % gx;
            by = w / gx;
        }
    }
#endif
    // single-view RL modes of the wave-private lengths: the lean item code (scalar row bases,
    // unconditional loads).  RATIO treats every (frame, view) image on its own, so it always qualifies.
    constexpr bool LEAN = RL_ROW_LEAN && WavePrivate<KCfg>::value && (MODE == ROW_RATIO || (MODE == ROW_UPDATE && ONEV));
    if constexpr (LEAN)
        rowlean_body<KCfg, Q, MODE, T>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
    else
        rowpass_body<KCfg, Q, MODE, ONEV, T>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
}

// ---- streaming kernels (wave-private lengths): persistent workgroups, twiddles in LDS ----
#ifndef RL_STREAM_Q32
#define RL_STREAM_Q32 8
#endif
#ifndef RL_STREAM_Q64
#define RL_STREAM_Q64 4
#endif
#ifndef RL_STREAM_ROW_MIN_WAVES
#define RL_STREAM_ROW_MIN_WAVES 1
#endif
#ifndef RL_STREAM_COL_MIN_WAVES
#define RL_STREAM_COL_MIN_WAVES 1
#endif
template <int L, int C, typename T>
__global__ void __launch_bounds__(64 * C, sizeof(T) == 4 ? RL_STREAM_COL_MIN_WAVES : 1) k_colstream(const ColParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    colstream_body<typename ColCfgFor<L>::type, C, T>(p, (int)threadIdx.x, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<cx<T>*>(smem), s);
}
template <int L, int Q, int MODE, typename T>
__global__ void __launch_bounds__(64 * Q, sizeof(T) == 4 ? RL_STREAM_ROW_MIN_WAVES : 1) k_rowstream(const RowParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    rowstream_body<typename CfgFor<L>::Cfg, Q, MODE, T>(p, (int)threadIdx.x, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<cx<T>*>(smem), s);
}

template <int N, typename T>
static constexpr size_t stream_lds_bytes() {
    return ((size_t)N * LdsSlots<Cfg>::value + StreamTw<Cfg>::COUNT) * sizeof(cx<T>);
}
template <int N, typename T>
static constexpr size_t col_stream_lds_bytes() {
    return ((size_t)N * LdsSlots<CCfg>::value + StreamTw<CCfg>::COUNT) * sizeof(cx<T>);
}

// workgroups of `fn` that the device holds at once (0 on error), a multiple of 8 (one share per XCD)
template <typename F>
static int resident_workgroups(F* fn, int threads, size_t lds) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (lds > 65536 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) return 0;
    const int n = per_cu * prop.multiProcessorCount;
    return n >= 8 ? n / 8 * 8 : n;
}

// ---- fused Richardson-Lucy loop (fused_rl.hpp) ----
#ifndef RL_FUSED_NW
#define RL_FUSED_NW 8     // waves per workgroup = columns per tile = row pairs per workgroup round
#endif
#ifndef RL_FUSED_MIN_WAVES
#define RL_FUSED_MIN_WAVES 4   // waves per SIMD asked of the register allocator (2 workgroups of 8 waves per CU)
#endif
template <int L, int NW, bool ACQ>
__global__ void __launch_bounds__(64 * NW, RL_FUSED_MIN_WAVES) k_rl_fused(const FusedParams<float> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using KCfg = typename CfgFor<L>::Cfg;
    constexpr size_t tile = (size_t)NW * LdsSlots<KCfg>::value * sizeof(cx<float>);
    fused_rl_body<KCfg, NW, ACQ, float>(p, reinterpret_cast<cx<float>*>(smem), reinterpret_cast<int*>(smem + tile));
}

template <bool ACQ>
static hipError_t launch_fused_t(const FusedParams<float>& p, int wgs_per_cu, hipStream_t s, int* grid_out) {
    if constexpr (WavePrivate<Cfg>::value && WavePrivate<CCfg>::value) {
        constexpr size_t lds = (size_t)RL_FUSED_NW * LdsSlots<Cfg>::value * sizeof(cx<float>) + 16;
        auto* fn = k_rl_fused<RL_CFG_L, RL_FUSED_NW, ACQ>;
        static const int resident = resident_workgroups(fn, 64 * RL_FUSED_NW, lds);
        if (resident < 8) return hipErrorLaunchFailure;
        int dev = 0, cus = 0;
        hipError_t e;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        int grid = resident;
        if (wgs_per_cu > 0 && wgs_per_cu * cus < grid) grid = wgs_per_cu * cus / 8 * 8;
        if (grid < 8) return hipErrorLaunchFailure;
        if ((e = hipMemsetAsync(p.ctrl, 0, kFusedCtrlWords * sizeof(unsigned), s)) != hipSuccess) return e;
        rl_launch(fn, dim3((unsigned)grid), dim3(64 * RL_FUSED_NW), lds, s, p);
        if (grid_out) *grid_out = grid;
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}
static hipError_t launch_fused(const void* params, int wgs_per_cu, int acquire, hipStream_t s, int* grid_out) {
    const FusedParams<float>& p = *static_cast<const FusedParams<float>*>(params);
    return acquire ? launch_fused_t<true>(p, wgs_per_cu, s, grid_out) : launch_fused_t<false>(p, wgs_per_cu, s, grid_out);
}

template <int C, typename T>
static hipError_t launch_col_stream_t(const void* params, hipStream_t s) {
    if constexpr (WavePrivate<CCfg>::value) {
        const ColParams<T>& p = *static_cast<const ColParams<T>*>(params);
        constexpr size_t lds = col_stream_lds_bytes<C, T>();
        static const int resident = resident_workgroups(k_colstream<RL_CFG_L, C, T>, 64 * C, lds);
        if (resident < 1) return hipErrorLaunchFailure;
        const long total = (long)p.images * ((p.kx + C - 1) / C);
        if (total < 1) return hipSuccess;
        long nwg = total < resident ? total : resident;
        if (nwg >= 8 && total % 8 == 0) nwg = nwg / 8 * 8;
        rl_launch(k_colstream<RL_CFG_L, C, T>, dim3((unsigned)nwg), dim3(64 * C), lds, s, p);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

template <int Q, int MODE, typename T>
static hipError_t launch_row_stream_m(const void* params, hipStream_t s) {
    const RowParams<T>& p = *static_cast<const RowParams<T>*>(params);
    constexpr size_t lds = stream_lds_bytes<Q, T>();
    static const int resident = resident_workgroups(k_rowstream<RL_CFG_L, Q, MODE, T>, 64 * Q, lds);
    if (resident < 1) return hipErrorLaunchFailure;
    const long total = (long)p.frames * ((p.ny + 1) / 2);
    if (total < 1) return hipSuccess;
    const long need = (total + Q - 1) / Q;
    rl_launch(k_rowstream<RL_CFG_L, Q, MODE, T>, dim3((unsigned)(need < resident ? need : resident)), dim3(64 * Q), lds, s, p);
    return hipGetLastError();
}

template <int Q, typename T>
static hipError_t launch_row_stream_t(int mode, const void* params, hipStream_t s) {
    if constexpr (WavePrivate<Cfg>::value) {
        if (mode == ROW_RATIO) return launch_row_stream_m<Q, ROW_RATIO, T>(params, s);
        if (mode == ROW_UPDATE) return launch_row_stream_m<Q, ROW_UPDATE, T>(params, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_col_stream(int dtype, const void* params, hipStream_t s) {
    return dtype == DT_F32 ? launch_col_stream_t<kC32, float>(params, s) : launch_col_stream_t<kC64, double>(params, s);
}
static hipError_t launch_row_stream(int dtype, int mode, const void* params, hipStream_t s) {
    return dtype == DT_F32 ? launch_row_stream_t<RL_STREAM_Q32, float>(mode, params, s)
                           : launch_row_stream_t<RL_STREAM_Q64, double>(mode, params, s);
}

template <int C, typename T>
static constexpr size_t lds_bytes() {
    return (size_t)C * LdsSlots<Cfg>::value * sizeof(cx<T>);
}
template <int C, typename T>
static constexpr size_t col_lds_bytes() {
    return (size_t)C * LdsSlots<CCfg>::value * sizeof(cx<T>);
}

template <int C, typename T>
static hipError_t launch_col_t(const void* params, unsigned gx, unsigned gy, hipStream_t s) {
    const ColParams<T>& p = *static_cast<const ColParams<T>*>(params);
    const dim3 grid(gx, gy), block(CCfg::T * C);
    constexpr size_t lds = col_lds_bytes<C, T>();
    if constexpr (WavePrivate<CCfg>::value) {
        if (p.psf_hat_re) {   // real PSF spectrum
            if (p.mode == COL_H_MULTI) rl_launch(k_colconv<RL_CFG_L, C, COL_H_MULTI, T, true>, grid, block, lds, s, p);
            else if (p.mode == COL_HT_SUM) rl_launch(k_colconv<RL_CFG_L, C, COL_HT_SUM, T, true>, grid, block, lds, s, p);
            else if (p.mode == COL_PER_IMAGE) rl_launch(k_colconv<RL_CFG_L, C, COL_PER_IMAGE, T, true>, grid, block, lds, s, p);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
        if (p.mode == COL_H_MULTI) {
            rl_launch(k_colconv<RL_CFG_L, C, COL_H_MULTI, T>, grid, block, lds, s, p);
            return hipGetLastError();
        }
        if (p.mode == COL_HT_SUM) {
            rl_launch(k_colconv<RL_CFG_L, C, COL_HT_SUM, T>, grid, block, lds, s, p);
            return hipGetLastError();
        }
    }
    if (p.mode != COL_PER_IMAGE) return hipErrorInvalidValue;
    rl_launch(k_colconv<RL_CFG_L, C, COL_PER_IMAGE, T>, grid, block, lds, s, p);
    return hipGetLastError();
}

// frame-pair row kernels (rowpair_body)
template <int L, int Q, int MODE, typename T>
__global__ void __launch_bounds__(CfgFor<L>::Cfg::T * Q, (row_min_waves<L, MODE == ROW_FWD ? ROW_RATIO : MODE, true, T>())) k_rowpair(const RowParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    using KCfg = typename CfgFor<L>::Cfg;
    if constexpr (WavePrivate<KCfg>::value || Q == 1)
        rowpair_body<KCfg, Q, MODE, T>(p, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, reinterpret_cast<cx<T>*>(smem), s);
}
template <int Q, typename T>
static hipError_t launch_row_pair_t(int mode, const void* params, unsigned gy, hipStream_t s) {
    if constexpr (WavePrivate<Cfg>::value || Q == 1) {
        const RowParams<T>& p = *static_cast<const RowParams<T>*>(params);
        const dim3 grid((unsigned)((p.ny + Q - 1) / Q), gy), block(Cfg::T * Q);
        const size_t lds = (size_t)Q * LdsSlots<Cfg>::value * sizeof(cx<T>);
        if (mode == ROW_FWD) rl_launch(k_rowpair<RL_CFG_L, Q, ROW_FWD, T>, grid, block, lds, s, p);
        else if (mode == ROW_RATIO) rl_launch(k_rowpair<RL_CFG_L, Q, ROW_RATIO, T>, grid, block, lds, s, p);
        else if (mode == ROW_UPDATE) rl_launch(k_rowpair<RL_CFG_L, Q, ROW_UPDATE, T>, grid, block, lds, s, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

template <int Q, int MODE, typename T>
static hipError_t launch_row_m(const void* params, unsigned gx, unsigned gy, hipStream_t s) {
    const RowParams<T>& p = *static_cast<const RowParams<T>*>(params);
    constexpr bool MULTI = (MODE == ROW_UPDATE || MODE == ROW_ADJ);
    if (MULTI && p.V == 1)   // single view: variant without accumulator registers
        rl_launch(k_rowpass<RL_CFG_L, Q, MODE, MULTI, T>, dim3(gx, gy), dim3(Cfg::T * Q), lds_bytes<Q, T>(), s, p);
    else
        rl_launch(k_rowpass<RL_CFG_L, Q, MODE, false, T>, dim3(gx, gy), dim3(Cfg::T * Q), lds_bytes<Q, T>(), s, p);
    return hipGetLastError();
}

template <int Q, typename T>
static hipError_t launch_row_t(int mode, const void* params, unsigned gx, unsigned gy, hipStream_t s) {
    switch (mode) {
        case ROW_FWD: return launch_row_m<Q, ROW_FWD, T>(params, gx, gy, s);
        case ROW_INV: return launch_row_m<Q, ROW_INV, T>(params, gx, gy, s);
        case ROW_RATIO: return launch_row_m<Q, ROW_RATIO, T>(params, gx, gy, s);
        case ROW_UPDATE: return launch_row_m<Q, ROW_UPDATE, T>(params, gx, gy, s);
        case ROW_ADJ: return launch_row_m<Q, ROW_ADJ, T>(params, gx, gy, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_col(int dtype, const void* params, unsigned gx, unsigned gy, hipStream_t s) {
    if constexpr (OuterCol<RL_CFG_L>::value) {
        if (dtype == DT_F32) {
            using OC = OuterCol<RL_CFG_L>;
            const ColParams<float>& p = *static_cast<const ColParams<float>*>(params);
            constexpr size_t lds = (size_t)OC::C * LdsSlots<typename OC::Core>::value * sizeof(cx<float>);
            const dim3 grid((unsigned)((p.kx + OC::C - 1) / OC::C), gy), block(64 * OC::C);
            if constexpr (OC::MULTI) {
                if (p.mode == COL_H_MULTI) {
                    if (p.psf_hat_re) rl_launch(k_colconv_outer<RL_CFG_L, OC::C, true, COL_H_MULTI>, grid, block, lds, s, p);
                    else rl_launch(k_colconv_outer<RL_CFG_L, OC::C, false, COL_H_MULTI>, grid, block, lds, s, p);
                    return hipGetLastError();
                }
                if (p.mode == COL_HT_SUM) {
                    if (p.psf_hat_re) rl_launch(k_colconv_outer<RL_CFG_L, OC::C, true, COL_HT_SUM>, grid, block, lds, s, p);
                    else rl_launch(k_colconv_outer<RL_CFG_L, OC::C, false, COL_HT_SUM>, grid, block, lds, s, p);
                    return hipGetLastError();
                }
            }
            if (p.mode != COL_PER_IMAGE) return hipErrorInvalidValue;
            if (p.psf_hat_re) rl_launch(k_colconv_outer<RL_CFG_L, OC::C, true>, grid, block, lds, s, p);
            else rl_launch(k_colconv_outer<RL_CFG_L, OC::C, false>, grid, block, lds, s, p);
            return hipGetLastError();
        }
    }
    return dtype == DT_F32 ? launch_col_t<kC32, float>(params, gx, gy, s)
                           : launch_col_t<kC64, double>(params, gx, gy, s);
}

static hipError_t launch_row(int dtype, int mode, const void* params, unsigned gx, unsigned gy, hipStream_t s) {
    return dtype == DT_F32 ? launch_row_t<kQ32, float>(mode, params, gx, gy, s)
                           : launch_row_t<kQ64, double>(mode, params, gx, gy, s);
}
// rows (= waves) per workgroup of the frame-pair row kernels, f32 (RL_PAIR_Q32; measured at 512^2: 2 / 4 / 8 / 16 rows
// 18.4 / 18.8 / 19.1 / ... k frames/s)
#ifndef RL_PAIR_Q32
#define RL_PAIR_Q32 8
#endif
constexpr int kPairQ32 = WavePrivate<Cfg>::value ? RL_PAIR_Q32 : kQ32;
// frame-pair row kernels exist for one transform per wave and for one workgroup-synchronous transform per workgroup
constexpr bool kPairRows = WavePrivate<Cfg>::value || (kQ32 == 1 && kQ64 == 1);
static hipError_t launch_row_pair(int dtype, int mode, const void* params, unsigned gy, hipStream_t s) {
    return dtype == DT_F32 ? launch_row_pair_t<kPairQ32, float>(mode, params, gy, s)
                           : launch_row_pair_t<kQ64, double>(mode, params, gy, s);
}

template <typename F>
static hipError_t allow_lds(F* fn, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int Q, typename T>
static hipError_t prepare_rows() {
    hipError_t e;
    const size_t b = lds_bytes<Q, T>();
    if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_FWD, false, T>, b)) != hipSuccess) return e;
    if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_INV, false, T>, b)) != hipSuccess) return e;
    if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_RATIO, false, T>, b)) != hipSuccess) return e;
    if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_UPDATE, false, T>, b)) != hipSuccess) return e;
    if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_UPDATE, true, T>, b)) != hipSuccess) return e;
    if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_ADJ, false, T>, b)) != hipSuccess) return e;
    if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_ADJ, true, T>, b)) != hipSuccess) return e;
    if constexpr (kPairRows) {
        constexpr int QP = sizeof(T) == 4 ? kPairQ32 : Q;
        const size_t bp = lds_bytes<QP, T>();
        if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_FWD, T>, bp)) != hipSuccess) return e;
        if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_RATIO, T>, bp)) != hipSuccess) return e;
        if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_UPDATE, T>, bp)) != hipSuccess) return e;
    }
    return hipSuccess;
}

static hipError_t prepare() {
    hipError_t e;
    if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
    if ((e = allow_lds(k_colconv<RL_CFG_L, kC64, COL_PER_IMAGE, double>, col_lds_bytes<kC64, double>())) != hipSuccess) return e;
    if constexpr (WavePrivate<CCfg>::value) {
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float, true>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC64, COL_PER_IMAGE, double, true>, col_lds_bytes<kC64, double>())) != hipSuccess) return e;
    }
    if constexpr (OuterCol<RL_CFG_L>::value) {
        using OC = OuterCol<RL_CFG_L>;
        constexpr size_t lds = (size_t)OC::C * LdsSlots<typename OC::Core>::value * sizeof(cx<float>);
        if ((e = allow_lds(k_colconv_outer<RL_CFG_L, OC::C, true>, lds)) != hipSuccess) return e;
        if ((e = allow_lds(k_colconv_outer<RL_CFG_L, OC::C, false>, lds)) != hipSuccess) return e;
        if constexpr (OC::MULTI) {
            if ((e = allow_lds(k_colconv_outer<RL_CFG_L, OC::C, true, COL_H_MULTI>, lds)) != hipSuccess) return e;
            if ((e = allow_lds(k_colconv_outer<RL_CFG_L, OC::C, false, COL_H_MULTI>, lds)) != hipSuccess) return e;
            if ((e = allow_lds(k_colconv_outer<RL_CFG_L, OC::C, true, COL_HT_SUM>, lds)) != hipSuccess) return e;
            if ((e = allow_lds(k_colconv_outer<RL_CFG_L, OC::C, false, COL_HT_SUM>, lds)) != hipSuccess) return e;
        }
    }
    if ((e = prepare_rows<kQ32, float>()) != hipSuccess) return e;
    if ((e = prepare_rows<kQ64, double>()) != hipSuccess) return e;
    return hipSuccess;
}

// (the length is a template parameter: this file is compiled once per length, and equally named entities of
// different translation units would be merged by the linker)
template <int L, bool OUTER>
struct OuterTw {   // column twiddle table of the f32 kernel: the outer-decimation kernel's, where the length has one
    static constexpr int count = PassTw<typename ColCfgFor<L>::type, false, 0>::TOTAL;
    static void fill(double* out) { fill_pass_twiddles<typename ColCfgFor<L>::type>(out); }
};
template <int L>
struct OuterTw<L, true> {
    using OC = OuterCol<L>;
    static constexpr int count = PassTw<typename OC::Core, false, 0>::TOTAL + (OC::M - 1) * OC::Core::L;
    static void fill(double* out) { fill_outer_twiddles<L>(out); }
};

const KernelTable* RL_TABLE_FN() {
    constexpr bool OUTER = OuterCol<RL_CFG_L>::value;
    constexpr int WP = WavePrivate<CCfg>::value ? 1 : 0;
    static const KernelTable t = {Cfg::L, Cfg::T, {OUTER ? OuterCol<RL_CFG_L>::C : kC32, kC64}, {kQ32, kQ64},
                                  {OUTER ? 1 : WP, WP}, {OUTER ? (OuterCol<RL_CFG_L>::MULTI ? 1 : 0) : WP, WP},
                                  PassTw<Cfg, false, 0>::TOTAL, fill_pass_twiddles<Cfg>,
                                  {OuterTw<RL_CFG_L, OUTER>::count, PassTw<CCfg, false, 0>::TOTAL},
                                  {OuterTw<RL_CFG_L, OUTER>::fill, fill_pass_twiddles<CCfg>}, launch_col, launch_row, prepare,
                                  WavePrivate<CCfg>::value ? launch_col_stream : nullptr,
                                  WavePrivate<Cfg>::value ? launch_row_stream : nullptr,
                                  (WavePrivate<Cfg>::value && WavePrivate<CCfg>::value) ? launch_fused : nullptr,
                                  kPairRows ? launch_row_pair : nullptr};
    return &t;
}

}  // namespace rl
