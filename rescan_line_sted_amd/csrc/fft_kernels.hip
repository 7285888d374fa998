// fft_kernels.hip -- gfx950 kernels for one FFT length (compile with
// -DRL_CFG_L=<L>).  Column pass (FFT_y * psf_hat -> IFFT_y) and the fused row
// passes (IFFT_x -> Richardson-Lucy pointwise step -> FFT_x) of the
// convolution path; bodies in conv_kernels.hpp.
#include <hip/hip_ext.h>
#include "kernel_table.hpp"
#include "conv_kernels.hpp"
#include "fft_configs.hpp"
#include "dev_sync.hpp"

// waves/SIMD requested for the f32 ROW_RATIO kernel (needs <= 96 VGPRs, which it has
// within 2 registers; the other modes spill under that bound and are left alone)
#ifndef RL_ROW_MIN_WAVES
#define RL_ROW_MIN_WAVES 5
#endif
#ifndef RL_UPD_MIN_WAVES
#define RL_UPD_MIN_WAVES 1
#endif
// (waves per SIMD requested for the long row kernels: 6 / 5 for RATIO / UPDATE -- five or six workgroups per CU instead of four --
// measured 2048^2 point 781 -> 712 frames/s, 4 views 247 -> 216: left to the compiler)
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

template <typename F>
static hipError_t allow_lds(F* fn, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}


using CF = CfgFor<RL_CFG_L>;
using Cfg = CF::Cfg;                               // row kernels
using CCfg = ColCfgFor<RL_CFG_L>::type;            // column kernels
constexpr int kC32 = CF::C32, kC64 = CF::C64, kQ32 = CF::Q32, kQ64 = CF::Q64;
#define RL_CAT_(a, b) a##b
#define RL_CAT(a, b) RL_CAT_(a, b)
#define RL_TABLE_FN RL_CAT(table_, RL_CFG_L)

// Kernels specialised for the 512 x 512 frames of the BASELINE headline (L = 576, f32): row / column counts at compile time.
#ifndef RL_N512
#define RL_N512 1
#endif
template <typename T>
constexpr bool kColN512 = RL_N512 != 0 && RL_CFG_L == 576 && sizeof(T) == 4;
// ... and the frame-pair row kernels of L = 2304 for 2048-pixel rows (register slots of 256 pixels: 8 of 9 hold pixels)
template <typename T>
constexpr bool kRowN2048 = RL_N512 != 0 && RL_CFG_L == 2304 && sizeof(T) == 4;


// NOTE: the transform length is a template parameter of the kernels so that the
// kernels of different lengths (built in separate translation units) have
// distinct symbol names.
template <int L, int C, int MODE, typename T, bool REALP = false, int NYC = 0, int CT = 0>
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
        colconv_wave_body<KCfg, C, MODE, T, REALP, NYC, CT>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
    else
        colconv_body<KCfg, C, T>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
}

// ---- long column transforms on the wave-private core (conv_kernels.hpp colconv_outer_body) ----
// L = 2304 = 4 x 576 and 4608 = 8 x 576, f32: fft_configs.hpp OuterCol<L>.  The f64 kernels of these lengths stay
// the workgroup-synchronous ones (4 x 9 complex doubles per lane would not fit the register file).
// complex LDS entries of the twiddle copy (conv_kernels.hpp colconv_outer_body TWLDS)
template <class OC>
constexpr size_t outer_tw_lds_elems(int twlds) {
    return (twlds > 0 ? PassTw<typename OC::Core, false, 0>::TOTAL : 0) + (twlds > 1 ? (OC::M - 1) * OC::Core::L : 0);
}
template <class OC>
constexpr size_t outer_whole_lds_bytes() {
    return ((size_t)OC::CW * LdsSlots<typename OC::Core>::value + (size_t)OC::PARK * 64 * OC::CW + outer_tw_lds_elems<OC>(OC::TWLDS)) * sizeof(cx<float>);
}
// NYC: the image's row count at compile time (conv_kernels.hpp colconv_outer_body): instantiated for M x 512 rows -- the
// 1024 / 2048 / 4096-row images whose residue classes are the 512-of-576 case of the core
template <int L, int C, bool REALP, int MODE = COL_PER_IMAGE, typename T = float, int NYC = 0>
__global__ void __launch_bounds__(64 * C, (sizeof(T) == 4 ? OuterCol<L>::MIN_WAVES : OuterCol<L>::MIN_WAVES64)) k_colconv_outer(const ColParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    using OC = OuterCol<L>;
    static_assert(sizeof(T) == 4 || MODE == COL_PER_IMAGE, "float64: the whole pass only");
    unsigned bx = blockIdx.x, by = blockIdx.y;
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gridDim.x * gridDim.y;
    if (total % 8 == 0) {   // XCD-contiguous work order (speed only)
        const unsigned lin = by * gx + bx;
        const unsigned w = (lin % 8) * (total / 8) + lin / 8;
        if constexpr (MODE == COL_SPLIT_INV || MODE == COL_SPLIT_INV_SUM) {
            // tile-major: the images of one column tile (views fastest, then frames) follow each other on one XCD, so the tile's
            // multipliers (and, COL_SPLIT_INV, the frame's parked spectra its V views share) are fetched once and then hit in L2
            bx = w / gy;
            by = w % gy;
        } else {   // image-major
            bx = w % gx;
            by = w / gx;
        }
    }
    if constexpr (sizeof(T) == 4)
        colconv_outer_body<typename OC::Core, OC::M, C, float, REALP, MODE, (MODE == COL_PER_IMAGE ? OC::PARK : 0), (MODE == COL_PER_IMAGE ? OC::TWLDS : OC::TWLDS_SPLIT), NYC>(
            p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<float>*>(smem), s);
    else
        colconv_outer_body<typename OC::Core, OC::M, C, double, REALP, COL_PER_IMAGE, OC::PARK64, 0, NYC>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<double>*>(smem), s);
}
// launch either the generic kernel or -- M x 512 rows, pitch a multiple of the tile width -- the one with the row count at compile time
template <int L, int C, bool REALP, int MODE, typename T>
static void launch_outer(const ColParams<T>& p, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    constexpr int NY = 512 * OuterCol<L>::M;
    if (RL_N512 != 0 && p.ny == NY && p.pitch % C == 0) rl_launch(k_colconv_outer<L, C, REALP, MODE, T, NY>, grid, block, lds, s, p);
    else rl_launch(k_colconv_outer<L, C, REALP, MODE, T>, grid, block, lds, s, p);
}
template <int L, int C, bool REALP, int MODE, typename T>
static hipError_t allow_outer(size_t lds) {
    hipError_t e = allow_lds(k_colconv_outer<L, C, REALP, MODE, T>, lds);
    if (e == hipSuccess) e = allow_lds(k_colconv_outer<L, C, REALP, MODE, T, 512 * OuterCol<L>::M>, lds);
    return e;
}
template <class OC>
constexpr size_t outer_whole_lds_bytes_f64() {
    return ((size_t)OC::C64 * LdsSlots<typename OC::Core>::value + (size_t)OC::PARK64 * 64 * OC::C64) * sizeof(cx<double>);
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
    if (sizeof(T) == 4 && !WavePrivate<typename CfgFor<L>::Cfg>::value) {   // the long lengths (256 threads per transform)
        if (MODE == ROW_RATIO) return 1;
        if (MODE == ROW_UPDATE && ONEV) return 1;
        return 1;
    }
    if (sizeof(T) != 4) return 1;
    if (MODE == ROW_RATIO) return RL_ROW_MIN_WAVES;
    if (MODE == ROW_UPDATE && ONEV) return RL_UPD_MIN_WAVES;
    return 1;
}

template <int L, int Q, int MODE, bool ONEV, typename T, bool PRESUM = false, int NXC = 0, int SUBC = -1>
__global__ void __launch_bounds__(CfgFor<L>::Cfg::T* Q, (row_min_waves<L, MODE, ONEV, T>()))
    k_rowpass(const RowParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    using KCfg = typename CfgFor<L>::Cfg;
    // (An XCD-consistent remap of (image, row group) items like k_colconv's measured neutral on time and cost traffic -- every
    // XCD's L2 then streams the whole normaliser instead of the eighth its row groups touch: removed.)
    const unsigned bx = blockIdx.x, by = blockIdx.y;
    // single-view RL modes of the wave-private lengths: the lean item code (scalar row bases,
    // unconditional loads).  RATIO treats every (frame, view) image on its own, so it always qualifies.
    constexpr bool LEAN = !PRESUM && WavePrivate<KCfg>::value && (MODE == ROW_RATIO || (MODE == ROW_UPDATE && ONEV));
    static_assert(NXC == 0 || LEAN, "the compile-time row length exists for the lean bodies");
    if constexpr (LEAN)
        rowlean_body<KCfg, Q, MODE, T, NXC, SUBC>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
    else
        rowpass_body<KCfg, Q, MODE, ONEV, T, PRESUM>(p, (int)threadIdx.x, (int)bx, (int)by, reinterpret_cast<cx<T>*>(smem), s);
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
        if constexpr (kColN512<T>) {   // 512 rows exactly, one view: the specialised kernels (conv_kernels.hpp colconv_wave_body NYC)
            if (p.mode == COL_PER_IMAGE && p.ny == 512 && p.V == 1 && p.pitch % C == 0) {
                if (p.residual && RL_CT_RESIDUAL) {   // the spectrum is that of `ratio - 1`: compact twiddles
                    if (p.psf_hat_re) rl_launch(k_colconv<RL_CFG_L, C, COL_PER_IMAGE, T, true, 512, 1>, grid, block, lds, s, p);
                    else rl_launch(k_colconv<RL_CFG_L, C, COL_PER_IMAGE, T, false, 512, 1>, grid, block, lds, s, p);
                    return hipGetLastError();
                }
                if (p.psf_hat_re) rl_launch(k_colconv<RL_CFG_L, C, COL_PER_IMAGE, T, true, 512>, grid, block, lds, s, p);
                else rl_launch(k_colconv<RL_CFG_L, C, COL_PER_IMAGE, T, false, 512>, grid, block, lds, s, p);
                return hipGetLastError();
            }
        }
        if constexpr (kColN512<T>) {   // the fused multi-view modes on 512-row images (real multiplier: what the reference's PSFs run)
            if (p.psf_hat_re && p.ny == 512 && p.pitch % C == 0 && (p.mode == COL_H_MULTI || p.mode == COL_HT_SUM)) {
                if (p.mode == COL_H_MULTI) rl_launch(k_colconv<RL_CFG_L, C, COL_H_MULTI, T, true, 512>, grid, block, lds, s, p);
                else if (p.residual && RL_CT_RESIDUAL) rl_launch(k_colconv<RL_CFG_L, C, COL_HT_SUM, T, true, 512, 1>, grid, block, lds, s, p);   // spectra of `ratio - 1`
                else rl_launch(k_colconv<RL_CFG_L, C, COL_HT_SUM, T, true, 512>, grid, block, lds, s, p);
                return hipGetLastError();
            }
        }
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
template <int L, int Q, int MODE, typename T, int NXC = 0, int SUBC = -1>
__global__ void __launch_bounds__(CfgFor<L>::Cfg::T * Q, (row_min_waves<L, MODE == ROW_FWD ? ROW_RATIO : MODE, true, T>())) k_rowpair(const RowParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevSync s;
    using KCfg = typename CfgFor<L>::Cfg;
    if constexpr (WavePrivate<KCfg>::value || Q == 1)
        rowpair_body<KCfg, Q, MODE, T, NXC, SUBC>(p, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, reinterpret_cast<cx<T>*>(smem), s);
}
template <int Q, typename T>
static hipError_t launch_row_pair_t(int mode, const void* params, unsigned gy, hipStream_t s) {
    if constexpr (WavePrivate<Cfg>::value || Q == 1) {
        const RowParams<T>& p = *static_cast<const RowParams<T>*>(params);
        const dim3 grid((unsigned)((p.ny + Q - 1) / Q), gy), block(Cfg::T * Q);
        const size_t lds = lds_bytes<Q, T>();
        if constexpr (kColN512<T>) {   // 512-pixel rows, one view, `ratio - 1`: the specialised kernels (rowpair_body NXC / SUBC)
            if (p.nx == 512 && p.V == 1 && p.sub_one != 0 && mode != ROW_FWD) {
                if (mode == ROW_RATIO) rl_launch(k_rowpair<RL_CFG_L, Q, ROW_RATIO, T, 512, 1>, grid, block, lds, s, p);
                else if (mode == ROW_UPDATE) rl_launch(k_rowpair<RL_CFG_L, Q, ROW_UPDATE, T, 512, 1>, grid, block, lds, s, p);
                else return hipErrorInvalidValue;
                return hipGetLastError();
            }
        }
        if constexpr (kRowN2048<T>) {
            if (p.nx == 2048 && p.V == 1 && p.sub_one != 0 && mode != ROW_FWD) {
                if (mode == ROW_RATIO) rl_launch(k_rowpair<RL_CFG_L, Q, ROW_RATIO, T, 2048, 1>, grid, block, lds, s, p);
                else if (mode == ROW_UPDATE) rl_launch(k_rowpair<RL_CFG_L, Q, ROW_UPDATE, T, 2048, 1>, grid, block, lds, s, p);
                else return hipErrorInvalidValue;
                return hipGetLastError();
            }
        }
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
    if constexpr (MODE == ROW_UPDATE) {
        if (p.V > 1 && p.sub_one) {   // the views' residual spectra are summed on their way in: one inverse transform (rowpass_body PRESUM)
            rl_launch(k_rowpass<RL_CFG_L, Q, MODE, true, T, true>, dim3(gx, gy), dim3(Cfg::T * Q), lds_bytes<Q, T>(), s, p);
            return hipGetLastError();
        }
    }
    if constexpr (kColN512<T> && (MODE == ROW_RATIO || MODE == ROW_UPDATE)) {   // per-frame lean bodies on 512-pixel rows (multi-view plans' RATIO,
        if (p.nx == 512 && p.sub_one != 0 && (MODE == ROW_RATIO || p.V == 1)) {   // the single-spectrum UPDATE behind the column view sum, f32)
            rl_launch(k_rowpass<RL_CFG_L, Q, MODE, MODE == ROW_UPDATE, T, false, 512, 1>, dim3(gx, gy), dim3(Cfg::T * Q), lds_bytes<Q, T>(), s, p);
            return hipGetLastError();
        }
    }
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
            constexpr size_t lds_split = lds + outer_tw_lds_elems<OC>(OC::TWLDS_SPLIT) * sizeof(cx<float>);
            if (p.mode == COL_SPLIT_FWD) {
                launch_outer<RL_CFG_L, OC::C, false, COL_SPLIT_FWD, float>(p, grid, block, lds_split, s);
                return hipGetLastError();
            }
            if (p.mode == COL_SPLIT_INV) {
                if (p.psf_hat_re) launch_outer<RL_CFG_L, OC::C, true, COL_SPLIT_INV, float>(p, grid, block, lds_split, s);
                else launch_outer<RL_CFG_L, OC::C, false, COL_SPLIT_INV, float>(p, grid, block, lds_split, s);
                return hipGetLastError();
            }
            if (p.mode == COL_SPLIT_INV_SUM) {
                if (p.psf_hat_re) launch_outer<RL_CFG_L, OC::C, true, COL_SPLIT_INV_SUM, float>(p, grid, block, lds_split, s);
                else launch_outer<RL_CFG_L, OC::C, false, COL_SPLIT_INV_SUM, float>(p, grid, block, lds_split, s);
                return hipGetLastError();
            }
            if (p.mode != COL_PER_IMAGE) return hipErrorInvalidValue;
            // the whole pass has a tile width of its own (OC::CW), its transform regions + parking space + twiddle copies
            constexpr size_t lds_whole = outer_whole_lds_bytes<OC>();
            const dim3 grid_w((unsigned)((p.kx + OC::CW - 1) / OC::CW), gy), block_w(64 * OC::CW);
            if (p.psf_hat_re) launch_outer<RL_CFG_L, OC::CW, true, COL_PER_IMAGE, float>(p, grid_w, block_w, lds_whole, s);
            else launch_outer<RL_CFG_L, OC::CW, false, COL_PER_IMAGE, float>(p, grid_w, block_w, lds_whole, s);
            return hipGetLastError();
        }
    }
    if constexpr (OuterCol<RL_CFG_L>::value64) {
        if (dtype != DT_F32) {   // float64: the whole pass on the outer-decimation body (per image; multi-view plans launch it per view)
            using OC = OuterCol<RL_CFG_L>;
            const ColParams<double>& p = *static_cast<const ColParams<double>*>(params);
            if (p.mode != COL_PER_IMAGE) return hipErrorInvalidValue;
            constexpr size_t lds = outer_whole_lds_bytes_f64<OC>();
            const dim3 grid((unsigned)((p.kx + OC::C64 - 1) / OC::C64), gy), block(64 * OC::C64);
            if (p.psf_hat_re) launch_outer<RL_CFG_L, OC::C64, true, COL_PER_IMAGE, double>(p, grid, block, lds, s);
            else launch_outer<RL_CFG_L, OC::C64, false, COL_PER_IMAGE, double>(p, grid, block, lds, s);
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
    if constexpr (kColN512<T>) {
        if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_RATIO, false, T, false, 512, 1>, b)) != hipSuccess) return e;
        if ((e = allow_lds(k_rowpass<RL_CFG_L, Q, ROW_UPDATE, true, T, false, 512, 1>, b)) != hipSuccess) return e;
    }
    if constexpr (kPairRows) {
        constexpr int QP = sizeof(T) == 4 ? kPairQ32 : Q;
        const size_t bp = lds_bytes<QP, T>();
        if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_FWD, T>, bp)) != hipSuccess) return e;
        if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_RATIO, T>, bp)) != hipSuccess) return e;
        if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_UPDATE, T>, bp)) != hipSuccess) return e;
        if constexpr (kColN512<T>) {
            if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_RATIO, T, 512, 1>, bp)) != hipSuccess) return e;
            if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_UPDATE, T, 512, 1>, bp)) != hipSuccess) return e;
        }
        if constexpr (kRowN2048<T>) {
            if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_RATIO, T, 2048, 1>, bp)) != hipSuccess) return e;
            if ((e = allow_lds(k_rowpair<RL_CFG_L, QP, ROW_UPDATE, T, 2048, 1>, bp)) != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

static hipError_t prepare() {
    hipError_t e;
    if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
    if ((e = allow_lds(k_colconv<RL_CFG_L, kC64, COL_PER_IMAGE, double>, col_lds_bytes<kC64, double>())) != hipSuccess) return e;
    if constexpr (kColN512<float>) {
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float, true, 512>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float, false, 512>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float, true, 512, 1>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float, false, 512, 1>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
    }
    if constexpr (WavePrivate<CCfg>::value) {
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC32, COL_PER_IMAGE, float, true>, col_lds_bytes<kC32, float>())) != hipSuccess) return e;
        if ((e = allow_lds(k_colconv<RL_CFG_L, kC64, COL_PER_IMAGE, double, true>, col_lds_bytes<kC64, double>())) != hipSuccess) return e;
    }
    if constexpr (OuterCol<RL_CFG_L>::value) {
        using OC = OuterCol<RL_CFG_L>;
        constexpr size_t lds = (size_t)OC::C * LdsSlots<typename OC::Core>::value * sizeof(cx<float>);
        constexpr size_t lds_whole = outer_whole_lds_bytes<OC>();
        static_assert(lds_whole <= 160 * 1024, "LDS of a CU");
        if ((e = allow_outer<RL_CFG_L, OC::CW, true, COL_PER_IMAGE, float>(lds_whole)) != hipSuccess) return e;
        if ((e = allow_outer<RL_CFG_L, OC::CW, false, COL_PER_IMAGE, float>(lds_whole)) != hipSuccess) return e;
        constexpr size_t lds_split = lds + outer_tw_lds_elems<OC>(OC::TWLDS_SPLIT) * sizeof(cx<float>);
        if ((e = allow_outer<RL_CFG_L, OC::C, false, COL_SPLIT_FWD, float>(lds_split)) != hipSuccess) return e;
        if ((e = allow_outer<RL_CFG_L, OC::C, true, COL_SPLIT_INV, float>(lds_split)) != hipSuccess) return e;
        if ((e = allow_outer<RL_CFG_L, OC::C, false, COL_SPLIT_INV, float>(lds_split)) != hipSuccess) return e;
        if ((e = allow_outer<RL_CFG_L, OC::C, true, COL_SPLIT_INV_SUM, float>(lds_split)) != hipSuccess) return e;
        if ((e = allow_outer<RL_CFG_L, OC::C, false, COL_SPLIT_INV_SUM, float>(lds_split)) != hipSuccess) return e;
        if constexpr (OC::value64) {
            constexpr size_t lds64 = outer_whole_lds_bytes_f64<OC>();
            static_assert(lds64 <= 160 * 1024, "LDS of a CU");
            if ((e = allow_outer<RL_CFG_L, OC::C64, true, COL_PER_IMAGE, double>(lds64)) != hipSuccess) return e;
            if ((e = allow_outer<RL_CFG_L, OC::C64, false, COL_PER_IMAGE, double>(lds64)) != hipSuccess) return e;
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
    static constexpr size_t split_tile = 0;
};
template <int L>
struct OuterTw<L, true> {
    using OC = OuterCol<L>;
    static constexpr int count = PassTw<typename OC::Core, false, 0>::TOTAL + (OC::M - 1) * OC::Core::L;
    static void fill(double* out) { fill_outer_twiddles<L>(out); }
    static constexpr size_t split_tile = OC::SPLIT ? outer_slots_tile_elems<typename OC::Core, OC::M, OC::C>() : 0;
};

const KernelTable* RL_TABLE_FN() {
    constexpr bool OUTER = OuterCol<RL_CFG_L>::value;
    constexpr int WP = WavePrivate<CCfg>::value ? 1 : 0;
    constexpr bool OUTER64 = OuterCol<RL_CFG_L>::value64;   // float64 column pass on the outer-decimation body too
    static const KernelTable t = {Cfg::L, Cfg::T, {OUTER ? OuterCol<RL_CFG_L>::C : kC32, OUTER64 ? OuterCol<RL_CFG_L>::C64 : kC64}, {kQ32, kQ64},
                                  {OUTER ? 1 : WP, OUTER64 ? 1 : WP}, {OUTER ? 0 : 3 * WP, OUTER64 ? 0 : 3 * WP},
                                  PassTw<Cfg, false, 0>::TOTAL, fill_pass_twiddles<Cfg>,
                                  {OuterTw<RL_CFG_L, OUTER>::count, OUTER64 ? OuterTw<RL_CFG_L, OUTER64>::count : PassTw<CCfg, false, 0>::TOTAL},
                                  {OuterTw<RL_CFG_L, OUTER>::fill, OUTER64 ? OuterTw<RL_CFG_L, OUTER64>::fill : fill_pass_twiddles<CCfg>}, launch_col, launch_row, prepare,
                                  kPairRows ? launch_row_pair : nullptr, OuterTw<RL_CFG_L, OUTER>::split_tile};
    return &t;
}

}  // namespace rl
