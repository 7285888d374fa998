// Host emulator for the workgroup bodies in rescan_line_sted_amd/csrc/
// conv_kernels.hpp: runs one OS thread per GPU thread with a pthread barrier
// standing in for __syncthreads().  TEST INFRASTRUCTURE ONLY -- it exists so
// the index logic of the HIP kernels (Stockham passes, two-rows-per-transform
// packing, pad/crop, view loops) can be checked against numpy on a machine
// without a GPU.  It is built by tests/test_emulated_kernels.py with g++ and
// is never loaded by the product.
#include <pthread.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#define RL_TILE_WIDE_F64 1   // the device runs the two-column tile I/O in float only; here its index logic is tested in double
#include "../../rescan_line_sted_amd/csrc/conv_kernels.hpp"
#include "../../rescan_line_sted_amd/csrc/fft_configs.hpp"
#include "../../rescan_line_sted_amd/csrc/philox_poisson.hpp"

using namespace rl;

struct EmuSync {
    pthread_barrier_t* bar;        // whole workgroup
    pthread_barrier_t* wave_bar;   // the 64 threads of this thread's wavefront
    double* xchg;                  // 64 slots shared by the wavefront (cross-lane shuffles)
    int lane;
    void wg() const { pthread_barrier_wait(bar); }
    void wave() const { pthread_barrier_wait(wave_bar); }
    template <int MASK>
    double shfl_xor(double v) const {
        xchg[lane] = v;
        pthread_barrier_wait(wave_bar);
        const double o = xchg[lane ^ MASK];
        pthread_barrier_wait(wave_bar);
        return o;
    }
    template <int MASK>
    float shfl_xor(float v) const { return (float)shfl_xor<MASK>((double)v); }
    // radix-2 exchange stage (DevSync::bfly): MASK bit clear -> x + partner, set -> partner - x
    template <int MASK, typename T>
    void bfly(cx<T>& x, int l) const {
        const T pr = shfl_xor<MASK>(x.re), pi = shfl_xor<MASK>(x.im);
        if (l & MASK) x = mk<T>(pr - x.re, pi - x.im);
        else x = mk<T>(x.re + pr, x.im + pi);
    }
};

template <class Body>
static void run_grid(int gx, int gy, int nthreads, size_t lds_bytes, Body body) {
    std::vector<unsigned char> lds(lds_bytes + 64);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, nthreads);
    const int nwaves = (nthreads + 63) / 64;
    std::vector<pthread_barrier_t> wbar(nwaves);
    std::vector<double> xchg((size_t)nwaves * 64);
    for (int w = 0; w < nwaves; ++w) {
        const int n = (w + 1) * 64 <= nthreads ? 64 : nthreads - w * 64;
        pthread_barrier_init(&wbar[w], nullptr, n);
    }
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx) {
            std::memset(lds.data(), 0xff, lds.size());   // poison: NaNs if read before written
            std::vector<std::thread> th;
            th.reserve(nthreads);
            for (int tid = 0; tid < nthreads; ++tid)
                th.emplace_back([&, tid]() {
                    EmuSync s{&bar, &wbar[tid / 64], &xchg[(size_t)(tid / 64) * 64], tid % 64};
                    body(tid, bx, by, lds.data(), s);
                });
            for (auto& t : th) t.join();
        }
    pthread_barrier_destroy(&bar);
    for (auto& b : wbar) pthread_barrier_destroy(&b);
}

template <class Cfg, typename T>
static std::vector<cx<T>> twiddles_of() {   // the per-pass table the device plan uploads for one geometry
    constexpr int n = PassTw<Cfg, false, 0>::TOTAL;
    std::vector<double> h(2 * (size_t)(n > 0 ? n : 1), 0.0);
    if (n > 0) fill_pass_twiddles<Cfg>(h.data());
    std::vector<cx<T>> tw(n > 0 ? n : 1);
    for (size_t i = 0; i < tw.size(); ++i) tw[i] = mk<T>((T)h[2 * i], (T)h[2 * i + 1]);
    return tw;
}
template <int L, typename T>
static std::vector<cx<T>> twiddles(bool column = false) {   // rows: CfgFor<L>::Cfg, columns: ColCfgFor<L>
    if (column) return twiddles_of<typename ColCfgFor<L>::type, T>();
    return twiddles_of<typename CfgFor<L>::Cfg, T>();
}

// emu_set_special(1): the bodies specialised for a compile-time image size (round 4: NYC / NXC / SUBC / CT template arguments,
// on the device instantiated for the 512 x 512 frames at L = 576) are run in their L = 256, 192-row / 192-pixel instantiation
static int g_special = 0;

template <int L, typename T>
static int col_t(const T* in, T* out, const T* psf_hat, int ny, int kx, int pitch, int V, int frames,
                 int in_sb, int in_sv, int mode) {
    using CF = CfgFor<L>;
    using Cfg = typename ColCfgFor<L>::type;      // the column kernels' geometry
    constexpr int C = sizeof(T) == 4 ? CF::C32 : CF::C64;
    auto tw = twiddles<L, T>(true);
    ColParams<T> p;
    p.in = reinterpret_cast<const cx<T>*>(in);
    p.out = reinterpret_cast<cx<T>*>(out);
    p.psf_hat = reinterpret_cast<const cx<T>*>(psf_hat);
    p.tw = tw.data();
    p.ny = ny; p.kx = kx; p.pitch = pitch; p.V = V; p.in_sb = in_sb; p.in_sv = in_sv;
    p.mode = mode;
    const int gy = (mode == COL_PER_IMAGE) ? frames * V : frames;   // as rlsted.cpp col_t()
    if (mode != COL_PER_IMAGE && !WavePrivate<Cfg>::value) return -3;
    p.images = gy; p.order = 1;
    run_grid((kx + C - 1) / C, gy, Cfg::T * C, (size_t)C * LdsSlots<Cfg>::value * sizeof(cx<T>),
             [&](int tid, int bx, int by, unsigned char* lds, EmuSync& s) {
                 if constexpr (L == 256) {
                     if (g_special && ny == 192 && V == 1 && mode == COL_PER_IMAGE && pitch % C == 0) {
                         colconv_wave_body<Cfg, C, COL_PER_IMAGE, T, false, 192, 1>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                         return;
                     }
                     if (g_special && ny == 128 && V == 1 && mode == COL_PER_IMAGE && pitch % C == 0) {   // (a multiple of 128 rows: the 16-byte tile I/O)
                         colconv_wave_body<Cfg, C, COL_PER_IMAGE, T, false, 128, 1>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                         return;
                     }
                 }
                 if constexpr (WavePrivate<Cfg>::value)
                     switch (mode) {   // same dispatch as launch_col_t in fft_kernels.hip
                         case COL_H_MULTI: colconv_wave_body<Cfg, C, COL_H_MULTI, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s); break;
                         case COL_HT_SUM: colconv_wave_body<Cfg, C, COL_HT_SUM, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s); break;
                         default: colconv_wave_body<Cfg, C, COL_PER_IMAGE, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                     }
                 else
                     colconv_body<Cfg, C, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
             });
    return 0;
}

template <int L, int MODE, typename T>
static int row_m(const RowParams<T>& p, int gy) {
    using CF = CfgFor<L>;
    using Cfg = typename CF::Cfg;
    constexpr int Q = sizeof(T) == 4 ? CF::Q32 : CF::Q64;
    const int pairs = (p.ny + 1) / 2;
    run_grid((pairs + Q - 1) / Q, gy, Cfg::T * Q, (size_t)Q * LdsSlots<Cfg>::value * sizeof(cx<T>),
             [&](int tid, int bx, int by, unsigned char* lds, EmuSync& s) {
                 constexpr bool MULTI = (MODE == ROW_UPDATE || MODE == ROW_ADJ);
                 if constexpr (WavePrivate<Cfg>::value && (MODE == ROW_RATIO || MODE == ROW_UPDATE)) {
                     if (MODE == ROW_RATIO || p.V == 1) {   // same dispatch as k_rowpass (LEAN) in fft_kernels.hip
                         rowlean_body<Cfg, Q, MODE, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                         return;
                     }
                 }
                 if constexpr (MODE == ROW_UPDATE) {
                     if (p.V > 1 && p.sub_one) {   // same dispatch as launch_row_m in fft_kernels.hip
                         rowpass_body<Cfg, Q, MODE, true, T, true>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                         return;
                     }
                 }
                 if (MULTI && p.V == 1)
                     rowpass_body<Cfg, Q, MODE, MULTI, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                 else
                     rowpass_body<Cfg, Q, MODE, false, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
             });
    return 0;
}

// RowParams::sub_one of every row launch (conv_kernels.hpp rl_ratio: ROW_RATIO stores rowFFT(ratio - 1), ROW_UPDATE
// multiplies by max(1 + acc / norm, 0))
static int g_sub_one = 0;

template <int L, typename T>
static int row_t(int mode, const T* spec_in, T* spec_out, const T* src, T* dst, const T* norm, const T* scale,
                 int ny, int nx, int pitch, int V, int gy) {
    auto tw = twiddles<L, T>();
    RowParams<T> p;
    p.sub_one = g_sub_one;
    p.spec_in = reinterpret_cast<const cx<T>*>(spec_in);
    p.spec_out = reinterpret_cast<cx<T>*>(spec_out);
    p.src = src; p.dst = dst; p.norm = norm; p.scale = scale; p.tw = tw.data();
    p.ny = ny; p.nx = nx; p.pitch = pitch; p.V = V;
    switch (mode) {
        case ROW_FWD: return row_m<L, ROW_FWD, T>(p, gy);
        case ROW_INV: return row_m<L, ROW_INV, T>(p, gy);
        case ROW_RATIO: return row_m<L, ROW_RATIO, T>(p, gy);
        case ROW_UPDATE: return row_m<L, ROW_UPDATE, T>(p, gy);
        case ROW_ADJ: return row_m<L, ROW_ADJ, T>(p, gy);
    }
    return -1;
}

#define DISPATCH_L(L, call)                         \
    switch (L) {                                    \
        case 64: { constexpr int LL = 64; return call; }     \
        case 192: { constexpr int LL = 192; return call; }   \
        case 256: { constexpr int LL = 256; return call; }   \
        case 576: { constexpr int LL = 576; return call; }   \
        default: return -2;                         \
    }

extern "C" {

void emu_set_sub_one(int on) { g_sub_one = on; }
void emu_set_special(int on) { g_special = on; }
void emu_set_park(int on);

// returns 1 if the column kernel of this length reads psf_hat transposed, 0 if not, <0 unknown L
int emu_geometry(int L, int* T, int* C, int* Q) {
#define GEO(LL) case LL: *T = CfgFor<LL>::Cfg::T; *C = CfgFor<LL>::C64; *Q = CfgFor<LL>::Q64; return WavePrivate<ColCfgFor<LL>::type>::value ? 1 : 0;
    switch (L) { GEO(64) GEO(192) GEO(256) GEO(576) GEO(1152) GEO(2304) GEO(4608) }
    return -2;
}

int emu_col_f64(int L, const double* in, double* out, const double* psf_hat, int ny, int kx, int pitch, int V,
                int frames, int in_sb, int in_sv, int mode) {
    DISPATCH_L(L, (col_t<LL, double>(in, out, psf_hat, ny, kx, pitch, V, frames, in_sb, in_sv, mode)))
}
int emu_col_f32(int L, const float* in, float* out, const float* psf_hat, int ny, int kx, int pitch, int V,
                int frames, int in_sb, int in_sv, int mode) {
    DISPATCH_L(L, (col_t<LL, float>(in, out, psf_hat, ny, kx, pitch, V, frames, in_sb, in_sv, mode)))
}
int emu_row_f64(int L, int mode, const double* spec_in, double* spec_out, const double* src, double* dst,
                const double* norm, const double* scale, int ny, int nx, int pitch, int V, int gy) {
    DISPATCH_L(L, (row_t<LL, double>(mode, spec_in, spec_out, src, dst, norm, scale, ny, nx, pitch, V, gy)))
}
int emu_row_f32(int L, int mode, const float* spec_in, float* spec_out, const float* src, float* dst,
                const float* norm, const float* scale, int ny, int nx, int pitch, int V, int gy) {
    DISPATCH_L(L, (row_t<LL, float>(mode, spec_in, spec_out, src, dst, norm, scale, ny, nx, pitch, V, gy)))
}

// rowpair_body (two frames in one complex image): spectra [pairs][ny][L]; frames = images covered by the launch
int emu_row_pair_f64(int L, int mode, const double* spec_in, double* spec_out, const double* src, double* dst, const double* norm,
                     int ny, int nx, int frames, int in_mod);
int emu_row_pair_f32(int L, int mode, const float* spec_in, float* spec_out, const float* src, float* dst, const float* norm,
                     int ny, int nx, int frames, int in_mod);

// colconv_outer_body (long column transforms on a wave-private core): L = M * Li, Li in {256, 576}.
// psf_hat: complex [V][kx][L] (transposed layout), or -- real_psf -- its real parts [V][kx][L].
}  // extern "C"
// ColParams-independent: waiting core results per lane kept in LDS by the whole pass (conv_kernels.hpp PARK); 0 = none
static int g_park = 0;
template <class Core, int M, typename T, int C = 8>
static int col_outer_t(const T* in, T* out, const T* psf_hat, int real_psf, int ny, int kx, int pitch, int V, int frames,
                       int in_sb, int in_sv, int mode = COL_PER_IMAGE) {
    constexpr int L = M * Core::L;
    constexpr int n_core = PassTw<Core, false, 0>::TOTAL;
    std::vector<double> h(2 * (size_t)(n_core + (M - 1) * Core::L));
    fill_pass_twiddles<Core>(h.data());
    for (int q = 1; q < M; ++q)
        for (int k = 0; k < Core::L; ++k) {
            const long double a = -6.283185307179586476925286766559005768L * (long double)q * (long double)k / (long double)L;
            h[2 * (size_t)(n_core + (q - 1) * Core::L + k)] = (double)cosl(a);
            h[2 * (size_t)(n_core + (q - 1) * Core::L + k) + 1] = (double)sinl(a);
        }
    std::vector<cx<T>> tw(h.size() / 2);
    for (size_t i = 0; i < tw.size(); ++i) tw[i] = mk<T>((T)h[2 * i], (T)h[2 * i + 1]);
    ColParams<T> p;
    p.in = reinterpret_cast<const cx<T>*>(in);
    p.out = reinterpret_cast<cx<T>*>(out);
    p.psf_hat = real_psf ? nullptr : reinterpret_cast<const cx<T>*>(psf_hat);
    p.psf_hat_re = real_psf ? psf_hat : nullptr;
    p.tw = tw.data();
    p.ny = ny; p.kx = kx; p.pitch = pitch; p.V = V; p.in_sb = in_sb; p.in_sv = in_sv;
    p.mode = mode; p.images = mode == COL_PER_IMAGE ? frames * V : frames; p.order = 1;
    // the device's choices: f32 3 of the 4 x 10 values (L = 2304), 14 of the 8 x 10 (L = 4608); float64 (g_park == 2) 10 and 24;
    // M = 2: 5 (for the test)
    auto parked = [&](auto park_c) {
        constexpr int PARK = decltype(park_c)::value;
        constexpr int TWLDS = (M == 8 || C == 16) ? 2 : 1;     // the twiddles from an LDS copy (L = 4608 and the 16-column tiles of 2304: the outer table too)
        run_grid((kx + C - 1) / C, p.images, 64 * C,
                 ((size_t)C * LdsSlots<Core>::value + (size_t)PARK * 64 * C + (TWLDS > 0 ? PassTw<Core, false, 0>::TOTAL : 0) + (TWLDS > 1 ? (M - 1) * Core::L : 0)) * sizeof(cx<T>),
                 [&](int tid, int bx, int by, unsigned char* lds, EmuSync& s) {
                     cx<T>* l = reinterpret_cast<cx<T>*>(lds);
                     if (real_psf) colconv_outer_body<Core, M, C, T, true, COL_PER_IMAGE, PARK, TWLDS>(p, tid, bx, by, l, s);
                     else colconv_outer_body<Core, M, C, T, false, COL_PER_IMAGE, PARK, TWLDS>(p, tid, bx, by, l, s);
                 });
    };
    if (g_park == 1) {
        parked(std::integral_constant<int, M == 4 ? 3 : M == 8 ? 14 : 5>{});
        return 0;
    }
    if (g_park == 2) {
        parked(std::integral_constant<int, M == 4 ? 10 : M == 8 ? 24 : 4>{});
        return 0;
    }
    run_grid((kx + C - 1) / C, p.images, 64 * C, (size_t)C * LdsSlots<Core>::value * sizeof(cx<T>),
             [&](int tid, int bx, int by, unsigned char* lds, EmuSync& s) {
                 cx<T>* l = reinterpret_cast<cx<T>*>(lds);
                 if constexpr (Core::L == 256) {   // emu_set_special: the row count at compile time (the device: M x 512 rows on the 576 core)
                     if (g_special && ny == M * 192 && pitch % C == 0) {
                         if (real_psf) colconv_outer_body<Core, M, C, T, true, COL_PER_IMAGE, 0, 0, M * 192>(p, tid, bx, by, l, s);
                         else colconv_outer_body<Core, M, C, T, false, COL_PER_IMAGE, 0, 0, M * 192>(p, tid, bx, by, l, s);
                         return;
                     }
                     if (g_special && ny == M * 128 && pitch % C == 0) {   // (classes of 128 rows: the 16-byte tile I/O)
                         if (real_psf) colconv_outer_body<Core, M, C, T, true, COL_PER_IMAGE, 0, 0, M * 128>(p, tid, bx, by, l, s);
                         else colconv_outer_body<Core, M, C, T, false, COL_PER_IMAGE, 0, 0, M * 128>(p, tid, bx, by, l, s);
                         return;
                     }
                 }
                 if (real_psf) colconv_outer_body<Core, M, C, T, true>(p, tid, bx, by, l, s);
                 else colconv_outer_body<Core, M, C, T, false>(p, tid, bx, by, l, s);
             });
    return 0;
}
template <class Core, int M, typename T>
static int col_outer_split_t(const T* in, T* out, const T* psf_hat, int real_psf, int ny, int kx, int pitch, int V, int frames,
                       int sum_views) {
    constexpr int C = 8, L = M * Core::L;
    constexpr int n_core = PassTw<Core, false, 0>::TOTAL;
    std::vector<double> h(2 * (size_t)(n_core + (M - 1) * Core::L));
    fill_pass_twiddles<Core>(h.data());
    for (int q = 1; q < M; ++q)
        for (int k = 0; k < Core::L; ++k) {
            const long double a = -6.283185307179586476925286766559005768L * (long double)q * (long double)k / (long double)L;
            h[2 * (size_t)(n_core + (q - 1) * Core::L + k)] = (double)cosl(a);
            h[2 * (size_t)(n_core + (q - 1) * Core::L + k) + 1] = (double)sinl(a);
        }
    std::vector<cx<T>> tw(h.size() / 2);
    for (size_t i = 0; i < tw.size(); ++i) tw[i] = mk<T>((T)h[2 * i], (T)h[2 * i + 1]);
    ColParams<T> p;
    p.in = reinterpret_cast<const cx<T>*>(in);
    p.out = reinterpret_cast<cx<T>*>(out);
    p.psf_hat = real_psf ? nullptr : reinterpret_cast<const cx<T>*>(psf_hat);
    p.psf_hat_re = real_psf ? psf_hat : nullptr;
    p.tw = tw.data();
    // The split pass as the plan runs it.  sum_views = 0 (H): COL_SPLIT_FWD over the `frames` input images, COL_SPLIT_INV per
    // (frame, view) -> frames * V output images.  sum_views = 1 (H_t): COL_SPLIT_FWD over the frames * V input images,
    // COL_SPLIT_INV_SUM per frame -> `frames` output images.  The slot-order spectra are poisoned with NaN first.
    const int n_in = sum_views ? frames * V : frames;
    const size_t xs_img = (size_t)((kx + C - 1) / C) * outer_slots_tile_elems<Core, M, C>();
    std::vector<cx<T>> xs((size_t)n_in * xs_img, mk<T>((T)NAN, (T)NAN));
    p.ny = ny; p.kx = kx; p.pitch = pitch; p.V = V; p.in_sb = 1; p.in_sv = 0; p.order = 1;
    p.xs_out = xs.data();
    p.xs_in = xs.data();
    constexpr int TWS = M == 8 ? 2 : M == 4 ? 1 : 0;      // the device's OuterCol<L>::TWLDS_SPLIT
    constexpr size_t lds_split = ((size_t)C * LdsSlots<Core>::value + (TWS > 0 ? PassTw<Core, false, 0>::TOTAL : 0) + (TWS > 1 ? (M - 1) * Core::L : 0)) * sizeof(cx<T>);
    p.mode = COL_SPLIT_FWD; p.images = n_in;
    run_grid((kx + C - 1) / C, p.images, 64 * C, lds_split,
             [&](int tid, int bx, int by, unsigned char* lds, EmuSync& s) {
                 colconv_outer_body<Core, M, C, T, false, COL_SPLIT_FWD, 0, TWS>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
             });
    p.mode = sum_views ? COL_SPLIT_INV_SUM : COL_SPLIT_INV; p.images = sum_views ? frames : frames * V;
    run_grid((kx + C - 1) / C, p.images, 64 * C, lds_split,
             [&](int tid, int bx, int by, unsigned char* lds, EmuSync& s) {
                 cx<T>* l = reinterpret_cast<cx<T>*>(lds);
                 if (sum_views) {
                     if (real_psf) colconv_outer_body<Core, M, C, T, true, COL_SPLIT_INV_SUM, 0, TWS>(p, tid, bx, by, l, s);
                     else colconv_outer_body<Core, M, C, T, false, COL_SPLIT_INV_SUM, 0, TWS>(p, tid, bx, by, l, s);
                 } else {
                     if (real_psf) colconv_outer_body<Core, M, C, T, true, COL_SPLIT_INV, 0, TWS>(p, tid, bx, by, l, s);
                     else colconv_outer_body<Core, M, C, T, false, COL_SPLIT_INV, 0, TWS>(p, tid, bx, by, l, s);
                 }
             });
    return 0;
}
template <int L, typename T>
static int row_pair_t(int mode, const T* spec_in, T* spec_out, const T* src, T* dst, const T* norm, int ny, int nx, int frames, int in_mod) {
    using CF = CfgFor<L>;
    using Cfg = typename CF::Cfg;
    if constexpr (WavePrivate<Cfg>::value) {
        constexpr int Q = sizeof(T) == 4 ? CF::Q32 : CF::Q64;
        auto tw = twiddles<L, T>();
        RowParams<T> p;
        p.spec_in = reinterpret_cast<const cx<T>*>(spec_in);
        p.spec_out = reinterpret_cast<cx<T>*>(spec_out);
        p.src = src; p.dst = dst; p.norm = norm; p.scale = nullptr; p.tw = tw.data();
        p.ny = ny; p.nx = nx; p.pitch = L; p.V = 1; p.frames = frames; p.in_mod = in_mod;
        p.sub_one = g_sub_one;
        auto run = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            run_grid((ny + Q - 1) / Q, (frames + 1) / 2, 64 * Q, (size_t)Q * LdsSlots<Cfg>::value * sizeof(cx<T>),
                     [&](int tid, int bx, int by, unsigned char* lds, EmuSync& s) {
                         if constexpr (L == 256 && MODE != ROW_FWD) {
                             if (g_special && nx == 192 && p.sub_one) {
                                 rowpair_body<Cfg, Q, MODE, T, 192, 1>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                                 return;
                             }
                         }
                         rowpair_body<Cfg, Q, MODE, T>(p, tid, bx, by, reinterpret_cast<cx<T>*>(lds), s);
                     });
        };
        if (mode == ROW_FWD) run(std::integral_constant<int, ROW_FWD>{});
        else if (mode == ROW_RATIO) run(std::integral_constant<int, ROW_RATIO>{});
        else if (mode == ROW_UPDATE) run(std::integral_constant<int, ROW_UPDATE>{});
        else return -1;
        return 0;
    } else {
        return -3;
    }
}
extern "C" {
int emu_row_pair_f64(int L, int mode, const double* spec_in, double* spec_out, const double* src, double* dst, const double* norm,
                     int ny, int nx, int frames, int in_mod) {
    DISPATCH_L(L, (row_pair_t<LL, double>(mode, spec_in, spec_out, src, dst, norm, ny, nx, frames, in_mod)))
}
int emu_row_pair_f32(int L, int mode, const float* spec_in, float* spec_out, const float* src, float* dst, const float* norm,
                     int ny, int nx, int frames, int in_mod) {
    DISPATCH_L(L, (row_pair_t<LL, float>(mode, spec_in, spec_out, src, dst, norm, ny, nx, frames, in_mod)))
}
void emu_set_park(int on) { g_park = on; }
int emu_col_outer_f64(int Li, int M, const double* in, double* out, const double* psf_hat, int real_psf, int ny, int kx,
                      int pitch, int V, int frames, int in_sb, int in_sv) {
    using C256 = CfgFor<256>::Cfg;
    using C576 = CfgFor<576>::Cfg;
    if (Li == 256 && M == 4) return col_outer_t<C256, 4, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, in_sb, in_sv);
    if (Li == 256 && M == 16) return col_outer_t<C256, 4, double, 16>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, in_sb, in_sv);   // M = 4 on 16-column tiles
    if (Li == 256 && M == 2) return col_outer_t<C256, 2, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, in_sb, in_sv);
    if (Li == 576 && M == 4) return col_outer_t<C576, 4, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, in_sb, in_sv);
    if (Li == 256 && M == 8) return col_outer_t<C256, 8, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, in_sb, in_sv);
    if (Li == 576 && M == 8) return col_outer_t<C576, 8, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, in_sb, in_sv);
    return -2;
}
// the split column pass (COL_SPLIT_FWD then COL_SPLIT_INV / COL_SPLIT_INV_SUM); sum_views: 0 = H (in [frames], out [frames * V]),
// 1 = H_t with the views summed (in [frames * V], out [frames])
int emu_col_outer_split_f64(int Li, int M, const double* in, double* out, const double* psf_hat, int real_psf, int ny, int kx,
                            int pitch, int V, int frames, int sum_views) {
    using C256 = CfgFor<256>::Cfg;
    using C576 = CfgFor<576>::Cfg;
    if (Li == 256 && M == 4) return col_outer_split_t<C256, 4, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, sum_views);
    if (Li == 256 && M == 2) return col_outer_split_t<C256, 2, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, sum_views);
    if (Li == 576 && M == 4) return col_outer_split_t<C576, 4, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, sum_views);
    if (Li == 256 && M == 8) return col_outer_split_t<C256, 8, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, sum_views);
    if (Li == 576 && M == 8) return col_outer_split_t<C576, 8, double>(in, out, psf_hat, real_psf, ny, kx, pitch, V, frames, sum_views);
    return -2;
}

// host build of the device Poisson sampler (philox_poisson.hpp)
int emu_poisson(const double* lam, int n, unsigned long long seed, unsigned image, double* out) {
    for (int i = 0; i < n; ++i) out[i] = philox_poisson(lam[i], seed, image, (unsigned)i);
    return 0;
}
// first-attempt-only variant used by the fast Poisson kernel: flags[i] = accepted
int emu_poisson_fast(const double* lam, int n, unsigned long long seed, unsigned image, double* out, int* flags) {
    for (int i = 0; i < n; ++i) flags[i] = philox_poisson_fast(lam[i], seed, image, (unsigned)i, &out[i]) ? 1 : 0;
    return 0;
}
void emu_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
    Philox4 o = philox4x32_10(c0, c1, c2, c3, k0, k1);
    for (int i = 0; i < 4; ++i) out[i] = o.x[i];
}
}
