// conv_kernels.hpp -- workgroup bodies of the FFT-convolution / Richardson-Lucy
// kernels.  Shared by the HIP kernels (rlsted_kernels.hip) and by the host
// emulator used in the CPU tests (tests/emu/emu.cpp).
//
// Reference semantics implemented here (figure_generation/line_sted_tools.py):
//   H   :567-577  per PSF: zero padded 'same' convolution, clamp negatives to 0
//   H_t :579-594  sum over PSFs of the same convolution (un-flipped PSF), each
//                 term clamped, then divided by H_t(ones)
//   iterate :520-531  est *= H_t(measurement / H(est))
//
// Data layout (all row-major, contiguous):
//   images   real  [img][ny][nx]
//   spectra  cx<T> [img][ny][pitch]   row-transformed half spectra, Kx = Lx/2+1
//                                     valid columns, pitch = Kx rounded up to 8
//   psf_hat  cx<T> [view][Ly][pitch]  2-D spectrum of the PSF wrapped around the
//                                     origin, pre-scaled by 1/(Ly*Lx); stored
//                                     transposed [view][Kx][Ly] for wave-private
//                                     column transforms (Ly config with T == 64)
// A 2-D circular convolution of size Ly x Lx with Ly >= ny + Py/2, Lx >= nx +
// Px/2 restricted to rows < ny, columns < nx equals the zero padded 'same'
// convolution exactly (DESIGN.md "wrap-free sizes").
#pragma once
#include "fft_core.hpp"

#ifndef RL_TILE_WIDE
#define RL_TILE_WIDE 1        // 16-byte accesses in the column tile I/O of the compile-time-size kernels (colconv_wave_body WIDE)
#endif
#ifndef RL_TILE_WIDE_F64
#define RL_TILE_WIDE_F64 0   // the two-column tile I/O in double: the host emulator only (its tests run the tile code in double)
#endif
#ifndef RL_CT_RESIDUAL
#define RL_CT_RESIDUAL 1      // compact twiddles in the transforms that carry `ratio - 1` (kernels with `sub_one` at compile time)
#endif

namespace rl {

// a / b.  float: hardware reciprocal (v_rcp_f32, <= 1 ulp) times a -- the f32 plans
// are specified to 1e-5, an IEEE-exact quotient costs ~10 instructions per element
// (measured in round 2: it moves the f32 error by < 1 % and costs 1.6 % frames/s);
// double: exact division.  On the host (emulator) both are plain divisions.
RL_HD float rl_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
RL_HD double rl_div(double a, double b) { return a / b; }

// ---------------------------------------------------------------------------
// "Ratio minus one" (RowParams::sub_one, f32 plans with non-negative PSFs).  The 'same' convolution is linear and
// zero padded, so  conv(ratio, p) = conv(ones, p) + conv(ratio - 1, p)  exactly, and conv(ones, p) summed over the
// views IS the normaliser H_t(ones) (ref:589-592; non-negative PSFs: its clamp never acts).  The second half of an
// iteration then transforms `ratio - 1` -- a residual of the size of the shot noise once the estimate explains the
// data -- instead of a ratio of order one, and
//     estimate *= max(1 + sum_v conv(ratio_v - 1, p_v) / norm, 0)        ( = sum_v conv(ratio_v, p_v) / norm, ref:527-530 )
// The transforms' f32 rounding error is proportional to what they carry: the white ~2e-7 per iteration that H_t's
// transforms stamped onto the update factor (it is not blurred by any later convolution, so every iteration adds to
// it) shrinks with the residual.  Same arithmetic count, identical in exact arithmetic; the clamp (ref:587) acts on
// the view sum, as in the fused-views mode.
// v: the model's prediction H(est) as it comes out of the inverse transform; the reference clamps it at 0 (ref:575) and divides
// (ref:524).  In exact arithmetic it is positive wherever the estimate and the PSFs are; a transform resolves it to eps * max
// only, so in a dark region wider than the PSF -- estimate ~ 1e-9 after the first iteration, ref:510 -- an f32 plan's
// prediction is rounding noise of either sign (a float64 plan's: from 1e-16 of the maximum down).  The reference then divides by
// zero: inf, NaN through the next fftconvolve, the whole frame lost.  Here a pixel whose prediction is not positive is NEUTRAL --
// ratio 1, residual 0: it neither raises nor lowers the estimate -- and every value stays finite
// (tests/test_gpu_parity.py::test_dark_background_narrow_psf_stays_finite).  Where v > 0 the clamp is the identity.
// `unresolved` collects, per lane, whether any pixel INSIDE the image met such a prediction: the row bodies add the lanes up in
// RowParams::unresolved (rl_count_unresolved), the plan reports the count (rl_deconv_unresolved) -- zero on data whose predictions
// the plan's arithmetic resolves; an f32 plan that counts should be a float64 plan.
template <typename T>
RL_HD T rl_ratio(T meas, T v, bool sub_one, bool in_image, bool& unresolved) {
    const T q = rl_div(sub_one ? meas - v : meas, v);
    const bool pos = v > (T)0;
    unresolved = unresolved || (in_image && !pos);
    return pos ? q : (sub_one ? (T)0 : (T)1);
}
RL_HD void rl_count_unresolved(unsigned long long* counter, bool lane_met_one) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one atomic per wavefront that met any (a dark frame has them in half its lanes: 64 atomics on one address per wave otherwise)
    const unsigned long long met = __ballot(lane_met_one);
    if (counter != nullptr && met != 0 && (int)(threadIdx.x & 63u) == __ffsll((long long)met) - 1)
        atomicAdd(counter, (unsigned long long)__popcll(met));
#else
    if (counter != nullptr && lane_met_one) __atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED);
#endif
}
// a: sum over the views of the back-transformed values -- clamped per view (plain mode) or raw (sub_one)
template <typename T>
RL_HD T rl_update_factor(T a, T nrm, bool sub_one) {
    const T f = rl_div(a, nrm);
    if (!sub_one) return f;
    const T g = (T)1 + f;
    return g > (T)0 ? g : (T)0;
}
template <typename T>
RL_HD T rl_clamp0(T x) { return x > (T)0 ? x : (T)0; }

// ---------------------------------------------------------------------------
// Storage-precision STUDY builds (BASELINE config 5: "fp32 vs fp16 convolve, tolerance study"):
// RL_SPEC_QUANT = 1 rounds every spectrum value to IEEE half precision on its way to global memory (scaled
// by the launch's power of two `qscale`, which brings the spectrum's DC term -- the bound of every
// coefficient of a non-negative image -- to 2^14), RL_SPEC_QUANT = 2 to bfloat16 (f32's range: no scale).
// The values are still stored as f32: the build measures what 16-bit spectrum storage between the row
// and the column kernels would do to the results, not its bandwidth.  0 (every product build): identity.
#ifndef RL_SPEC_QUANT
#define RL_SPEC_QUANT 0
#endif
template <typename T>
RL_HD cx<T> rl_spec_round(cx<T> v, float qscale) {
#if RL_SPEC_QUANT == 1 && defined(__HIP_DEVICE_COMPILE__)
    const float inv = 1.0f / qscale;   // qscale is a power of two: exact
    return mk<T>((T)((float)(_Float16)((float)v.re * qscale) * inv), (T)((float)(_Float16)((float)v.im * qscale) * inv));
#elif RL_SPEC_QUANT == 2 && defined(__HIP_DEVICE_COMPILE__)
    (void)qscale;
    return mk<T>((T)(float)(__bf16)(float)v.re, (T)(float)(__bf16)(float)v.im);
#else
    (void)qscale;
    return v;
#endif
}

// ---------------------------------------------------------------------------
// Layout of a row-transformed spectrum image in global memory: ny rows x pitch columns of complex T, row-major
// (pitch a multiple of 8).  (A blocked layout -- row pairs x 8-column blocks, one 128-byte line per tile and row
// pair -- was built and measured in round 1: the column kernel's fabric reads did not change, 1.1 % slower; removed.)
RL_HD size_t spec_image_elems(int ny, int pitch) { return (size_t)ny * pitch; }
RL_HD size_t spec_off(int row, int col, int pitch) { return (size_t)row * pitch + col; }

// ------------------------------ column pass --------------------------------
// For one tile of C spectrum columns: forward FFT along y (rows >= ny are
// zero), multiply by psf_hat, inverse FFT along y, keep rows < ny.
template <typename T>
struct ColParams {
    const cx<T>* in;        // [n_in_img][ny][pitch]
    cx<T>* out;             // [gridDim.y][ny][pitch]
    const cx<T>* psf_hat;   // [V][Ly][pitch]
    const cx<T>* tw;        // [Ly] exp(-2 pi i m / Ly)
    int ny, kx, pitch;
    int V;                  // views per frame; blockIdx.y = frame*V + view
    int in_sb, in_sv;       // COL_PER_IMAGE: input image index = frame*in_sb + view*in_sv
    int mode;               // ColMode (wave-private column kernel only; others: per image)
    int images;             // streaming kernel: output images covered by the launch (grid.y of the tiled kernel)
    int order;              // tiled kernel's work order: images per block of the tile order (1 = image-major)
    // A point-symmetric PSF has a real spectrum: its real parts alone, same indexing as the transposed
    // psf_hat ([view][kx][L]); used by the REALP instantiations of the wave-private column kernel
    // (half the multiplier bytes, a real x complex product).  nullptr: the complex multiplier.
    const T* psf_hat_re = nullptr;
    // the split column pass: column spectra in register-slot order (colconv_outer_body: outer_slots_image_elems per image)
    const cx<T>* xs_in = nullptr;   // COL_SPLIT_INV: image frame*in_sb + view*in_sv; COL_SPLIT_INV_SUM: images frame*V + v
    cx<T>* xs_out = nullptr;        // COL_SPLIT_FWD: one per launch row
    float qscale = 1.0f;    // storage-precision study builds only (rl_spec_round)
    int residual = 0;       // the input is the spectrum of `ratio - 1` (H_t of a sub_one plan): compact twiddles may serve it
};

template <class Cfg, int C, typename T, class Sync>
RL_HD void colconv_body(const ColParams<T>& p, int tid, int bx, int by, cx<T>* lds, Sync& sync_in) {
    // The threads of one transform are tid = c + C*t here (consecutive threads = consecutive
    // columns, for coalescing): they are spread over the waves of the workgroup whatever T is,
    // so the inter-pass exchanges always need workgroup barriers.
    struct WorkgroupSync {
        Sync& s;
        RL_HD void wg() const { s.wg(); }
        RL_HD void wave() const { s.wg(); }
    } sync{sync_in};
    constexpr int NP = Cfg::NP;
    constexpr int VMAX = CfgRegs<Cfg>::VMAX;
    const int c = tid % C, t = tid / C;
    const int col = bx * C + c;
    const bool colok = col < p.kx;
    const int frame = by / p.V, view = by % p.V;
    const size_t img = spec_image_elems(p.ny, p.pitch);
    const cx<T>* __restrict__ in = p.in + (size_t)(frame * p.in_sb + view * p.in_sv) * img;
    cx<T>* __restrict__ out = p.out + (size_t)by * img;
    const cx<T>* __restrict__ ph = p.psf_hat + (size_t)view * Cfg::L * p.pitch;
    LdsView<T, C, LdsGather<Cfg::L>::value, LdsPadShift<Cfg::L>::value> view_lds{lds + c};

    cx<T> v[VMAX];
    cx<T> tl = mk<T>((T)0, (T)0);   // tail element (wave-private L = 576 only; unused here)
    {   // forward pass 0 operands straight from global memory
        using F0 = PassInfo<Cfg, false, 0>;
#pragma unroll
        for (int nb = 0; nb < F0::NB; ++nb) {
            const int j = t + nb * Cfg::T;
#pragma unroll
            for (int r = 0; r < F0::R; ++r) {
                const int i = j + r * F0::NBF;
                cx<T> x = mk<T>((T)0, (T)0);
                if (j < F0::NBF && colok && i < p.ny) x = in[spec_off(i, col, p.pitch)];
                v[nb * F0::R + r] = x;
            }
        }
    }
    run_passes<Cfg, false, 0, true>(v, tl, t, view_lds, p.tw, sync);
    {   // pointwise multiply in registers (element index of the last forward pass)
        using FL = PassInfo<Cfg, false, NP - 1>;
#pragma unroll
        for (int nb = 0; nb < FL::NB; ++nb) {
            const int j = t + nb * Cfg::T;
#pragma unroll
            for (int r = 0; r < FL::R; ++r) {
                const int i = j + r * FL::NBF;
                if (j < FL::NBF && colok) v[nb * FL::R + r] = cmul(v[nb * FL::R + r], ph[(size_t)i * p.pitch + col]);
            }
        }
    }
    run_passes<Cfg, true, 0, true>(v, tl, t, view_lds, p.tw, sync);
    {
        using IL = PassInfo<Cfg, true, NP - 1>;
#pragma unroll
        for (int nb = 0; nb < IL::NB; ++nb) {
            const int j = t + nb * Cfg::T;
#pragma unroll
            for (int r = 0; r < IL::R; ++r) {
                const int i = j + r * IL::NBF;
                if (j < IL::NBF && colok && i < p.ny) out[spec_off(i, col, p.pitch)] = rl_spec_round(v[nb * IL::R + r], p.qscale);
            }
        }
    }
}


// Column pass for wave-private transforms (Cfg::T == 64): workgroup = C waves,
// wave w owns spectrum column col0 + w.
//   1. the whole workgroup loads the [ny][C] tile with coalesced C*8-byte row
//      segments into LDS, column major (one padded transform per column)
//   2. each wave: FFT_y -> * psf_hat -> IFFT_y on its own column, in LDS and
//      registers, no workgroup barrier; result back to its LDS column
//   3. the whole workgroup stores rows < ny, coalesced as in 1.
// psf_hat is read transposed here: psf_hat_t[view][kx][Ly] (contiguous along ky).
//
// p.mode selects how views are walked (blockIdx.y = `by`):
//   COL_PER_IMAGE  by = frame*V + view: one input image, one output image (H_t per view, V == 1)
//   COL_H_MULTI    by = frame: ONE forward transform of the frame's spectrum feeds V
//                  multiply + inverse transforms -> output images frame*V + view   (H, ref:573-576)
//   COL_HT_SUM     by = frame: the V products FFT_y(in[frame*V+view]) * psf_hat[view] are summed
//                  in the Fourier domain and inverse transformed once -> output image frame
//                  (H_t, ref:584-588, with the per-view clamp replaced by one clamp of the sum:
//                  identical in exact arithmetic, see DESIGN.md "fused views")
enum ColMode { COL_PER_IMAGE = 0, COL_H_MULTI = 1, COL_HT_SUM = 2,
               // the split column pass of colconv_outer_body (round 3): forward half -> column spectra in register-slot order,
               // inverse half from there -- one image per launch row, or the V views of a frame summed
               COL_SPLIT_FWD = 3, COL_SPLIT_INV = 4, COL_SPLIT_INV_SUM = 5 };

// MODE is a compile-time parameter: each mode is its own kernel, so the single-view path
// does not inherit the register footprint of the multi-view loops.
// (A twiddle copy in LDS like colconv_outer_body's was tried for the fused multi-view modes -- two workgroups per CU, room for it:
// 512^2 x 4 views 5941 -> 5973 frames/s, noise.  These kernels wait for the vector ALU, not for L1.)
// NYC > 0 (round 4; COL_PER_IMAGE launches of single-view plans, and the fused multi-view modes): the image has exactly NYC rows, a multiple of 64 -- which rows of the
// tile exist is then known at compile time (no row compares, no exec branches around the loads and stores; the rows that do
// not exist are written to LDS as constants), and the pad columns kx .. pitch - 1 of the last tile are loaded and stored like
// the others (pitch is a multiple of C: in bounds; nobody reads them as data).
// CT = 1: compact twiddles in both transforms (the H_t launches of `ratio - 1` plans: fft_core.hpp pass_compute).
template <class Cfg, int C, int MODE, typename T, bool REALP = false, int NYC = 0, int CT = 0, class Sync>
RL_HD void colconv_wave_body(const ColParams<T>& p, int tid, int bx, int by, cx<T>* lds, Sync& sync) {
    static_assert(Cfg::T == 64, "wave-private body needs one wave per transform");
    static_assert(NYC == 0 || (NYC % 64 == 0 && NYC <= Cfg::L), "compile-time row count");
    constexpr int NP = Cfg::NP, L = Cfg::L, LP = LdsSlots<Cfg>::value;
    constexpr int VMAX = CfgRegs<Cfg>::VMAX;
    constexpr int NT = 64 * C;
    static_assert((L * C) % NT == 0, "tile must divide evenly over the workgroup");
    constexpr int NLD = (L * C) / NT;
    using FL = PassInfo<Cfg, false, NP - 1>;
    using IL = PassInfo<Cfg, true, NP - 1>;
    static_assert(!IL::TAIL, "the inverse must end on a lane-local pass");
    const int w = rl_uniform(tid / 64), lane = tid % 64;
    const int col0 = bx * C, col = col0 + w;
    const bool colok = col < p.kx;
    const int ny = NYC > 0 ? NYC : p.ny;
    const size_t img = spec_image_elems(ny, p.pitch);
    LdsView<T, 1, LdsGather<L>::value> view_lds{lds + w * LP};

    // Tile element e = tid + it*NT  <->  (row = e / C = r_lo + 64 it, column c = e % C): one 32-bit byte offset per thread,
    // the rows of step `it` behind a uniform base -- `scalar base + lane offset` loads and stores, no 64-bit lane arithmetic.
    const unsigned c_ = (unsigned)tid % C, r_lo = (unsigned)tid / C;
    const bool cok = NYC > 0 ? true : (int)(col0 + c_) < p.kx;
    const unsigned boff = (r_lo * (unsigned)p.pitch + (unsigned)col0 + c_) * (unsigned)sizeof(cx<T>);
    const size_t step = (size_t)64 * p.pitch * sizeof(cx<T>);
    // WIDE (round 4; the compile-time-size kernels): a thread moves TWO neighbouring columns of a row with one 16-byte access (f64:
    // two) -- thread -> (row tid / (C/2) + 128 it, columns 2 (tid % (C/2)), + 1) -- half the vector-memory instructions of the
    // tile load and store for the same bytes (16 bytes per lane is the width the memory pipeline is built for).
    constexpr bool WIDE = RL_TILE_WIDE != 0 && (sizeof(T) == 4 || RL_TILE_WIDE_F64 != 0) && NYC > 0 && NYC % 128 == 0 && C % 2 == 0;
    struct alignas(2 * sizeof(cx<T>)) cx2 {
        cx<T> a, b;
    };
    const unsigned w_cp = (unsigned)tid % (C / 2), w_r = (unsigned)tid / (C / 2);      // (64 C threads: 128 rows per step)
    const unsigned w_off = (w_r * (unsigned)p.pitch + (unsigned)col0 + 2 * w_cp) * (unsigned)sizeof(cx<T>);
    const size_t w_step = (size_t)128 * p.pitch * sizeof(cx<T>);
    auto load_tile = [&](const cx<T>* __restrict__ in) {
        if constexpr (WIDE) {
            constexpr int NW = NYC / 128;
            cx2 y[NW];
#pragma unroll
            for (int it = 0; it < NW; ++it) y[it] = *reinterpret_cast<const cx2*>(reinterpret_cast<const char*>(in) + it * w_step + w_off);
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                lds[(2 * w_cp) * LP + view_lds.nat(w_r + 128 * it)] = y[it].a;
                lds[(2 * w_cp + 1) * LP + view_lds.nat(w_r + 128 * it)] = y[it].b;
            }
            // rows NYC .. L - 1 of the tile are zero
#pragma unroll
            for (int e = tid; e < (L - NYC) * C; e += NT) lds[(e % C) * LP + view_lds.nat(NYC + e / C)] = mk<T>((T)0, (T)0);
            return;
        }
        cx<T> x[NLD];   // all global loads are issued before the first LDS write
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const char* sb = reinterpret_cast<const char*>(in) + it * step;
            x[it] = mk<T>((T)0, (T)0);
            if (cok && (int)(r_lo + 64 * it) < ny) x[it] = rl_ldg(sync, reinterpret_cast<const cx<T>*>(sb + boff));
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) lds[c_ * LP + view_lds.nat(r_lo + 64 * it)] = x[it];
    };
    auto store_tile = [&](cx<T>* __restrict__ out) {
        if constexpr (WIDE) {
#pragma unroll
            for (int it = 0; it < NYC / 128; ++it) {
                cx2 y;
                y.a = rl_spec_round(lds[(2 * w_cp) * LP + view_lds.nat(w_r + 128 * it)], p.qscale);
                y.b = rl_spec_round(lds[(2 * w_cp + 1) * LP + view_lds.nat(w_r + 128 * it)], p.qscale);
                *reinterpret_cast<cx2*>(reinterpret_cast<char*>(out) + it * w_step + w_off) = y;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            char* sb = reinterpret_cast<char*>(out) + it * step;
            if (cok && (int)(r_lo + 64 * it) < ny)
                *reinterpret_cast<cx<T>*>(sb + boff) = rl_spec_round(lds[c_ * LP + view_lds.nat(r_lo + 64 * it)], p.qscale);
        }
    };
    // v, tl *= psf_hat[view] column (register layout of the last forward pass)
    auto multiply = [&](cx<T>* v, cx<T>& tl, int view) {
        if constexpr (REALP) {   // real spectrum of a point-symmetric PSF
            const T* __restrict__ ph = p.psf_hat_re + ((size_t)view * p.kx + col) * L;
#pragma unroll
            for (int nb = 0; nb < FL::NBM; ++nb) {
                const int j = lane + nb * 64;
                if (j < FL::NBF) {
#pragma unroll
                    for (int r = 0; r < FL::R; ++r) v[nb * FL::R + r] = scale(v[nb * FL::R + r], ph[j + r * FL::NBF]);
                }
            }
            if constexpr (FL::TAIL) tl = scale(tl, ph[(64 + (lane & 7)) + bitrev3(lane >> 3) * FL::NBF]);
        } else {
            const cx<T>* __restrict__ ph = p.psf_hat + ((size_t)view * p.kx + col) * L;
#pragma unroll
            for (int nb = 0; nb < FL::NBM; ++nb) {
                const int j = lane + nb * 64;
                if (j < FL::NBF) {
#pragma unroll
                    for (int r = 0; r < FL::R; ++r) v[nb * FL::R + r] = cmul(v[nb * FL::R + r], ph[j + r * FL::NBF]);
                }
            }
            if constexpr (FL::TAIL)   // the tail value is output bitrev3(p) of butterfly 64 + jj
                tl = cmul(tl, ph[(64 + (lane & 7)) + bitrev3(lane >> 3) * FL::NBF]);
        }
    };
    // inverse transform of (v, tl), result in natural order into this wave's LDS column
    auto inverse_to_lds = [&](cx<T>* v, cx<T>& tl, const cx<T>* tw) {
        run_passes<Cfg, true, 0, true, CT>(v, tl, lane, view_lds, tw, sync);
        sync.wave();   // last pass' LDS reads are done before the column is overwritten
#pragma unroll
        for (int nb = 0; nb < IL::NB; ++nb) {
            const int j = lane + nb * 64;
            if (j < IL::NBF) {
#pragma unroll
                for (int r = 0; r < IL::R; ++r) view_lds.template at_step<IL::NBF>(j, view_lds.nat(j), r) = v[nb * IL::R + r];
            }
        }
    };

    if constexpr (MODE == COL_PER_IMAGE) {
        const int frame = NYC > 0 ? by : by / p.V, view = NYC > 0 ? 0 : by % p.V;   // (NYC: single-view launches)
        rl_stamp(sync, 0);
        load_tile(p.in + (size_t)(frame * p.in_sb + view * p.in_sv) * img);
        rl_stamp(sync, 1);
        sync.wg();
        rl_stamp(sync, 2);
        if (colok) {
            cx<T> v[VMAX];
            cx<T> tl = mk<T>((T)0, (T)0);
            run_passes<Cfg, false, 0, false, CT>(v, tl, lane, view_lds, p.tw, sync);
            rl_stamp(sync, 3);
            multiply(v, tl, view);
            rl_stamp(sync, 4);
            inverse_to_lds(v, tl, p.tw);
        }
        rl_stamp(sync, 5);
        sync.wg();
        rl_stamp(sync, 6);
        store_tile(p.out + (size_t)by * img);
        rl_stamp(sync, 7);
    } else if constexpr (MODE == COL_H_MULTI) {
        load_tile(p.in + (size_t)by * img);
        sync.wg();
        cx<T> f[VMAX];
        cx<T> ftl = mk<T>((T)0, (T)0);
        if (colok) run_passes<Cfg, false, 0, false>(f, ftl, lane, view_lds, p.tw, sync);
        for (int view = 0; view < p.V; ++view) {
            // keep the compiler from hoisting the (view-invariant) twiddle loads out of
            // this loop: ~60 registers that would cost a workgroup of occupancy
            const cx<T>* tw = p.tw;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+s"(tw));
#endif
            if (colok) {
                cx<T> v[VMAX];
#pragma unroll
                for (int i = 0; i < VMAX; ++i) v[i] = f[i];
                cx<T> tl = ftl;
                multiply(v, tl, view);
                inverse_to_lds(v, tl, tw);
            }
            sync.wg();
            store_tile(p.out + ((size_t)by * p.V + view) * img);
            if (view + 1 < p.V) sync.wg();   // the tile is read out before the next inverse scatters into it
        }
    } else {   // COL_HT_SUM
        cx<T> acc[VMAX];
        cx<T> atl = mk<T>((T)0, (T)0);
#pragma unroll
        for (int i = 0; i < VMAX; ++i) acc[i] = mk<T>((T)0, (T)0);
        for (int view = 0; view < p.V; ++view) {
            if (view > 0) sync.wg();         // every wave is done with the previous tile
            load_tile(p.in + ((size_t)by * p.V + view) * img);
            sync.wg();
            const cx<T>* tw = p.tw;   // not hoisted out of the view loop (see COL_H_MULTI)
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+s"(tw));
#endif
            if (colok) {
                cx<T> v[VMAX];
                cx<T> tl = mk<T>((T)0, (T)0);
                run_passes<Cfg, false, 0, false, CT>(v, tl, lane, view_lds, tw, sync);
                multiply(v, tl, view);
#pragma unroll
                for (int i = 0; i < FL::NBM * FL::R; ++i) acc[i] = acc[i] + v[i];
                atl = atl + tl;
            }
        }
        if (colok) {
            sync.wave();
            inverse_to_lds(acc, atl, p.tw);
        }
        sync.wg();
        store_tile(p.out + (size_t)by * img);
    }
}

// ---------------------------------------------------------------------------------------------
// Column pass for a long transform L = M * Li on the wave-private core of length Li (M = 2 or 4):
// one outer decimation-in-time step in registers around M core transforms per column.
//     x_q[m] = x[M m + q]                      (q = 0 .. M-1: the rows of residue q)
//     Y_q    = FFT_Li(x_q)                     (the core, one wave per column, exchanges through LDS)
//     X[k + Li j] = sum_q W_M^(q j) * (W_L^(q k) * Y_q[k])         (radix-M butterfly per register slot)
//     X *= psf_hat
//     Z_q[k] = conj(W_L^(q k)) * sum_j conj(W_M)^(q j) * X[k + Li j];   x_q = IFFT_Li(Z_q)
// Why: a workgroup that holds whole columns of L = 2304 in LDS fits 6 columns (124 KB, one workgroup
// per CU, 48-byte row segments, barriers between the passes): 0.44 TB/s.  Here LDS holds one residue
// class at a time -- 8 columns x 576 rows, the L = 576 kernel's 51 KB -- the M core results wait in
// registers (M * 9 complex values per lane), and the core is the barrier-free wave-private transform.
// Rows >= ny are zero on the way in and never stored (2048 of 2304: every residue class is the
// 512-of-576 case of the L = 576 kernel).  MODE as in colconv_wave_body (COL_PER_IMAGE / COL_H_MULTI / COL_HT_SUM).
// (Tried and measured slower, 907 vs 569 us per 16-frame launch at one workgroup per CU: decimation in frequency
// over M blocks of contiguous rows with the core transforms chained forward -> multiply -> inverse in registers.
// It loads all M blocks before the first transform and keeps all M x 9 values live through every core transform;
// the residue-class form below interleaves loads and transforms and lets the live set grow with them.)
// Twiddles: p.tw = [core table PassTw<Cfg>][ (M-1) x Li entries W_L^(q k), q = 1 .. M-1 ].
// PARK (whole pass only): that many of the waiting core results per lane -- classes 0, 1, ... on the way in, M-1, M-2, ... on the
// way back -- wait in LDS instead of registers ([value][thread] behind the transform regions: private to the thread, no barrier).
// The whole pass needs 4 x 10 (8 x 10) values per lane beside a core transform's working set and spills 25-31 dwords per lane
// at the register budget its residency allows: scratch that streams through HBM (7.4 MB each way per 2048^2 image).
// NYC > 0 (round 4): the image has exactly NYC rows, a multiple of 64 M -- every residue class then has NYC / M rows, a multiple
// of 64, and which rows of a class's tile exist is known at compile time (as colconv_wave_body's NYC: no row compares, no exec
// branches around the loads and stores; pad columns of the last tile are loaded and stored like the others -- pitch a multiple of C).
template <class Cfg, int M, int C, typename T, bool REALP = false, int MODE = COL_PER_IMAGE, int PARK = 0, int TWLDS = 0, int NYC = 0, class Sync>
RL_HD void colconv_outer_body(const ColParams<T>& p, int tid, int bx, int by, cx<T>* lds, Sync& sync) {
    static_assert(Cfg::T == 64, "the core must be a wave-private transform");
    static_assert(M == 2 || M == 4 || M == 8, "outer radix 2, 4 or 8");
    static_assert(NYC == 0 || (NYC % (64 * M) == 0 && NYC / M <= Cfg::L), "compile-time row count: whole 64-row steps per residue class");
    constexpr int NP = Cfg::NP, Li = Cfg::L, L = M * Li, LP = LdsSlots<Cfg>::value;
    constexpr int NT = 64 * C;
    static_assert((Li * C) % NT == 0, "tile must divide evenly over the workgroup");
    constexpr int NLD = (Li * C) / NT;
    using FL = PassInfo<Cfg, false, NP - 1>;
    using IL = PassInfo<Cfg, true, NP - 1>;
    static_assert(!IL::TAIL, "the inverse must end on a lane-local pass");
    constexpr int NV = FL::NBM * FL::R;                  // lane-local register slots of one core transform
    constexpr int VMAX = CfgRegs<Cfg>::VMAX;
    static_assert(NV <= VMAX, "register slots");
    const int w = rl_uniform(tid / 64), lane = tid % 64;
    const int col0 = bx * C, col = col0 + w;
    const bool colok = col < p.kx;
    const int ny = NYC > 0 ? NYC : p.ny;
    const size_t img = spec_image_elems(ny, p.pitch);
    // TWLDS: the core's twiddle table (2016 entries for 576; TWLDS = 2: the (M - 1) x Li outer twiddles behind it too) is copied
    // behind the parking space once per workgroup and read from there: one workgroup per CU exposes every L1 round trip, and 70 % of
    // this body's vector-memory reads are twiddles
    const cx<T>* tw_core = p.tw;
    const cx<T>* ctw = p.tw + PassTw<Cfg, false, 0>::TOTAL;   // W_L^(q k) at (q - 1) * Li + k
    if constexpr (TWLDS > 0) {
        constexpr int NTW = PassTw<Cfg, false, 0>::TOTAL + (TWLDS > 1 ? (M - 1) * Li : 0);
        cx<T>* const tw_lds = lds + C * LP + PARK * NT;
        for (int e = tid; e < NTW; e += NT) tw_lds[e] = p.tw[e];
        tw_core = tw_lds;      // (the first workgroup barrier of the pass stands between this copy and its first use)
        if constexpr (TWLDS > 1) ctw = tw_lds + PassTw<Cfg, false, 0>::TOTAL;
        // (the inverse halves of the split pass use the outer twiddles before their first barrier)
        if constexpr (TWLDS > 1 && (MODE == COL_SPLIT_INV || MODE == COL_SPLIT_INV_SUM)) sync.wg();
    }
    LdsView<T, 1, LdsGather<Li>::value> view_lds{lds + w * LP};

    // residue class q of the tile: element e = tid + it*NT <-> (m = e / C = r_lo + 64 it, column c = e % C), row M*m + q.  One 32-bit
    // byte offset per thread, the (class, step) part behind a uniform base: `scalar base + lane offset` loads and stores.
    // (fetch_class / park_class: the global loads of a class and their way into LDS, apart -- COL_SPLIT_FWD requests the next
    // class's rows before it transforms the current one)
    const unsigned c_ = (unsigned)tid % C, r_lo = (unsigned)tid / C;
    const bool cok = NYC > 0 ? true : (int)(col0 + c_) < p.kx;
    const unsigned boff = ((unsigned)M * r_lo * (unsigned)p.pitch + (unsigned)col0 + c_) * (unsigned)sizeof(cx<T>);
    const size_t row_bytes = (size_t)p.pitch * sizeof(cx<T>);
    auto row_ok = [&](int q, int it) -> bool { return (int)(M * (r_lo + 64 * it)) + q < ny; };   // (NYC: it < NYC / (64 M), whatever q)
    // WIDE (as colconv_wave_body's): two neighbouring columns of a row per 16-byte access -- thread -> (class row tid / (C/2) + 128 it,
    // columns 2 (tid % (C/2)), + 1); x[2 it], x[2 it + 1] hold the pair
    constexpr int NYQ = NYC / M;                                    // rows of a residue class
    // (float only: a complex double is a 16-byte access already, and the 32-byte pairs of the float64 kernel at M = 8 stayed in
    // scratch -- 160 bytes per lane, 56 stores + 56 loads per column: 4096^2 float64 61 -> 68 frames/s without them)
    constexpr bool WIDE = RL_TILE_WIDE != 0 && (sizeof(T) == 4 || RL_TILE_WIDE_F64 != 0) && NYC > 0 && NYQ % 128 == 0 && C % 2 == 0 && 2 * (NYQ / 128) <= NLD;
    struct alignas(2 * sizeof(cx<T>)) cx2 {
        cx<T> a, b;
    };
    const unsigned w_cp = (unsigned)tid % (C / 2), w_r = (unsigned)tid / (C / 2);
    const unsigned w_off = ((unsigned)M * w_r * (unsigned)p.pitch + (unsigned)col0 + 2 * w_cp) * (unsigned)sizeof(cx<T>);
    auto fetch_class = [&](const cx<T>* __restrict__ in, int q, cx<T> (&x)[NLD]) {
        if constexpr (WIDE) {
#pragma unroll
            for (int it = 0; it < NYQ / 128; ++it) {
                const cx2 y = *reinterpret_cast<const cx2*>(reinterpret_cast<const char*>(in) + ((size_t)M * 128 * it + q) * row_bytes + w_off);
                x[2 * it] = y.a;
                x[2 * it + 1] = y.b;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const char* sb = reinterpret_cast<const char*>(in) + ((size_t)M * 64 * it + q) * row_bytes;
            x[it] = mk<T>((T)0, (T)0);
            if (cok && row_ok(q, it)) x[it] = rl_ldg(sync, reinterpret_cast<const cx<T>*>(sb + boff));
        }
    };
    auto park_class = [&](const cx<T> (&x)[NLD]) {
        if constexpr (WIDE) {
#pragma unroll
            for (int it = 0; it < NYQ / 128; ++it) {
                lds[(2 * w_cp) * LP + view_lds.nat(w_r + 128 * it)] = x[2 * it];
                lds[(2 * w_cp + 1) * LP + view_lds.nat(w_r + 128 * it)] = x[2 * it + 1];
            }
#pragma unroll
            for (int e = tid; e < (Li - NYQ) * C; e += NT) lds[(e % C) * LP + view_lds.nat(NYQ + e / C)] = mk<T>((T)0, (T)0);   // rows NYQ .. Li - 1 are zero
            return;
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) lds[c_ * LP + view_lds.nat(r_lo + 64 * it)] = x[it];
    };
    auto load_class = [&](const cx<T>* __restrict__ in, int q) {
        cx<T> x[NLD];
        fetch_class(in, q, x);
        park_class(x);
    };
    auto store_class = [&](cx<T>* __restrict__ out, int q) {
        if constexpr (WIDE) {
#pragma unroll
            for (int it = 0; it < NYQ / 128; ++it) {
                cx2 y;
                y.a = rl_spec_round(lds[(2 * w_cp) * LP + view_lds.nat(w_r + 128 * it)], p.qscale);
                y.b = rl_spec_round(lds[(2 * w_cp + 1) * LP + view_lds.nat(w_r + 128 * it)], p.qscale);
                *reinterpret_cast<cx2*>(reinterpret_cast<char*>(out) + ((size_t)M * 128 * it + q) * row_bytes + w_off) = y;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            char* sb = reinterpret_cast<char*>(out) + ((size_t)M * 64 * it + q) * row_bytes;
            if (cok && row_ok(q, it))
                *reinterpret_cast<cx<T>*>(sb + boff) = rl_spec_round(lds[c_ * LP + view_lds.nat(r_lo + 64 * it)], p.qscale);
        }
    };
    // element index (within the core transform) a lane holds in register slot s; s == NV: the tail element
    auto slot_index = [&](int s) -> int {
        if (s == NV) return (64 + (lane & 7)) + bitrev3(lane >> 3) * FL::NBF;
        return (lane + (s / FL::R) * 64) + (s % FL::R) * FL::NBF;
    };
    auto slot_live = [&](int s) -> bool { return s == NV ? true : (lane + (s / FL::R) * 64) < FL::NBF; };

    using Regs = cx<T>[M][NV + 1];   // [q][slot]; slot NV = tail element (unused when the core has none)
    // Y <- the M core transforms of the residue classes of image `in`; `first`: no workgroup is still reading LDS
    auto forward_classes = [&](const cx<T>* __restrict__ in, Regs& Y, bool first) {
        if constexpr (MODE == COL_SPLIT_FWD) {   // (the same in the whole kernel: measured neutral, +-1 %)
            cx<T> x[NLD];
            fetch_class(in, 0, x);
#pragma unroll
            for (int q = 0; q < M; ++q) {
                if (q > 0) sync.wg();
                park_class(x);
                if (q + 1 < M) fetch_class(in, q + 1, x);     // in flight while this class is transformed
                sync.wg();
                cx<T> v[VMAX];
                cx<T> tl = mk<T>((T)0, (T)0);
                if (colok) run_passes<Cfg, false, 0, false>(v, tl, lane, view_lds, tw_core, sync);
#pragma unroll
                for (int s = 0; s < NV; ++s) Y[q][s] = v[s];
                Y[q][NV] = tl;
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < M; ++q) {
            if (q > 0 || !first) sync.wg();          // every wave is done with the previous class in LDS
            load_class(in, q);
            sync.wg();
            cx<T> v[VMAX];
            cx<T> tl = mk<T>((T)0, (T)0);
            if (colok) run_passes<Cfg, false, 0, false>(v, tl, lane, view_lds, tw_core, sync);
#pragma unroll
            for (int s = 0; s < NV; ++s) Y[q][s] = v[s];
            Y[q][NV] = tl;
        }
    };
    // per register slot: X[k + Li j] = radix-M butterfly of the twiddled core results, times psf_hat[view]
    auto slot_spectrum = [&](const Regs& Y, int s, int view, cx<T> (&u)[M]) {
        const int k = slot_index(s);
        const size_t pcol = ((size_t)view * p.kx + col) * L;
        u[0] = Y[0][s];
#pragma unroll
        for (int q = 1; q < M; ++q) u[q] = cmul(Y[q][s], ctw[(q - 1) * Li + k]);
        dft<M, false>(u);                                   // u[j] = X[k + Li j]
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if constexpr (REALP) u[j] = scale(u[j], p.psf_hat_re[pcol + k + Li * j]);
            else u[j] = cmul(u[j], p.psf_hat[pcol + k + Li * j]);
        }
    };
    // the way back: Z_q[k] = conj(W_L^(q k)) * sum_j conj(W_M)^(q j) X[k + Li j]
    auto slot_classes = [&](cx<T> (&u)[M], int s, Regs& Z) {
        const int k = slot_index(s);
        dft<M, true>(u);
        Z[0][s] = u[0];
#pragma unroll
        for (int q = 1; q < M; ++q) {
            const cx<T> t = ctw[(q - 1) * Li + k];
            Z[q][s] = cmul(u[q], mk<T>(t.re, -t.im));
        }
    };
    // inverse core transforms of the M classes, each stored as soon as it is back in LDS
    auto inverse_classes = [&](const Regs& Z, cx<T>* __restrict__ out) {
#pragma unroll
        for (int q = 0; q < M; ++q) {
            sync.wg();                     // LDS free: the previous class is stored (or the forward transforms are done)
            if (colok) {
                cx<T> v[VMAX];
#pragma unroll
                for (int s = 0; s < NV; ++s) v[s] = Z[q][s];
                cx<T> tl = Z[q][NV];
                run_passes<Cfg, true, 0, true>(v, tl, lane, view_lds, tw_core, sync);
                sync.wave();   // last pass' LDS reads are done before the column is overwritten
#pragma unroll
                for (int nb = 0; nb < IL::NB; ++nb) {
                    const int j = lane + nb * 64;
                    if (j < IL::NBF) {
#pragma unroll
                        for (int r = 0; r < IL::R; ++r) view_lds.template at_step<IL::NBF>(j, view_lds.nat(j), r) = v[nb * IL::R + r];
                    }
                }
            }
            sync.wg();
            store_class(out, q);
        }
    };
#define RL_FOR_LIVE_SLOTS(s)                                  \
    _Pragma("unroll") for (int s = 0; s <= NV; ++s)           \
        if ((s != NV || FL::TAIL) && slot_live(s))

    // The SPLIT pass (round 3): the column spectrum X[k + Li j] of an image parked in global memory in the order the lanes hold
    // it -- [tile][slot s][j][wave][lane], every access one contiguous 512-byte row per wave, no LDS, no barrier -- between a
    // forward launch and an inverse launch.  Multi-view plans then transform a frame's spectrum ONCE for its V views (H: 1 + V
    // transforms per column instead of 2 V) and sum the views' products slot by slot before ONE inverse transform (H_t: V + 1),
    // with one register set: neither half keeps a second set alive the way the fused multi-view modes must.
    const size_t xs_img = (size_t)((p.kx + C - 1) / C) * (NV + 1) * M * NT;
    auto xs_at = [&](int s, int j) -> size_t { return (((size_t)bx * (NV + 1) + s) * M + j) * NT + tid; };
    if constexpr (MODE == COL_PER_IMAGE && PARK > 0) {
        static_assert(PARK <= (M - 1) * (NV + 1), "at most all classes but the one being transformed");
        const int frame = by / p.V, view = by % p.V;
        const cx<T>* __restrict__ in = p.in + (size_t)(frame * p.in_sb + view * p.in_sv) * img;
        cx<T>* __restrict__ out = p.out + (size_t)by * img;
        cx<T>* __restrict__ const park = lds + C * LP + tid;       // value e of this thread at park[e * NT]
        Regs Y;
        // forward: as forward_classes(), a finished class goes to the parking space while the later ones are transformed
        // One workgroup per CU (M = 8, and M = 4 on 16-column tiles): the next class's rows are requested before this one is
        // transformed (4096^2 K = 100 34.6 -> 35.9 frames/s; 2048^2 801 -> 811).  With two workgroups per CU the other workgroup
        // fills the wait (M = 4 on 8-column tiles: 770 -> 759).
        constexpr bool PREFETCH = M >= 8 || C >= 16;
        cx<T> x[NLD];
        if constexpr (PREFETCH) fetch_class(in, 0, x);
#pragma unroll
        for (int q = 0; q < M; ++q) {
            if (q > 0) sync.wg();
            if constexpr (PREFETCH) {
                park_class(x);
                if (q + 1 < M) fetch_class(in, q + 1, x);
            } else {
                load_class(in, q);
            }
            sync.wg();
            cx<T> v[VMAX];
            cx<T> tl = mk<T>((T)0, (T)0);
            if (colok) run_passes<Cfg, false, 0, false>(v, tl, lane, view_lds, tw_core, sync);
#pragma unroll
            for (int s = 0; s <= NV; ++s) {
                const cx<T> val = s == NV ? tl : v[s];
                if (q * (NV + 1) + s < PARK) park[(q * (NV + 1) + s) * NT] = val;
                else Y[q][s] = val;
            }
        }
        // the radix-M steps and the multiplier, slot by slot: a parked value comes straight from its place, and the place takes the
        // value that waits for the inverse of class M-1, M-2, ... (same slot: read before written)
        // (M = 8, 256 registers: all of them back in registers for the loop -- the compiler's allocation: 0-8 bytes of scratch
        // against 40-60 the other way; M = 4, 128 registers: 76-84 bytes in place against 52-124)
        constexpr bool IN_PLACE = M <= 4;
        if constexpr (!IN_PLACE) {
#pragma unroll
            for (int e = 0; e < PARK; ++e) Y[e / (NV + 1)][e % (NV + 1)] = park[e * NT];
        }
        if (colok) {
            RL_FOR_LIVE_SLOTS(s) {
                cx<T> u[M];
                if constexpr (IN_PLACE) {
#pragma unroll
                    for (int q = 0; q < M; ++q)
                        if (q * (NV + 1) + s < PARK) Y[q][s] = park[(q * (NV + 1) + s) * NT];
                }
                slot_spectrum(Y, s, view, u);
                slot_classes(u, s, Y);
                if constexpr (IN_PLACE) {
#pragma unroll
                    for (int q = 0; q < M; ++q)
                        if ((M - 1 - q) * (NV + 1) + s < PARK) park[((M - 1 - q) * (NV + 1) + s) * NT] = Y[q][s];
                }
            }
        }
        if constexpr (!IN_PLACE) {
#pragma unroll
            for (int e = 0; e < PARK; ++e) park[e * NT] = Y[M - 1 - e / (NV + 1)][e % (NV + 1)];
        }
#pragma unroll
        for (int q = 0; q < M; ++q) {
            sync.wg();
            if (colok) {
                cx<T> v[VMAX];
                cx<T> tl;
#pragma unroll
                for (int s = 0; s <= NV; ++s) {
                    const int e = (M - 1 - q) * (NV + 1) + s;
                    const cx<T> val = e < PARK ? park[e * NT] : Y[q][s];
                    if (s == NV) tl = val;
                    else v[s] = val;
                }
                run_passes<Cfg, true, 0, true>(v, tl, lane, view_lds, tw_core, sync);
                sync.wave();
#pragma unroll
                for (int nb = 0; nb < IL::NB; ++nb) {
                    const int j = lane + nb * 64;
                    if (j < IL::NBF) {
#pragma unroll
                        for (int r = 0; r < IL::R; ++r) view_lds.template at_step<IL::NBF>(j, view_lds.nat(j), r) = v[nb * IL::R + r];
                    }
                }
            }
            sync.wg();
            store_class(out, q);
        }
    } else if constexpr (MODE == COL_PER_IMAGE) {
        const int frame = by / p.V, view = by % p.V;
        Regs Y;
        forward_classes(p.in + (size_t)(frame * p.in_sb + view * p.in_sv) * img, Y, true);
        if (colok) {
            RL_FOR_LIVE_SLOTS(s) {
                cx<T> u[M];
                slot_spectrum(Y, s, view, u);
                slot_classes(u, s, Y);
            }
        }
        inverse_classes(Y, p.out + (size_t)by * img);
    } else if constexpr (MODE == COL_SPLIT_FWD) {
        Regs Y;
        forward_classes(p.in + (size_t)by * img, Y, true);
        cx<T>* __restrict__ xs = p.xs_out + (size_t)by * xs_img;
        if (colok) {
            RL_FOR_LIVE_SLOTS(s) {
                const int k = slot_index(s);
                cx<T> u[M];
                u[0] = Y[0][s];
#pragma unroll
                for (int q = 1; q < M; ++q) u[q] = cmul(Y[q][s], ctw[(q - 1) * Li + k]);
                dft<M, false>(u);
#pragma unroll
                for (int j = 0; j < M; ++j) xs[xs_at(s, j)] = u[j];
            }
        }
    } else if constexpr (MODE == COL_SPLIT_INV || MODE == COL_SPLIT_INV_SUM) {
        const int frame = MODE == COL_SPLIT_INV ? by / p.V : by, view0 = MODE == COL_SPLIT_INV ? by % p.V : 0;
        const int nv = MODE == COL_SPLIT_INV ? 1 : p.V;
        const cx<T>* __restrict__ xs0 = p.xs_in + (size_t)(MODE == COL_SPLIT_INV ? frame * p.in_sb + view0 * p.in_sv : frame * p.V) * xs_img;
        Regs Z;   // [j][s]: the products' sum X[k + Li j] first, then -- slot by slot, in place -- the classes' spectra
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int s = 0; s <= NV; ++s) Z[j][s] = mk<T>((T)0, (T)0);
        }
        if (colok) {
            for (int v = 0; v < nv; ++v) {   // (all of a view's loads are independent: they are in flight together)
                const cx<T>* __restrict__ xs = xs0 + (size_t)v * xs_img;
                const size_t pcol = ((size_t)(view0 + v) * p.kx + col) * L;
                RL_FOR_LIVE_SLOTS(s) {
                    const int k = slot_index(s);
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        const cx<T> x = xs[xs_at(s, j)];
                        if constexpr (REALP) Z[j][s] = Z[j][s] + scale(x, p.psf_hat_re[pcol + k + Li * j]);
                        else Z[j][s] = Z[j][s] + cmul(x, p.psf_hat[pcol + k + Li * j]);
                    }
                }
            }
            RL_FOR_LIVE_SLOTS(s) {
                cx<T> u[M];
#pragma unroll
                for (int j = 0; j < M; ++j) u[j] = Z[j][s];
                slot_classes(u, s, Z);
            }
        }
        inverse_classes(Z, p.out + (size_t)by * img);
    } else {
        static_assert(MODE == COL_PER_IMAGE, "the long transforms have no fused multi-view modes: the split pass serves them");
    }
#undef RL_FOR_LIVE_SLOTS
}
// complex elements of one image's column spectra in register-slot order (ColParams::xs_in / xs_out)
template <class Cfg, int M, int C>
constexpr size_t outer_slots_tile_elems() {
    using FL = PassInfo<Cfg, false, Cfg::NP - 1>;
    return (size_t)(FL::NBM * FL::R + 1) * M * 64 * C;
}

// (Round 3 also carried colconv_outer4_body -- FOUR waves per column, wave g owning residue class g and the spectrum quarter
// X[k + Li g], the radix-4 step an all-to-all through LDS, one register set per wave -- for the fused multi-view modes, fed by
// ratio spectra in a 4 x 4 blocked layout: COL_HT_SUM on it beat V per-image launches by 7-21 %, the split pass above beats it
// by another 11 % with a third of the code.  Removed with the layout)

// -------------------------------- row pass ---------------------------------
// Two real image rows (2p, 2p+1) ride through one complex transform of length
// Lx (real row in .re, the next row in .im).  Depending on MODE the body does
//   [inverse rows of V spectra -> clamp -> accumulate] -> pointwise -> [forward]
// entirely in LDS/registers.
enum RowMode {
    ROW_FWD = 0,      // spec_out[f]      = rowFFT(src[f] * scale[f])
    ROW_INV = 1,      // dst[f*V+v]       = max(rowIFFT(spec_in[f*V+v]), 0)          (H output / noiseless)
    ROW_RATIO = 2,    // spec_out[f*V+v]  = rowFFT(meas[f*V+v] / max(rowIFFT(spec_in[f*V+v]), 0))
    ROW_UPDATE = 3,   // acc = sum_v max(rowIFFT(spec_in[f*V+v]),0); est[f] *= acc/norm; spec_out[f] = rowFFT(est[f])
    ROW_ADJ = 4       // dst[f] = sum_v max(rowIFFT(spec_in[f*V+v]),0) (/ norm if norm != nullptr)
};

template <typename T>
struct RowParams {
    const cx<T>* spec_in;   // [..][ny][pitch]
    cx<T>* spec_out;        // [..][ny][pitch]
    const T* src;           // ROW_FWD: [frames][ny][nx]; ROW_RATIO: measurement [frames*V][ny][nx]
    T* dst;                 // ROW_INV / ROW_ADJ output; ROW_UPDATE: estimate (read + written)
    const T* norm;          // [ny][nx] H_t(ones) (ROW_UPDATE; optional for ROW_ADJ)
    const T* scale;         // ROW_FWD: per-frame multiplier or nullptr
    const cx<T>* tw;        // [Lx]
    int ny, nx, pitch, V;
    int frames;             // streaming kernels: images covered by the launch (grid.y of the tiled kernels)
    // > 0: image i reads input spectrum i % in_mod (single-spectrum modes).  The first RL iteration starts
    // from estimate = 1 (ref:522), whose H(est) is the same for every frame: its V column-transformed
    // spectra are computed once per plan and every frame's ROW_RATIO reads them.
    int in_mod = 0;
    int sub_one = 0;        // ROW_RATIO stores rowFFT(ratio - 1), ROW_UPDATE multiplies by max(1 + acc / norm, 0): see rl_ratio
    float qscale = 1.0f;    // storage-precision study builds only (rl_spec_round)
    unsigned long long* unresolved = nullptr;   // ROW_RATIO: + the lanes that met a prediction <= 0 inside the image (rl_ratio); nullptr: not counted
};

// ONEV: compile-time single view (n_psf == 1): no accumulator registers, no view loop.  PRESUM (with ONEV, ROW_UPDATE): p.V
// spectra per image, summed before the one inverse transform.
template <class Cfg, int Q, int MODE, bool ONEV, typename T, bool PRESUM = false, class Sync>
RL_HD void rowpass_body(const RowParams<T>& p, int tid, int bx, int by, cx<T>* lds, Sync& sync) {
    static_assert(!PRESUM || (ONEV && MODE == ROW_UPDATE), "PRESUM is a form of the single-spectrum update");
    constexpr int NP = Cfg::NP, L = Cfg::L, TT = Cfg::T;
    constexpr int VMAX = CfgRegs<Cfg>::VMAX;
    const int q = tid / TT, t = tid % TT;
    const int r0 = 2 * (bx * Q + q), r1 = r0 + 1;
    const bool ok0 = r0 < p.ny, ok1 = r1 < p.ny;
    const size_t simg = spec_image_elems(p.ny, p.pitch), rimg = (size_t)p.ny * p.nx;
    LdsView<T, 1, LdsGather<L>::value, LdsPadShift<L>::value> view_lds{lds + q * LdsSlots<Cfg>::value};

    // element index held in register slot (nb, r) after an inverse / before a forward
    using IL = PassInfo<Cfg, true, NP - 1>;
    using F0 = PassInfo<Cfg, false, 0>;
    static_assert(IL::R == F0::R && IL::NB == F0::NB, "inverse must end on the forward's first radix");
    static_assert(!F0::TAIL, "the pass that touches the images must be lane-local");
    constexpr int R = F0::R, NB = F0::NB, NBF = F0::NBF;

    cx<T> v[VMAX];
    cx<T> tl = mk<T>((T)0, (T)0);   // cross-lane tail element of the TAIL passes (fft_core.hpp)
    cx<T> acc[(MODE == ROW_UPDATE || MODE == ROW_ADJ) && !ONEV ? NB * R : 1];

    // Operands of the pointwise stage are requested before the inverse transform
    // starts, so their HBM latency hides behind it (measurement for ROW_RATIO, the
    // current estimate for ROW_UPDATE).
    constexpr bool PREFETCH = (MODE == ROW_RATIO || MODE == ROW_UPDATE);
    // (float64 at L = 4608: a 576-thread workgroup is 9 waves, so a lane has at most 168 registers, and 8 complex doubles held
    // through the inverse transform were 124-460 bytes of scratch per lane: there the operands are requested behind the transform)
    constexpr bool EARLY = !(sizeof(T) == 8 && L >= 4608);
    cx<T> pre[PREFETCH ? NB * R : 1];
    rl_stamp(sync, 0);
    auto request_operands = [&] {
        const T* __restrict__ src = (MODE == ROW_RATIO ? p.src : p.dst) + (size_t)by * rimg;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int j = t + nb * TT;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = j + r * NBF;
                const bool inx = (j < NBF) && (i < p.nx);
                cx<T> m = mk<T>((T)0, (T)0);
                if (inx && ok0) m.re = src[(size_t)r0 * p.nx + i];
                if (inx && ok1) m.im = src[(size_t)r1 * p.nx + i];
                pre[nb * R + r] = m;
            }
        }
    };
    if constexpr (PREFETCH && EARLY) request_operands();

    if constexpr (MODE != ROW_FWD) {
        constexpr bool MULTI = (MODE == ROW_UPDATE || MODE == ROW_ADJ);
        const int nview = (MULTI && !ONEV) ? p.V : 1;
        if constexpr (MULTI && !ONEV) {
#pragma unroll
            for (int s = 0; s < NB * R; ++s) acc[s] = mk<T>((T)0, (T)0);
        }
        // PRESUM (`ratio - 1` updates sum the views' residuals as they are and clamp the SUM, rl_update_factor): the transform is
        // linear, so the p.V spectra of an image are added on their way in and ONE inverse transform serves them all
        const int nsum = PRESUM ? p.V : 1;
        for (int vw = 0; vw < nview; ++vw) {
            const size_t im = MULTI ? (size_t)by * p.V + vw : (p.in_mod > 0 ? (size_t)(by % p.in_mod) : (size_t)by);
            const cx<T>* __restrict__ sp = p.spec_in + im * simg;
            fft_sync<Cfg>(sync);   // LDS free
            // pack the two half spectra into one Hermitian-free complex row
            // (all global loads first, then the LDS writes)
            constexpr int NPK = (L / 2 + TT) / TT;          // ceil((L/2 + 1) / TT)
            cx<T> A[NPK], B[NPK];
#pragma unroll
            for (int it = 0; it < NPK; ++it) {
                const int k = t + it * TT;
                A[it] = mk<T>((T)0, (T)0);
                B[it] = mk<T>((T)0, (T)0);
                if (k <= L / 2 && ok0) A[it] = sp[spec_off(r0, k, p.pitch)];
                if (k <= L / 2 && ok1) B[it] = sp[spec_off(r1, k, p.pitch)];
            }
            if constexpr (PRESUM) {   // the other views, three at a time: their loads are in flight together
                for (int u0 = 1; u0 < nsum; u0 += 3) {
                    cx<T> EA[3][NPK], EB[3][NPK];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const cx<T>* __restrict__ su = sp + (size_t)(u0 + j) * simg;
                        const bool on = u0 + j < nsum;
#pragma unroll
                        for (int it = 0; it < NPK; ++it) {
                            const int k = t + it * TT;
                            EA[j][it] = mk<T>((T)0, (T)0);
                            EB[j][it] = mk<T>((T)0, (T)0);
                            if (on && k <= L / 2 && ok0) EA[j][it] = su[spec_off(r0, k, p.pitch)];
                            if (on && k <= L / 2 && ok1) EB[j][it] = su[spec_off(r1, k, p.pitch)];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
#pragma unroll
                        for (int it = 0; it < NPK; ++it) {
                            A[it] = A[it] + EA[j][it];
                            B[it] = B[it] + EB[j][it];
                        }
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NPK; ++it) {
                const int k = t + it * TT;
                if (k <= L / 2) {
                    view_lds.at(k) = mk<T>(A[it].re - B[it].im, A[it].im + B[it].re);
                    if (k > 0 && k < L / 2) view_lds.at(L - k) = mk<T>(A[it].re + B[it].im, B[it].re - A[it].im);
                }
            }
            fft_sync<Cfg>(sync);
            rl_stamp(sync, 1);   // spectrum rows have arrived and are packed in LDS
            run_passes<Cfg, true, 0, false>(v, tl, t, view_lds, p.tw, sync);
            rl_stamp(sync, 2);
            if constexpr (MULTI && !ONEV) {
                const bool raw = MODE == ROW_UPDATE && p.sub_one;   // residual views are summed as they are, the sum is clamped
#pragma unroll
                for (int s = 0; s < NB * R; ++s) {
                    acc[s].re += raw ? v[s].re : rl_clamp0(v[s].re);
                    acc[s].im += raw ? v[s].im : rl_clamp0(v[s].im);
                }
            }
        }
    }

    if constexpr (PREFETCH && !EARLY) request_operands();
    // ROW_UPDATE / ROW_ADJ: every normaliser value is requested before the first store of the
    // pointwise stage.  (vmcnt retires in order: a load issued behind a store cannot be
    // waited for without waiting for that store, and the compiler may not move the loads up
    // across stores through a pointer it cannot prove distinct -- left interleaved, the
    // stage is 2*NB*R serial memory round trips.)
    constexpr bool NORMED = (MODE == ROW_UPDATE || MODE == ROW_ADJ);
    cx<T> nrm[NORMED ? NB * R : 1];
    if constexpr (NORMED) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int j = t + nb * TT;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = j + r * NBF;
                const bool inx = (j < NBF) && (i < p.nx);
                cx<T> m = mk<T>((T)1, (T)1);
                if (p.norm && inx && ok0) m.re = p.norm[(size_t)r0 * p.nx + i];
                if (p.norm && inx && ok1) m.im = p.norm[(size_t)r1 * p.nx + i];
                nrm[nb * R + r] = m;
            }
        }
    }

    // pointwise stage on elements i = j + r*NBF (column index), rows r0 (.re), r1 (.im)
    bool unresolved = false;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = t + nb * TT;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = nb * R + r;
            const int i = j + r * NBF;
            const bool inx = (j < NBF) && (i < p.nx);
            cx<T> z = mk<T>((T)0, (T)0);
            if constexpr (MODE == ROW_FWD) {
                const T sc = p.scale ? p.scale[by] : (T)1;
                const T* __restrict__ src = p.src + (size_t)by * rimg;
                if (inx && ok0) z.re = src[(size_t)r0 * p.nx + i] * sc;
                if (inx && ok1) z.im = src[(size_t)r1 * p.nx + i] * sc;
            } else if constexpr (MODE == ROW_INV) {
                T* __restrict__ dst = p.dst + (size_t)by * rimg;
                if (inx && ok0) dst[(size_t)r0 * p.nx + i] = v[s].re > (T)0 ? v[s].re : (T)0;
                if (inx && ok1) dst[(size_t)r1 * p.nx + i] = v[s].im > (T)0 ? v[s].im : (T)0;
            } else if constexpr (MODE == ROW_RATIO) {
                if (inx && ok0) z.re = rl_ratio(pre[PREFETCH ? s : 0].re, v[s].re, p.sub_one != 0, true, unresolved);
                if (inx && ok1) z.im = rl_ratio(pre[PREFETCH ? s : 0].im, v[s].im, p.sub_one != 0, true, unresolved);
            } else if constexpr (MODE == ROW_UPDATE) {
                T* __restrict__ est = p.dst + (size_t)by * rimg;
                const bool sub = p.sub_one != 0;
                const cx<T> a = ONEV ? (sub ? v[s] : mk<T>(rl_clamp0(v[s].re), rl_clamp0(v[s].im))) : acc[ONEV ? 0 : s];
                if (inx && ok0) {
                    const size_t o = (size_t)r0 * p.nx + i;
                    z.re = pre[PREFETCH ? s : 0].re * rl_update_factor(a.re, nrm[NORMED ? s : 0].re, sub);
                    est[o] = z.re;
                }
                if (inx && ok1) {
                    const size_t o = (size_t)r1 * p.nx + i;
                    z.im = pre[PREFETCH ? s : 0].im * rl_update_factor(a.im, nrm[NORMED ? s : 0].im, sub);
                    est[o] = z.im;
                }
            } else if constexpr (MODE == ROW_ADJ) {
                T* __restrict__ dst = p.dst + (size_t)by * rimg;
                const cx<T> a = ONEV ? mk<T>(v[s].re > (T)0 ? v[s].re : (T)0, v[s].im > (T)0 ? v[s].im : (T)0) : acc[ONEV ? 0 : s];
                if (inx && ok0) {
                    const size_t o = (size_t)r0 * p.nx + i;
                    dst[o] = p.norm ? a.re / nrm[NORMED ? s : 0].re : a.re;
                }
                if (inx && ok1) {
                    const size_t o = (size_t)r1 * p.nx + i;
                    dst[o] = p.norm ? a.im / nrm[NORMED ? s : 0].im : a.im;
                }
            }
            v[s] = z;
        }
    }

    rl_stamp(sync, 3);   // pointwise stage done (its operands have arrived)
    if constexpr (MODE == ROW_RATIO) rl_count_unresolved(p.unresolved, unresolved);
    if constexpr (MODE == ROW_FWD || MODE == ROW_RATIO || MODE == ROW_UPDATE) {
        run_passes<Cfg, false, 0, true>(v, tl, t, view_lds, p.tw, sync);
        rl_stamp(sync, 4);
        // natural-order spectrum to LDS, then split it into the two rows' half spectra
        using FL = PassInfo<Cfg, false, NP - 1>;
        fft_sync<Cfg>(sync);
        if constexpr (FL::TAIL) view_lds.at((64 + (t & 7)) + bitrev3(t >> 3) * FL::NBF) = tl;
#pragma unroll
        for (int nb = 0; nb < FL::NBM; ++nb) {
            const int j = t + nb * TT;
            if (j < FL::NBF) {
#pragma unroll
                for (int r = 0; r < FL::R; ++r) view_lds.template at_step<FL::NBF>(j, view_lds.nat(j), r) = v[nb * FL::R + r];
            }
        }
        fft_sync<Cfg>(sync);
        cx<T>* __restrict__ so = p.spec_out + (size_t)by * simg;
        constexpr int NUP = (L / 2 + TT) / TT;
#pragma unroll
        for (int it = 0; it < NUP; ++it) {
            const int k = t + it * TT;
            if (k <= L / 2) {
                const cx<T> zk = view_lds.at(k), zm = view_lds.at((L - k) % L);
                if (ok0) so[spec_off(r0, k, p.pitch)] = rl_spec_round(mk<T>((T)0.5 * (zk.re + zm.re), (T)0.5 * (zk.im - zm.im)), p.qscale);
                if (ok1) so[spec_off(r1, k, p.pitch)] = rl_spec_round(mk<T>((T)0.5 * (zk.im + zm.im), (T)0.5 * (zm.re - zk.re)), p.qscale);
            }
        }
        rl_stamp(sync, 5);
    }
}

// ---------------------------------------------------------------------------------------------
// Lean single-view row kernels (ROW_RATIO, ROW_UPDATE with one view; wave-private lengths): one row pair per wave.
//
// Addressing: the row pair is a property of the wave, so all bases are scalar (rl_uniform) and a
// lane adds its own small offset: loads and stores are `scalar base + lane offset + immediate`.
// Loads are unconditional and unclamped: lanes past the end of a row read the following bytes
// (the plan allocates RL_STREAM_SLACK bytes behind every buffer for the very last row) and the
// values are discarded where they would be used.  A load under a lane-dependent branch would make
// every later counted `s_waitcnt vmcnt(N)` collapse to vmcnt(0).
// (Rounds 1-2 also carried persistent, prefetching "streaming" forms of these kernels and of the column kernel:
// whole-batch launches 8-15 % faster, a tie inside the sliced two-stream loop -- DESIGN.md section 4 -- removed.)
constexpr size_t RL_STREAM_SLACK = 16384;   // >= (L - nx) elements of any dtype for the wave-private lengths

// The two half spectra of one row pair: lanes t + 64*it of rows r0 (A) and r0 + 1 (B; row r0
// again when the pair has one row -- zeroed at the use).  Scalar row bases, no clamps.
template <class Cfg, typename T>
struct RowSpectra {
    static constexpr int NPK = (Cfg::L / 2 + 64) / 64;   // ceil((L/2 + 1) / 64)
    cx<T> A[NPK], B[NPK];
    template <class Sync>
    RL_HD void request(const RowParams<T>& p, int by, int r0, unsigned t, Sync& sync) {
        const cx<T>* __restrict__ sa = p.spec_in + (size_t)(p.in_mod > 0 ? by % p.in_mod : by) * spec_image_elems(p.ny, p.pitch) +
                                       (size_t)r0 * p.pitch;
        const cx<T>* __restrict__ sb = sa + (r0 + 1 < p.ny ? (unsigned)p.pitch : 0u);
#pragma unroll
        for (int it = 0; it < NPK; ++it) {
            A[it] = rl_ldg(sync, sa + (t + it * 64));
            B[it] = rl_ldg(sync, sb + (t + it * 64));
        }
    }
};

// One row pair of ROW_RATIO / ROW_UPDATE (single view) for a wave: spectra `in` (already
// requested) -> pack -> inverse -> pointwise -> forward -> split -> store.  by, r0: image and first
// row (wave uniform); t: lane; tw: twiddle table.
// NXC / SUBC as rowpair_body's (round 4): the row length (a multiple of 64) and `sub_one` at compile time -- no per-pixel
// selects; SUBC == 1 also puts compact twiddles on the transform that carries `ratio - 1`.
template <class Cfg, int MODE, typename T, int NXC = 0, int SUBC = -1, class View, class Sync>
RL_HD void row_item(const RowParams<T>& p, unsigned t, int by, int r0, RowSpectra<Cfg, T>& in, View view_lds, const cx<T>* tw, Sync& sync) {
    static_assert(MODE == ROW_RATIO || MODE == ROW_UPDATE, "RL modes only");
    constexpr int NP = Cfg::NP, L = Cfg::L;
    constexpr int VMAX = CfgRegs<Cfg>::VMAX;
    using IL = PassInfo<Cfg, true, NP - 1>;
    using F0 = PassInfo<Cfg, false, 0>;
    using FL = PassInfo<Cfg, false, NP - 1>;
    static_assert(IL::R == F0::R && IL::NB == F0::NB, "inverse must end on the forward's first radix");
    static_assert(!F0::TAIL, "the pass that touches the images must be lane-local");
    constexpr int R = F0::R, NB = F0::NB, NBF = F0::NBF;
    constexpr int NPK = RowSpectra<Cfg, T>::NPK;
    static_assert(NXC == 0 || (NXC % 64 == 0 && NXC <= L && NB == 1 && NBF == 64), "compile-time row length: whole 64-pixel slots");
    const int nx = NXC > 0 ? NXC : p.nx;
    const size_t simg = spec_image_elems(p.ny, p.pitch), rimg = (size_t)p.ny * nx;
    const int tl_ = (int)t;
    const int r1 = r0 + 1;
    const bool ok1 = r1 < p.ny;
    rl_stamp(sync, 0);
    // operands of the pointwise stage, requested ahead of the inverse transform: measurement (ROW_RATIO) / current estimate (ROW_UPDATE)
    cx<T> pre[NB * R];
    T* __restrict__ const est0 = p.dst + (size_t)by * rimg + (size_t)r0 * nx;
    T* __restrict__ const est1 = est0 + (ok1 ? nx : 0);
    {
        const T* __restrict__ s0 = MODE == ROW_RATIO ? p.src + (size_t)by * rimg + (size_t)r0 * nx : est0;
        const T* __restrict__ s1 = s0 + (ok1 ? nx : 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < R; ++r) pre[nb * R + r] = mk<T>(s0[t + (nb * 64 + r * NBF)], s1[t + (nb * 64 + r * NBF)]);
    }
    // pack the two half spectra into one Hermitian-free complex row
    fft_sync<Cfg>(sync);   // LDS free
#pragma unroll
    for (int it = 0; it < NPK; ++it) {
        const int kk = tl_ + it * 64;
        if (kk <= L / 2) {
            const cx<T> a = in.A[it], b = mk<T>(ok1 ? in.B[it].re : (T)0, ok1 ? in.B[it].im : (T)0);
            view_lds.at(kk) = mk<T>(a.re - b.im, a.im + b.re);
            if (kk > 0 && kk < L / 2) view_lds.at(L - kk) = mk<T>(a.re + b.im, b.re - a.im);
        }
    }
    fft_sync<Cfg>(sync);
    rl_stamp(sync, 1);
    cx<T> v[VMAX];
    cx<T> tl = mk<T>((T)0, (T)0);
    run_passes<Cfg, true, 0, false, (SUBC == 1 && MODE == ROW_UPDATE) ? RL_CT_RESIDUAL : 0>(v, tl, tl_, view_lds, tw, sync);
    rl_stamp(sync, 2);
    // normaliser values (ROW_UPDATE): all of them requested before the first store of the pointwise stage (see
    // rowpass_body), behind the inverse transform (ahead of it they cost NB*R live registers through the transform)
    cx<T> nrm[MODE == ROW_UPDATE ? NB * R : 1];
    if constexpr (MODE == ROW_UPDATE) {
        const T* __restrict__ n0 = p.norm + (size_t)r0 * nx;
        const T* __restrict__ n1 = n0 + (ok1 ? nx : 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < R; ++r) nrm[nb * R + r] = mk<T>(n0[t + (nb * 64 + r * NBF)], n1[t + (nb * 64 + r * NBF)]);
    }
    bool unresolved = false;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = (int)t + nb * 64;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = nb * R + r;
            const int i = j + r * NBF;
            const bool inx = NXC > 0 ? r * 64 < NXC : (j < NBF) && (i < nx);
            cx<T> z = mk<T>((T)0, (T)0);
            const bool sub = SUBC < 0 ? p.sub_one != 0 : SUBC != 0;
            if constexpr (MODE == ROW_RATIO) {
                z.re = inx ? rl_ratio(pre[s].re, v[s].re, sub, inx, unresolved) : (T)0;
                z.im = inx && ok1 ? rl_ratio(pre[s].im, v[s].im, sub, inx && ok1, unresolved) : (T)0;
            } else {
                z.re = inx ? pre[s].re * rl_update_factor(sub ? v[s].re : rl_clamp0(v[s].re), nrm[s].re, sub) : (T)0;
                z.im = inx && ok1 ? pre[s].im * rl_update_factor(sub ? v[s].im : rl_clamp0(v[s].im), nrm[s].im, sub) : (T)0;
                if (inx) est0[t + (nb * 64 + r * NBF)] = z.re;
                if (inx && ok1) est1[t + (nb * 64 + r * NBF)] = z.im;
            }
            v[s] = z;
        }
    }
    rl_stamp(sync, 3);
    if constexpr (MODE == ROW_RATIO) rl_count_unresolved(p.unresolved, unresolved);
    run_passes<Cfg, false, 0, true, (SUBC == 1 && MODE == ROW_RATIO) ? RL_CT_RESIDUAL : 0>(v, tl, tl_, view_lds, tw, sync);
    rl_stamp(sync, 4);
    // natural-order spectrum to LDS, then split it into the two rows' half spectra
    fft_sync<Cfg>(sync);
    if constexpr (FL::TAIL) view_lds.at((64 + (tl_ & 7)) + bitrev3(tl_ >> 3) * FL::NBF) = tl;
#pragma unroll
    for (int nb = 0; nb < FL::NBM; ++nb) {
        const int j = tl_ + nb * 64;
        if (j < FL::NBF) {
#pragma unroll
            for (int r = 0; r < FL::R; ++r) view_lds.template at_step<FL::NBF>(j, view_lds.nat(j), r) = v[nb * FL::R + r];
        }
    }
    fft_sync<Cfg>(sync);
    cx<T>* __restrict__ so0 = p.spec_out + (size_t)by * simg + (size_t)r0 * p.pitch;
    cx<T>* __restrict__ so1 = so0 + p.pitch;
#pragma unroll
    for (int it = 0; it < NPK; ++it) {
        const int kk = tl_ + it * 64;
        if (kk <= L / 2) {
            const cx<T> zk = view_lds.at(kk), zm = view_lds.at((L - kk) % L);
            so0[t + it * 64] = rl_spec_round(mk<T>((T)0.5 * (zk.re + zm.re), (T)0.5 * (zk.im - zm.im)), p.qscale);
            if (ok1) so1[t + it * 64] = rl_spec_round(mk<T>((T)0.5 * (zk.im + zm.im), (T)0.5 * (zm.re - zk.re)), p.qscale);
        }
    }
    rl_stamp(sync, 5);
}

// One row pair per wave, Q waves per workgroup, grid (ceil(pairs / Q), images); twiddles from global memory (L1).
// Replaces rowpass_body for the single-view RL modes of the wave-private lengths: scalar row bases and unconditional
// loads save ~15 % of its VALU instructions and all of its per-load exec branches.
template <class Cfg, int Q, int MODE, typename T, int NXC = 0, int SUBC = -1, class Sync>
RL_HD void rowlean_body(const RowParams<T>& p, int tid, int bx, int by, cx<T>* lds, Sync& sync) {
    static_assert(Cfg::T == 64, "lean row body needs wave-private transforms");
    static_assert((size_t)Cfg::L * sizeof(cx<T>) <= RL_STREAM_SLACK, "slack too small");   // overrun < L elements
    constexpr int LP = LdsSlots<Cfg>::value;
    const int q = rl_uniform(tid / 64);
    const unsigned t = (unsigned)(tid % 64);
    const int r0 = 2 * (bx * Q + q);
    if (r0 >= p.ny) return;   // whole wave; the wave-private row kernels have no workgroup barrier
    RowSpectra<Cfg, T> in;
    in.request(p, by, r0, t, sync);
    row_item<Cfg, MODE, T, NXC, SUBC>(p, t, by, r0, in, LdsView<T, 1, LdsGather<Cfg::L>::value>{lds + q * LP}, p.tw, sync);
}


// ---------------------------------------------------------------------------------------------
// Frame pairs: TWO FRAMES ride through one complex image -- frame 2p in the real part, frame 2p+1 in the imaginary
// part.  The PSF is real, so  IFFT2(FFT2(a + i b) * psf_hat) = conv(a, psf) + i conv(b, psf):  the row kernels
// transform whole complex rows of the pair and there is no Hermitian packing in front of the inverse transform
// and no splitting behind the forward one (rowpass_body / row_item do both around every transform, through LDS).
// The spectrum of a pair is [ny][L] complex, full width -- 576 columns against 2 x 289 for the two frames -- and the
// column kernels run on it unchanged (kx = pitch = L, psf_hat at full width).  One wave = one row of one pair:
//   ROW_FWD     spec_out = rowFFT(src[2p] + i src[2p+1])                            (the spectrum of the estimate)
//   ROW_RATIO   z = rowIFFT(spec_in); spec_out = rowFFT(meas_a / max(Re z, 0) + i meas_b / max(Im z, 0))
//   ROW_UPDATE  z = rowIFFT(spec_in); est_a *= max(Re z, 0) / norm, est_b *= max(Im z, 0) / norm; spec_out = rowFFT(est)
// The spectrum row is loaded straight into the register layout the first inverse pass takes (the layout the last forward
// pass leaves: that is what lets FFT -> pointwise -> IFFT chain in the column kernel) and stored from it.
// Wave-private lengths; an odd frame count leaves the last pair's imaginary part empty.  Multi-view plans: ROW_RATIO runs
// per (pair, view) image (p.V views), ROW_FWD / ROW_UPDATE on the pair's single (view-summed) spectrum with p.V = 1.
// NXC > 0 (round 4): the rows have exactly NXC pixels, a multiple of the slot width NBF (64 at L = 576, 256 at L = 2304) -- which register slots hold pixels is known at compile
// time (no selects, no exec branches around the loads and stores).  SUBC: p.sub_one at compile time (-1: run time).
// An odd frame count: the last pair's second frame is a PHANTOM COPY of its first (it reads the first frame's images and is
// never stored), so no value of the pointwise stage depends on whether the partner exists.
template <class Cfg, int Q, int MODE, typename T, int NXC = 0, int SUBC = -1, class Sync>
RL_HD void rowpair_body(const RowParams<T>& p, int tid, int bx, int by, cx<T>* lds, Sync& sync) {
    static_assert(Cfg::T == 64 || Q == 1, "one transform per wave, or one (workgroup-synchronous) transform per workgroup");
    static_assert(MODE == ROW_FWD || MODE == ROW_RATIO || MODE == ROW_UPDATE, "pair modes");
    constexpr int NP = Cfg::NP, L = Cfg::L, TT = Cfg::T, LP = LdsSlots<Cfg>::value, VMAX = CfgRegs<Cfg>::VMAX;
    using I0 = PassInfo<Cfg, true, 0>;        // spectrum side, on the way in
    using F0 = PassInfo<Cfg, false, 0>;       // image side
    using FL = PassInfo<Cfg, false, NP - 1>;  // spectrum side, on the way out
    using IL = PassInfo<Cfg, true, NP - 1>;
    static_assert(I0::R == FL::R && I0::NBF == FL::NBF && I0::TAIL == FL::TAIL && I0::NBM == FL::NBM, "spectrum-side layouts must agree");
    static_assert(IL::R == F0::R && IL::NB == F0::NB && !F0::TAIL, "image-side layouts must agree");
    constexpr int R = F0::R, NB = F0::NB, NBF = F0::NBF;
    static_assert(NXC == 0 || (NXC % NBF == 0 && NXC <= L && NB == 1), "compile-time row length: whole register slots (NBF pixels each)");
    const int nx = NXC > 0 ? NXC : p.nx;
    const bool sub = SUBC < 0 ? p.sub_one != 0 : SUBC != 0;
    const int q = rl_uniform(tid / TT);
    const int t = tid % TT;
    const int row = bx * Q + q;
    if (row >= p.ny) return;   // a whole wave (wave-private transforms) or the whole workgroup (Q == 1): no barrier is missed
    LdsView<T, 1, LdsGather<L>::value, LdsPadShift<L>::value> view_lds{lds + q * LP};
    const size_t simg = spec_image_elems(p.ny, p.pitch), rimg = (size_t)p.ny * nx;
    // by = pair * V + view (ROW_RATIO of a multi-view plan: measurement images are [frame][view]); V = 1 otherwise
    const int pr = NXC > 0 ? by : by / p.V, vw = NXC > 0 ? 0 : by % p.V;   // (NXC: single-view launches)
    const int V = NXC > 0 ? 1 : p.V;
    const bool okb = 2 * pr + 1 < p.frames;
    const size_t ra = ((size_t)(2 * pr * V + vw) * p.ny + row) * nx, rb = okb ? ra + (size_t)V * rimg : ra;   // frame a / b, this row
    const int tail_k = (64 + (t & 7)) + bitrev3(t >> 3) * FL::NBF;   // (TAIL passes exist for T == 64 only)
    // does register slot s hold a pixel?  (NXC: known at compile time, slot by slot)
    auto in_row = [&](int s) -> bool {
        const int nb = s / R, r = s % R;
        if constexpr (NXC > 0) return r * NBF < NXC;
        else return (t + nb * TT) < NBF && (t + nb * TT) + r * NBF < nx;
    };

    // operands of the pointwise stage: measurement / estimate requested ahead of the inverse transform, the normaliser
    // (an L2 hit: one image shared by all frames) ahead of it too or right behind it (1: registers against waits)
    cx<T> pre[NB * R];
    T nrm[MODE == ROW_UPDATE ? NB * R : 1];
    {
        const T* __restrict__ s0 = (MODE == ROW_UPDATE ? p.dst : p.src) + ra;
        const T* __restrict__ s1 = (MODE == ROW_UPDATE ? p.dst : p.src) + rb;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = (t + nb * TT) + r * NBF;
                pre[nb * R + r] = in_row(nb * R + r) ? mk<T>(s0[i], s1[i]) : mk<T>((T)0, (T)0);
            }
    }
    auto request_norm = [&] {
        if constexpr (MODE == ROW_UPDATE) {
            const T* __restrict__ n0 = p.norm + (size_t)row * nx;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = (t + nb * TT) + r * NBF;
                    nrm[nb * R + r] = in_row(nb * R + r) ? n0[i] : (T)1;
                }
        }
    };
    request_norm();   // (ahead of the inverse transform: requesting it behind the transform saved 8 registers and measured slower)
    cx<T> v[VMAX];
    cx<T> tl = mk<T>((T)0, (T)0);
    if constexpr (MODE != ROW_FWD) {
        const cx<T>* __restrict__ si = p.spec_in + (size_t)(p.in_mod > 0 ? by % p.in_mod : by) * simg + (size_t)row * p.pitch;
#pragma unroll
        for (int nb = 0; nb < I0::NBM; ++nb)
#pragma unroll
            for (int r = 0; r < I0::R; ++r) {
                const int j = t + nb * TT;
                v[nb * I0::R + r] = j < I0::NBF ? rl_ldg(sync, si + (j + r * I0::NBF)) : mk<T>((T)0, (T)0);
            }
        if constexpr (I0::TAIL) tl = rl_ldg(sync, si + tail_k);
        // (SUBC == 1: the update's inverse transform carries H_t(ratio - 1), the ratio kernel's forward transform ratio - 1)
        run_passes<Cfg, true, 0, true, (SUBC == 1 && MODE == ROW_UPDATE) ? RL_CT_RESIDUAL : 0>(v, tl, t, view_lds, p.tw, sync);
    }
    bool unresolved = false;
#pragma unroll
    for (int s = 0; s < NB * R; ++s) {
        const int nb = s / R, r = s % R;
        const int i = (t + nb * TT) + r * NBF;
        const bool inx = in_row(s);
        cx<T> z = mk<T>((T)0, (T)0);
        if constexpr (MODE == ROW_FWD) {
            z = pre[s];
        } else {
            if constexpr (MODE == ROW_RATIO) {      // (.im of a phantom partner -- an odd batch's last pair -- is not a frame: not counted)
                if (inx) z = mk<T>(rl_ratio(pre[s].re, v[s].re, sub, true, unresolved), rl_ratio(pre[s].im, v[s].im, sub, okb, unresolved));
            } else {
                if (inx) {
                    z = mk<T>(pre[s].re * rl_update_factor(sub ? v[s].re : rl_clamp0(v[s].re), nrm[s], sub),
                              pre[s].im * rl_update_factor(sub ? v[s].im : rl_clamp0(v[s].im), nrm[s], sub));
                    p.dst[ra + i] = z.re;
                    if (okb) p.dst[rb + i] = z.im;
                }
            }
        }
        v[s] = z;
    }
    if constexpr (MODE == ROW_RATIO) rl_count_unresolved(p.unresolved, unresolved);
    if constexpr (MODE == ROW_UPDATE) {
        if (p.spec_out == nullptr) return;   // last iteration of a run: nobody reads the new estimate's spectrum (uniform)
    }
    if constexpr (MODE != ROW_FWD) fft_sync<Cfg>(sync);   // the inverse's last LDS reads are done
    run_passes<Cfg, false, 0, true, (SUBC == 1 && MODE == ROW_RATIO) ? RL_CT_RESIDUAL : 0>(v, tl, t, view_lds, p.tw, sync);
    cx<T>* __restrict__ so = p.spec_out + (size_t)by * simg + (size_t)row * p.pitch;
#pragma unroll
    for (int nb = 0; nb < FL::NBM; ++nb)
#pragma unroll
        for (int r = 0; r < FL::R; ++r) {
            const int j = t + nb * TT;
            if (j < FL::NBF) so[j + r * FL::NBF] = rl_spec_round(v[nb * FL::R + r], p.qscale);
        }
    if constexpr (FL::TAIL) so[tail_k] = rl_spec_round(tl, p.qscale);
}

}  // namespace rl
