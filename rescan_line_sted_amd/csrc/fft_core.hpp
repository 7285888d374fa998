// fft_core.hpp -- register/LDS Stockham FFT building blocks for gfx950.
//
// One length-L complex transform is executed by T cooperating threads.  Each
// pass p has radix R_p: L/R_p butterflies, thread t owns butterflies
// j = t + nb*T.  A butterfly reads elements j + r*L/R (coalesced across t),
// applies the inter-pass twiddles, does an R-point DFT in registers and
// scatters to j0 + r*Ns (Stockham autosort, natural order in and out).
// The first pass takes its inputs from registers and the last pass leaves its
// outputs in registers, at element index j + r*L/R in both cases, so
//   * global loads/stores are issued by the first/last pass directly, and
//   * an inverse transform whose first radix equals the previous forward's
//     last radix (radix list reversed) chains on with no LDS exchange:
//     FFT -> pointwise -> IFFT and IFFT -> pointwise -> FFT stay in registers.
//
// The bodies are plain templates usable from device code and from the host
// emulator in tests/emu (same index logic, run under pthread barriers).
#pragma once

#if defined(__HIPCC__)
#define RL_HD __host__ __device__ __forceinline__
#else
#define RL_HD inline __attribute__((always_inline))
#endif

namespace rl {

template <typename T>
struct cx {
    T re, im;
};

template <typename T>
RL_HD cx<T> mk(T a, T b) {
    cx<T> r;
    r.re = a;
    r.im = b;
    return r;
}
template <typename T>
RL_HD cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.re + b.re, a.im + b.im); }
template <typename T>
RL_HD cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.re - b.re, a.im - b.im); }
template <typename T>
RL_HD cx<T> cmul(cx<T> a, cx<T> b) {
    return mk<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <typename T>
RL_HD cx<T> scale(cx<T> a, T s) { return mk<T>(a.re * s, a.im * s); }
// multiply by -i (forward transforms) or +i (inverse transforms)
template <bool INV, typename T>
RL_HD cx<T> rot90(cx<T> a) { return INV ? mk<T>(-a.im, a.re) : mk<T>(a.im, -a.re); }

// (Round 2 measured a packed-f32 form of this arithmetic -- cx<float> on a 2-vector, complex multiply = v_pk_mul_f32 + one
// v_pk_fma_f32 with op_sel / neg modifiers: 24 % fewer VALU instructions per wave in every RL kernel and no gain in
// time, a packed op costs ~1.6 scalar ops on this part; DESIGN.md section 8.  Removed.)
template <typename T>
RL_HD cx<T> cmul_const(cx<T> a, T c, T s) { return mk<T>(a.re * c - a.im * s, a.re * s + a.im * c); }
template <bool INV, typename T>
RL_HD cx<T> add_rot(cx<T> a, cx<T> b) { return a + rot90<INV>(b); }
template <bool INV, typename T>
RL_HD cx<T> sub_rot(cx<T> a, cx<T> b) { return a - rot90<INV>(b); }

// v *= exp(-+ 2 pi i m / R) for compile-time-foldable (R, m)
template <bool INV, typename T>
RL_HD cx<T> twiddle_const(cx<T> a, int R, int m) {
    m %= R;
    if (m == 0) return a;
    if (4 * m == R) return rot90<INV>(a);
    if (2 * m == R) return scale(a, (T)-1);
    if (4 * m == 3 * R) return rot90<!INV>(a);
    const double ang = 6.283185307179586476925286766559 * (double)m / (double)R;
    const T c = (T)__builtin_cos(ang);
    const T s = (T)(INV ? __builtin_sin(ang) : -__builtin_sin(ang));
    return cmul_const(a, c, s);
}

template <int R>
struct radix_split {
    static constexpr int R1 = (R % 4 == 0 && R > 4) ? 4 : (R % 2 == 0 ? 2 : (R % 3 == 0 ? 3 : 5));
    static constexpr int R2 = R / R1;
};

// In-register R-point DFT, forward sign exp(-2 pi i nk/R); INV conjugates it.
template <int R, bool INV, typename T>
RL_HD void dft(cx<T>* v) {
    if constexpr (R == 1) {
    } else if constexpr (R == 2) {
        cx<T> a = v[0] + v[1], b = v[0] - v[1];
        v[0] = a;
        v[1] = b;
    } else if constexpr (R == 3) {
        cx<T> t = v[1] + v[2];
        cx<T> m = v[0] - scale(t, (T)0.5);
        cx<T> d = scale(v[1] - v[2], (T)0.86602540378443864676372317075294);
        v[0] = v[0] + t;
        v[1] = add_rot<INV>(m, d);
        v[2] = sub_rot<INV>(m, d);
    } else if constexpr (R == 4) {
        cx<T> t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3];
        cx<T> t3 = v[1] - v[3];
        v[0] = t0 + t2;
        v[1] = add_rot<INV>(t1, t3);
        v[2] = t0 - t2;
        v[3] = sub_rot<INV>(t1, t3);
    } else if constexpr (R == 5) {
        const T c1 = (T)0.30901699437494742410229341718282, c2 = (T)-0.80901699437494742410229341718282;
        const T s1 = (T)0.95105651629515357211643933337938, s2 = (T)0.58778525229247312916870595463907;
        cx<T> a1 = v[1] + v[4], a2 = v[2] + v[3], b1 = v[1] - v[4], b2 = v[2] - v[3];
        cx<T> m1 = v[0] + scale(a1, c1) + scale(a2, c2);
        cx<T> m2 = v[0] + scale(a1, c2) + scale(a2, c1);
        cx<T> n1 = scale(b1, s1) + scale(b2, s2);
        cx<T> n2 = scale(b1, s2) - scale(b2, s1);
        v[0] = v[0] + a1 + a2;
        v[1] = add_rot<INV>(m1, n1);
        v[4] = sub_rot<INV>(m1, n1);
        v[2] = add_rot<INV>(m2, n2);
        v[3] = sub_rot<INV>(m2, n2);
    } else {
        // Cooley-Tukey R = R1*R2: n = R2*n1 + n2, k = k1 + R1*k2
        constexpr int R1 = radix_split<R>::R1, R2 = radix_split<R>::R2;
        cx<T> u[R1];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) {
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) u[n1] = v[R2 * n1 + n2];
            dft<R1, INV>(u);
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) v[R2 * k1 + n2] = twiddle_const<INV>(u[k1], R, n2 * k1);
        }
        cx<T> y[R];
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            cx<T> z[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) z[n2] = v[R2 * k1 + n2];
            dft<R2, INV>(z);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) y[k1 + R1 * k2] = z[k2];
        }
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = y[k];
    }
}

// ---------------------------------------------------------------------------
// Transform configuration: length L, T threads per transform, forward radices.
// The inverse runs the radix list reversed.
// ---------------------------------------------------------------------------
template <int L_, int T_, int... Rs>
struct FftCfg {
    static constexpr int L = L_;
    static constexpr int T = T_;
    static constexpr int NP = sizeof...(Rs);
    static constexpr int radix(int p) {
        constexpr int r[NP] = {Rs...};
        return r[p];
    }
    static constexpr int product() {
        int p = 1;
        for (int i = 0; i < NP; ++i) p *= radix(i);
        return p;
    }
    static_assert(product() == L_, "radices must multiply to L");
    // A transform on fewer than 64 threads must tile the wavefront exactly: with T = 48 a wave holds lanes of
    // two transforms in different butterfly rounds (a (4,6,8) x 48 geometry for L = 192 produced wrong rows on
    // hardware in lanes 32-47 while the host emulation, which has no wavefronts, passed).  Not supported.
    static_assert(T_ >= 64 || 64 % T_ == 0, "a sub-wave transform must use a thread count that divides 64");
};

template <class Cfg, bool INV, int P>
struct PassInfo {
    static constexpr int NP = Cfg::NP;
    static constexpr int R = Cfg::radix(INV ? NP - 1 - P : P);
    static constexpr int ns_() {
        int n = 1;
        for (int i = 0; i < P; ++i) n *= Cfg::radix(INV ? NP - 1 - i : i);
        return n;
    }
    static constexpr int NS = ns_();
    static constexpr int NBF = Cfg::L / R;                    // butterflies in this pass
    static constexpr int NB = (NBF + Cfg::T - 1) / Cfg::T;    // butterflies per thread
    // Wave-private radix-8 pass with 64 + 8 butterflies (L = 576): the 8 left-over
    // butterflies are 64 elements -- one per lane -- and are done ACROSS the lanes
    // (tail_* below) instead of as a second, 8-lanes-active round of the pass.
    static constexpr bool TAIL = (Cfg::T == 64 && R == 8 && NBF == 72);
    static constexpr int NBM = TAIL ? 1 : NB;                 // rounds done lane-locally
};

template <class Cfg>
struct CfgRegs {
    static constexpr int vmax_() {
        int m = 0;
        for (int p = 0; p < Cfg::NP; ++p) {
            int r = Cfg::radix(p);
            int nb = (Cfg::L / r + Cfg::T - 1) / Cfg::T;
            if (Cfg::T == 64 && r == 8 && Cfg::L / r == 72) nb = 1;   // PassInfo::TAIL
            if (nb * r > m) m = nb * r;
        }
        return m;
    }
    static constexpr int VMAX = vmax_();   // complex registers per thread
};

// LDS index padding: one extra slot per 8 keeps the stride-R scatter of the
// radix-8 passes off a single bank group (LDS layout).
// SH: one pad slot per 2^SH elements (3 unless fft_configs.hpp says otherwise for a length).
template <int L>
struct LdsPadShift {
    static constexpr int value = 3;
};
template <int SH = 3>
RL_HD int lds_pad(int idx) { return idx + (idx >> SH); }
template <int L>
struct LdsLen {
    static constexpr int value = L + (L >> LdsPadShift<L>::value) + 1;
};

// Which LDS layout a transform length uses.  false: natural order with one pad slot per 8
// (above).  true (set in fft_configs.hpp for the wave-private lengths): the "gathered" layout --
// natural order is stored unpadded, and an inter-pass exchange stores register slot r of
// butterfly j at r*S + j (contiguous across the lanes: the 6-cycle ds_write_b64 never conflicts)
// while the next pass gathers its operands with a per-lane base and a constant step per register.
// S is chosen per exchange so that the 32 lanes of a ds_read_b64 group fall on 32 different
// 8-byte bank pairs (Exchange::pick).  Under `idx + idx/8` every stride-1 read window spans a pad
// slot and takes 3 LDS cycles instead of 2 (46 % of the LDS cycles were bank conflicts).
template <int L>
struct LdsGather {
    static constexpr bool value = false;
};
// gathered layout: most LDS slots one exchange may take (0: no cap); fft_configs.hpp sets it for L = 576
template <int L>
struct LdsMaxSlots {
    static constexpr int value = 0;
};

// View of one transform's LDS storage.  CS = element stride (1 for the row
// kernels' [fft][idx] layout, C for the column kernels' [idx][column] layout).
// G: gathered layout (natural order unpadded).
template <typename T, int CS, bool G = false, int SH = 3>
struct LdsView {
    static constexpr bool gathered = G;
    static constexpr int pad_shift = SH;
    cx<T>* base;
    RL_HD static int nat(int idx) { return G ? idx : lds_pad<SH>(idx); }   // slot of natural-order element idx
    RL_HD cx<T>& at(int idx) const { return base[nat(idx) * CS]; }
    RL_HD cx<T>& slot(int s) const { return base[s * CS]; }
    // Element idx0 + k*STEP, with the padded position of idx0 already known.  For a step that
    // is a multiple of 8 the padding is linear, lds_pad(idx0 + k*STEP) = lds_pad(idx0) +
    // k*(STEP + STEP/8): one lane address plus a compile-time offset that folds into the
    // immediate field of the ds_read / ds_write, instead of shift + add per access.
    template <int STEP>
    RL_HD cx<T>& at_step(int idx0, int pad0, int k) const {
        if constexpr (G) return base[(idx0 + k * STEP) * CS];
        else if constexpr (STEP % (1 << SH) == 0) return base[(pad0 + k * (STEP + (STEP >> SH))) * CS];
        else return base[lds_pad<SH>(idx0 + k * STEP) * CS];
    }
};

// ---------------------------------------------------------------------------
// Inter-pass twiddles.  One table per transform length, laid out so that the
// lanes of a wave read consecutive entries:
//   [direction][pass P >= 1][r - 1][j]   = exp(-+ 2 pi i * r * (j mod Ns) / (Ns * R))
// (j = butterfly index of the pass, direction 0 = forward, 1 = inverse).  The
// table is a few KB and stays in L1/L2.
// ---------------------------------------------------------------------------
template <class Cfg, bool INV, int P>
struct PassTw {
    static constexpr int block(bool inv, int p) {   // entries of (inv, p)
        const int r = Cfg::radix(inv ? Cfg::NP - 1 - p : p);
        return p == 0 ? 0 : (r - 1) * (Cfg::L / r);
    }
    static constexpr int offset_() {
        int o = 0;
        for (int d = 0; d < 2; ++d)
            for (int p = 0; p < Cfg::NP; ++p) {
                if (d == (INV ? 1 : 0) && p == P) return o;
                o += block(d == 1, p);
            }
        return o;
    }
    static constexpr int OFFSET = offset_();
    static constexpr int main_total_() {
        int o = 0;
        for (int d = 0; d < 2; ++d)
            for (int p = 0; p < Cfg::NP; ++p) o += block(d == 1, p);
        return o;
    }
    // Behind the per-pass blocks, for the cross-lane tail butterflies of the TAIL passes (PassInfo::TAIL): per (direction,
    // pass P >= 1 with a tail) 64 entries -- the inter-pass twiddle of the ONE tail element lane l holds (input r = l >> 3 of
    // butterfly 64 + (l & 7); 1 for r = 0, so the multiply needs no lane condition) -- and last one 64-entry block with the
    // factor of the 8-point DFT's first exchange stage (tail_dft8): lane l, p = l >> 3: exp(-2 pi i (p & 3) / 8) if p >= 4, else 1.
    static constexpr bool tail_pass(bool inv, int p) {
        const int r = Cfg::radix(inv ? Cfg::NP - 1 - p : p);
        return Cfg::T == 64 && r == 8 && Cfg::L / r == 72;
    }
    static constexpr int tail_block(bool inv, int p) { return (p > 0 && tail_pass(inv, p)) ? 64 : 0; }
    static constexpr int tail_offset_() {
        int o = main_total_();
        for (int d = 0; d < 2; ++d)
            for (int p = 0; p < Cfg::NP; ++p) {
                if (d == (INV ? 1 : 0) && p == P) return o;
                o += tail_block(d == 1, p);
            }
        return o;
    }
    static constexpr int TAIL_OFFSET = tail_offset_();
    static constexpr int w8_offset_() {
        int o = main_total_();
        for (int d = 0; d < 2; ++d)
            for (int p = 0; p < Cfg::NP; ++p) o += tail_block(d == 1, p);
        return o;
    }
    static constexpr int W8_OFFSET = w8_offset_();
    static constexpr bool any_tail_() {
        for (int d = 0; d < 2; ++d)
            for (int p = 0; p < Cfg::NP; ++p)
                if (tail_pass(d == 1, p)) return true;
        return false;
    }
    static constexpr int TOTAL = W8_OFFSET + (any_tail_() ? 64 : 0);
};

// Host-side generator of that table (interleaved re, im doubles, 2*TOTAL values).
template <class Cfg, bool INV, int P>
inline void fill_pass_twiddles_one(double* out) {
    using PI = PassInfo<Cfg, INV, P>;
    if constexpr (P > 0) {
        double* dst = out + 2 * PassTw<Cfg, INV, P>::OFFSET;
        for (int r = 1; r < PI::R; ++r)
            for (int j = 0; j < PI::NBF; ++j) {
                const long double num = (long double)r * (long double)(j % PI::NS);
                const long double ang = 6.283185307179586476925286766559005768L * num / (long double)(PI::NS * PI::R);
                dst[2 * ((r - 1) * PI::NBF + j)] = (double)__builtin_cosl(ang);
                dst[2 * ((r - 1) * PI::NBF + j) + 1] = (double)(INV ? __builtin_sinl(ang) : -__builtin_sinl(ang));
            }
        if constexpr (PI::TAIL) {   // the tail element of lane l: input r = l >> 3 of butterfly 64 + (l & 7)
            double* tdst = out + 2 * PassTw<Cfg, INV, P>::TAIL_OFFSET;
            for (int l = 0; l < 64; ++l) {
                const int r = l >> 3, j = 64 + (l & 7);
                const long double num = (long double)r * (long double)(j % PI::NS);
                const long double ang = 6.283185307179586476925286766559005768L * num / (long double)(PI::NS * PI::R);
                tdst[2 * l] = (double)__builtin_cosl(ang);
                tdst[2 * l + 1] = (double)(INV ? __builtin_sinl(ang) : -__builtin_sinl(ang));
            }
        }
    }
    if constexpr (P + 1 < Cfg::NP) fill_pass_twiddles_one<Cfg, INV, P + 1>(out);
}
template <class Cfg>
inline void fill_pass_twiddles(double* out) {
    fill_pass_twiddles_one<Cfg, false, 0>(out);
    fill_pass_twiddles_one<Cfg, true, 0>(out);
    if constexpr (PassTw<Cfg, false, 0>::TOTAL > PassTw<Cfg, false, 0>::W8_OFFSET) {   // forward sign; the inverse conjugates
        double* dst = out + 2 * PassTw<Cfg, false, 0>::W8_OFFSET;
        for (int l = 0; l < 64; ++l) {
            const int p = l >> 3;
            const long double ang = 6.283185307179586476925286766559005768L * (long double)((p & 4) ? (p & 3) : 0) / 8.0L;
            dst[2 * l] = (double)__builtin_cosl(ang);
            dst[2 * l + 1] = (double)(-__builtin_sinl(ang));
        }
    }
}

// Compact twiddles (round 3; conditions fixed in round 4 -- the switches that chose them are gone, their measurements stay here):
// a pass of radix > 4 loads only w^1, w^2, w^3 and the multiples of four w^4, w^8, w^12 of a butterfly's twiddle (rows of the
// SAME table: 6/15 of a radix-16 table's lines are ever touched) and forms each of the others as ONE product of two loaded values
// (loading only the powers of two, products up to three deep, measured the same speed with more rounding; base 4 -- only w^1 and w^4
// of a radix-16 pass -- the same speed with deeper products).  Where:
//   f32  L >= 1152 (the long transforms' 69 KB tables do not stay in L1: 2048^2 point 779 -> 879 frames/s) and L <= 256 (the 128^2
//        configs: 208 -> 217 k frames/s); NOT the wave-private 576 by default -- 3-4 % faster there too, but it is the
//        estimate-carrying transforms' rounding that the pixelwise error measures -- except on the transforms that carry
//        `ratio - 1` (CT below, round 4)
//   f64  L >= 576 and L <= 256 (tables twice the size: 2048^2 253 -> 311 frames/s, 512^2 9448 -> 9600)
template <int L, typename T>
constexpr bool compact_twiddles_default() {
    return L <= 256 || L >= (sizeof(T) == 4 ? 1152 : 576);
}
// CT = 1: the compact form whatever the length (round 4: the transforms of an RL iteration that carry `ratio - 1`, a residual
// of the size of the shot noise -- their rounding error is proportional to what they carry, so the products' extra rounding is
// of no consequence there, while the transforms that carry the estimate keep their full tables).
template <class Cfg, bool INV, int P, int CT = 0, typename T>
RL_HD void pass_compute(cx<T>* v, int t, const cx<T>* __restrict__ tw) {
    using PI = PassInfo<Cfg, INV, P>;
    constexpr int R = PI::R;
#pragma unroll
    for (int nb = 0; nb < PI::NBM; ++nb) {
        const int j = t + nb * Cfg::T;
        if (j < PI::NBF) {
            if constexpr (PI::NS > 1) {
                const cx<T>* __restrict__ w = tw + PassTw<Cfg, INV, P>::OFFSET + j;
                constexpr bool COMPACT = R > 4 && (CT != 0 || compact_twiddles_default<Cfg::L, T>());
                if constexpr (COMPACT) {
                    cx<T> wp[R];
#pragma unroll
                    for (int r = 1; r < R; ++r) {
                        // loaded: r < 4 and the multiples of 4; every other one is ONE product of two loaded values
                        const bool loaded = r < 4 || r % 4 == 0;
                        wp[r] = loaded ? w[(r - 1) * PI::NBF] : cmul(wp[r - r % 4], wp[r % 4]);
                        v[nb * R + r] = cmul(v[nb * R + r], wp[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 1; r < R; ++r) v[nb * R + r] = cmul(v[nb * R + r], w[(r - 1) * PI::NBF]);
                }
            }
            dft<R, INV>(&v[nb * R]);
        }
    }
}

// Gathered layout of the exchange between pass P (producer) and pass P + 1 (consumer).
// Producer: slot(r, j) = r*S + j.  Consumer register r' of butterfly j' is transform element
// idx = j' + r'*NBF'; the producer butterfly that wrote it is j = (idx/(NS R)) NS + idx % NS with
// output r = (idx/NS) % R, and since NS R divides NBF' both split into a part that depends on j'
// only and a constant step per r':  slot = base(j') + r' * STEP.
template <class Cfg, bool INV, int P>
struct Exchange {
    using PI = PassInfo<Cfg, INV, P>;
    using PN = PassInfo<Cfg, INV, P + 1>;
    static constexpr int R = PI::R, NS = PI::NS, NBF = PI::NBF;
    static constexpr int STEP = (Cfg::L / (NS * R * PN::R)) * NS;
    static constexpr int base_(int jn, int S) { return ((jn / NS) % R) * S + (jn / (NS * R)) * NS + jn % NS; }
    // extra LDS cycles of one consumer read under stride S: lanes 0-31 and 32-63 are serviced
    // separately, each over 32 bank pairs (8 bytes per lane)
    static constexpr int conflicts(int S) {
        int extra = 0;
        for (int g = 0; g < 2; ++g) {
            int cnt[32] = {};
            int worst = 0;
            for (int l = 32 * g; l < 32 * g + 32 && l < PN::NBF; ++l) {
                const int c = ++cnt[base_(l, S) % 32];
                if (c > worst) worst = c;
            }
            extra += worst > 1 ? worst - 1 : 0;
        }
        return extra;
    }
    // (LdsMaxSlots<L>: a cap on R * S -- the LDS a transform may take -- for lengths whose residency the LDS bounds: the
    // least-conflict stride under the cap instead of the first conflict-free one)
    static constexpr int pick() {
        constexpr int cap = LdsMaxSlots<Cfg::L>::value;
        int best = NBF, bc = conflicts(NBF);
        for (int S = NBF + 1; S <= NBF + 32 && bc > 0 && (cap == 0 || R * S <= cap); ++S)
            if (conflicts(S) < bc) { best = S; bc = conflicts(S); }
        return best;
    }
    static constexpr int S = pick();
    static constexpr int SLOTS = R * S;
    RL_HD static int consumer_base(int jn) { return base_(jn, S); }
};

// LDS slots of one transform: the padded natural order, or -- gathered -- the largest exchange
// (and the unpadded natural order), rounded up to 2 mod 16 so that the 8 columns a 16-lane
// ds_write_b64 group of the column kernel's tile load touches fall on different bank pairs.
template <class Cfg, bool INV, int P>
constexpr int exchange_slots_from() {
    if constexpr (P + 1 < Cfg::NP) {
        constexpr int a = Exchange<Cfg, INV, P>::SLOTS, b = exchange_slots_from<Cfg, INV, P + 1>();
        return a > b ? a : b;
    } else {
        return 0;
    }
}
template <class Cfg>
struct LdsSlots {
    static constexpr int gathered_() {
        int m = Cfg::L;
        const int a = exchange_slots_from<Cfg, false, 0>(), b = exchange_slots_from<Cfg, true, 0>();
        if (a > m) m = a;
        if (b > m) m = b;
        return (m + 13) / 16 * 16 + 2;
    }
    static constexpr int value = LdsGather<Cfg::L>::value ? gathered_() : LdsLen<Cfg::L>::value;
};

template <class Cfg, bool INV, int P, typename T, class View>
RL_HD void pass_store_lds(const cx<T>* v, int t, View lds) {
    using PI = PassInfo<Cfg, INV, P>;
    constexpr int R = PI::R;
#pragma unroll
    for (int nb = 0; nb < PI::NBM; ++nb) {
        const int j = t + nb * Cfg::T;
        if (j < PI::NBF) {
            if constexpr (View::gathered) {
                using X = Exchange<Cfg, INV, P>;
#pragma unroll
                for (int r = 0; r < R; ++r) lds.slot(j + r * X::S) = v[nb * R + r];
            } else {
                const int j0 = (j / PI::NS) * (PI::NS * R) + (j % PI::NS);
                // NS == 1, R == 8, one pad per 8: lds_pad(8j + r) = 9j + r
                constexpr bool NINE = PI::NS == 1 && R == 8 && View::pad_shift == 3;
                const int p0 = NINE ? 9 * j : View::nat(j0);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if constexpr (NINE) lds.slot(p0 + r) = v[nb * R + r];
                    else lds.template at_step<PI::NS>(j0, p0, r) = v[nb * R + r];
                }
            }
        }
    }
}

template <class Cfg, bool INV, int P, typename T, class View>
RL_HD void pass_load_lds(cx<T>* v, int t, View lds) {
    using PI = PassInfo<Cfg, INV, P>;
    constexpr int R = PI::R;
#pragma unroll
    for (int nb = 0; nb < PI::NBM; ++nb) {
        const int j = t + nb * Cfg::T;
        if (j < PI::NBF) {
            if constexpr (View::gathered && P > 0) {
                using X = Exchange<Cfg, INV, (P > 0 ? P - 1 : 0)>;
                const int b = X::consumer_base(j);
#pragma unroll
                for (int r = 0; r < R; ++r) v[nb * R + r] = lds.slot(b + r * X::STEP);
            } else {   // natural order (first pass, or the padded layout)
                const int p0 = View::nat(j);
#pragma unroll
                for (int r = 0; r < R; ++r) v[nb * R + r] = lds.template at_step<PI::NBF>(j, p0, r);
            }
        }
    }
}

// A transform whose T threads are exactly one 64-lane wavefront exchanges data
// through LDS without workgroup barriers: LDS operations of one wave complete
// in issue order, so only the compiler must be kept from reordering them
// (Sync::wave()).  Other geometries use workgroup barriers (Sync::wg()).
template <class Cfg>
struct WavePrivate {
    static constexpr bool value = (Cfg::T == 64);
};
// The same holds when several whole transforms share a wavefront (T divides 64 and workgroups are
// built of whole transforms in thread order): the T threads of a transform sit in one wave, so its
// inter-pass exchanges need wave-level ordering only.  WavePrivate (T == 64) additionally selects
// the kernels written for one transform per wave (cross-lane tails, scalar row bases, streaming).
template <class Cfg>
struct WaveLocal {
    static constexpr bool value = (64 % Cfg::T == 0);
};

// A value that is the same in every lane of the wavefront, moved to a scalar register so that
// everything derived from it (row / image base addresses) is scalar arithmetic and the memory
// instructions take the `scalar base + 32-bit lane offset + immediate` form.
RL_HD int rl_uniform(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}

// Optional phase stamps: a Sync policy that has stamp(int) (tools/stamp_probe.hip) gets
// called at the phase boundaries of the kernel bodies; for every other policy this is nothing.
template <class Sync>
RL_HD auto rl_stamp_impl(Sync& s, int k, int) -> decltype(s.stamp(k), void()) { s.stamp(k); }
template <class Sync>
RL_HD void rl_stamp_impl(Sync&, int, long) {}
template <class Sync>
RL_HD void rl_stamp(Sync& s, int k) { rl_stamp_impl(s, k, 0); }

// Optional load policy for the spectrum loads of the kernel bodies: a Sync policy that has
// ldg(const cx<T>*) (the fused Richardson-Lucy kernel: loads that bypass the CU's L1, because the
// spectra are handed from workgroup to workgroup inside one launch) is used for them; every
// other policy loads plainly.
template <class Sync, typename T>
RL_HD auto rl_ldg_impl(Sync& s, const cx<T>* p, int) -> decltype(s.ldg(p)) { return s.ldg(p); }
template <class Sync, typename T>
RL_HD cx<T> rl_ldg_impl(Sync&, const cx<T>* p, long) { return *p; }
template <class Sync, typename T>
RL_HD cx<T> rl_ldg(Sync& s, const cx<T>* p) { return rl_ldg_impl(s, p, 0); }

template <class Cfg, class Sync>
RL_HD void fft_sync(Sync& sync) {
    if constexpr (WaveLocal<Cfg>::value) sync.wave();
    else sync.wg();
}

// ---------------------------------------------------------------------------
// Tail butterflies of a TAIL pass, across the lanes of the wave.
// Lane l = jj + 8*p  (jj = l & 7, p = l >> 3) works on butterfly j = 64 + jj and
// holds ONE element of it: input r = p of the 8-point DFT, i.e. transform element
// (64 + jj) + 72*p.  The DFT runs as three radix-2 exchange stages with partners
// l^32, l^16, l^8 (Sync::bfly).
//   DIF (natural in  -> bit-reversed out): lane p ends with output k = bitrev3(p)
//   DIT (bit-reversed in -> natural out):  lane p starts with input k = bitrev3(p)
// A forward DIF tail followed by an inverse DIT tail chains in registers exactly
// like the lane-local butterflies do.
// ---------------------------------------------------------------------------
RL_HD int bitrev3(int p) { return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1); }

// Round 4: each exchange stage is Sync::bfly<MASK>(x, lane) -- lanes with the MASK bit clear end with x + partner, lanes
// with it set with partner - x.  On the device that is two swaps and two additions in which EVERY lane does useful work
// (dev_sync.hpp: the halves trade real and imaginary parts, so one half adds / subtracts the real parts of both outputs and
// the other the imaginary parts), where rounds 1-3 fetched the partner's value with a copy, a swap and a select per
// component, computed sum AND difference and every candidate twiddle product, and selected (46 -> 18 vector instructions
// per tail).  The factor of the first stage, exp(-+ 2 pi i (p & 3) / 8) on the lanes p >= 4 and 1 elsewhere, is a per-lane
// table entry `w8` (PassTw::W8_OFFSET; forward sign, the caller conjugates), the second stage's -+i a select.
template <bool INV, bool DIT, typename T, class Sync>
RL_HD cx<T> tail_dft8(cx<T> x, int lane, cx<T> w8, Sync& sync) {
    const bool rot = (lane & 24) == 24;   // p & 3 == 3: the lanes whose second-stage factor is -+i
    if constexpr (!DIT) {
        sync.template bfly<32>(x, lane);
        x = cmul(x, w8);
        sync.template bfly<16>(x, lane);
        const cx<T> q = rot90<INV>(x);
        x = mk<T>(rot ? q.re : x.re, rot ? q.im : x.im);
        sync.template bfly<8>(x, lane);
    } else {
        sync.template bfly<8>(x, lane);
        const cx<T> q = rot90<INV>(x);
        x = mk<T>(rot ? q.re : x.re, rot ? q.im : x.im);
        sync.template bfly<16>(x, lane);
        x = cmul(x, w8);
        sync.template bfly<32>(x, lane);
    }
    return x;
}

// Element index (within the transform) of the tail value a lane holds, on the
// input side (natural order: r = p) and on the output side of a DIF tail.
template <class Cfg, bool INV, int P>
RL_HD int tail_in_index(int lane) { return (64 + (lane & 7)) + (lane >> 3) * PassInfo<Cfg, INV, P>::NBF; }
template <class Cfg, bool INV, int P>
RL_HD int tail_out_index(int lane, int k) {   // Stockham scatter position of output k of butterfly 64 + jj
    using PI = PassInfo<Cfg, INV, P>;
    const int j = 64 + (lane & 7);
    const int j0 = (j / PI::NS) * (PI::NS * PI::R) + (j % PI::NS);
    return j0 + k * PI::NS;
}

// One TAIL pass for the tail element: [load] -> inter-pass twiddle -> cross-lane DFT.
// FROM_REGS (only the chained inverse first pass): tl arrives in bit-reversed order
// from the previous forward tail and the DIT network is used (NS == 1, no twiddle).
template <class Cfg, bool INV, int P, bool FROM_REGS, typename T, class View, class Sync>
RL_HD void tail_compute(cx<T>& tl, int lane, View lds, const cx<T>* __restrict__ tw, Sync& sync) {
    using PI = PassInfo<Cfg, INV, P>;
    cx<T> w8 = tw[PassTw<Cfg, INV, P>::W8_OFFSET + lane];
    if constexpr (INV) w8.im = -w8.im;
    if constexpr (FROM_REGS) {
        static_assert(PI::NS == 1, "a register-chained tail must be the first pass");
        tl = tail_dft8<INV, true>(tl, lane, w8, sync);
    } else {
        if constexpr (View::gathered && P > 0) {   // consumer element j' = 64 + jj, register r' = lane >> 3
            using X = Exchange<Cfg, INV, (P > 0 ? P - 1 : 0)>;
            tl = lds.slot(X::consumer_base(64 + (lane & 7)) + (lane >> 3) * X::STEP);
        } else {
            tl = lds.at(tail_in_index<Cfg, INV, P>(lane));
        }
        if constexpr (PI::NS > 1) tl = cmul(tl, tw[PassTw<Cfg, INV, P>::TAIL_OFFSET + lane]);   // (1 where r = 0)
        tl = tail_dft8<INV, false>(tl, lane, w8, sync);
    }
}

// Runs passes P..NP-1.  Pass P takes its input from registers when FROM_REGS,
// otherwise from LDS (natural order).  The last pass leaves its output in
// registers: slot nb*R + r  <->  element (t + nb*T) + r*NBF of the last pass; for
// a TAIL pass additionally `tl` <-> element tail index (64+jj) + 72*bitrev3(p).
// Every LDS scatter is bracketed by syncs (all threads of the transform --
// the whole workgroup unless the transform is wave private -- must call this).
template <class Cfg, bool INV, int P, bool FROM_REGS, int CT = 0, typename T, class View, class Sync>
RL_HD void run_passes(cx<T>* v, cx<T>& tl, int t, View lds, const cx<T>* __restrict__ tw, Sync& sync) {
    using PI = PassInfo<Cfg, INV, P>;
    if constexpr (!FROM_REGS) pass_load_lds<Cfg, INV, P>(v, t, lds);
    if constexpr (PI::TAIL) tail_compute<Cfg, INV, P, FROM_REGS>(tl, t, lds, tw, sync);
    pass_compute<Cfg, INV, P, CT>(v, t, tw);
    if constexpr (P + 1 < Cfg::NP) {
        fft_sync<Cfg>(sync);   // everyone has finished reading the previous LDS contents
        pass_store_lds<Cfg, INV, P>(v, t, lds);
        if constexpr (PI::TAIL) {
            const int k = FROM_REGS ? (t >> 3) : bitrev3(t >> 3);   // DIT leaves natural order, DIF bit-reversed
            if constexpr (View::gathered) lds.slot(k * Exchange<Cfg, INV, P>::S + 64 + (t & 7)) = tl;   // output k of butterfly 64 + jj
            else lds.at(tail_out_index<Cfg, INV, P>(t, k)) = tl;
        }
        fft_sync<Cfg>(sync);
        run_passes<Cfg, INV, P + 1, false, CT>(v, tl, t, lds, tw, sync);
    }
}

}  // namespace rl
