// philox_poisson.hpp -- counter-based Poisson sampler for the shot-noise step
// of create_data_from_object (line_sted_tools.py:508-511; the reference calls
// numpy.random.poisson, whose MT19937 stream cannot be reproduced in parallel).
//
// SPEC (DESIGN.md "Device Poisson"; the oracle twin oracle/philox_poisson.py
// implements the same spec independently in numpy and must agree bit for bit):
//   * generator  Philox4x32-10, key = (seed lo32, seed hi32),
//                counter = (pixel, image, block, 0x504F4953) with pixel = row*nx
//                + col, image = frame*n_psf + view, block = 0,1,2,... per pixel
//   * uniforms   each block gives u1 = dbl(x0,x1), u2 = dbl(x2,x3),
//                dbl(a,b) = ((a>>5)*2^26 + (b>>6)) * 2^-53   in [0,1)
//   * lambda==0  -> 0;  lambda < 10 -> multiplication method, two uniforms per
//                block;  lambda >= 10 -> Hormann's PTRS (the same algorithm and
//                constants numpy's legacy generator uses), one attempt per block
//   * log / exp / log-gamma are evaluated by the polynomial code below using
//     only IEEE +,-,*,/,sqrt,floor in double with contraction OFF, so that a
//     CPU twin reproduces every accept/reject decision exactly.
//   * at most 64 blocks per pixel, then the last candidate is taken.
#pragma once
#include <stdint.h>
#include "fft_core.hpp"   // RL_HD

// No fused multiply-add inside the sampler: the CPU twin has none.
#if defined(__clang__)
#define RL_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define RL_FP_STRICT
#endif

namespace rl {

struct Philox4 {
    uint32_t x[4];
};

RL_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    Philox4 o;
    o.x[0] = c0; o.x[1] = c1; o.x[2] = c2; o.x[3] = c3;
    return o;
}

RL_HD double u53(uint32_t a, uint32_t b) {
    RL_FP_STRICT
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

union DblBits {
    double d;
    uint64_t u;
};

// natural log, x > 0 finite and normal.  x = m * 2^e, m in [sqrt(1/2), sqrt 2)
RL_HD double det_log(double x) {
    RL_FP_STRICT
    DblBits b;
    b.d = x;
    int e = (int)((b.u >> 52) & 0x7ff) - 1023;
    b.u = (b.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;   // m in [1,2)
    double m = b.d;
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

// exp(x) for x in [-40, 0]
RL_HD double det_exp(double x) {
    RL_FP_STRICT
    const double n = __builtin_floor(x * 1.4426950408889634 + 0.5);
    const double r = (x - n * 0.6931471803691238) - n * 1.9082149292705877e-10;
    double p = 1.0 / 6227020800.0;      // 1/13!
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    DblBits b;
    b.u = (uint64_t)(1023 + (int)n) << 52;   // 2^n, n in [-58, 0]
    return p * b.d;
}

// log(k!) for integer-valued k >= 0
RL_HD double det_logfact(double k) {
    RL_FP_STRICT
    if (k < 20.5) {
        double f = 1.0;
        for (int i = 2; i <= (int)k; ++i) f = f * (double)i;   // exact up to 20!
        return det_log(f);
    }
    const double x = k + 1.0;
    const double xi = 1.0 / x, x2 = xi * xi;
    double c = -1.0 / 1680.0;
    c = c * x2 + 1.0 / 1260.0;
    c = c * x2 - 1.0 / 360.0;
    c = c * x2 + 1.0 / 12.0;
    return ((x - 0.5) * det_log(x) - x) + 0.9189385332046727 + c * xi;
}

// First attempt only (block 0) for lam >= 10: returns true and the variate when the
// PTRS squeeze accepts it (~87 % of pixels), false when the pixel needs the full
// sampler.  Exactly the first loop trip of philox_poisson(), so a pixel finished
// here gets the identical value.
RL_HD bool philox_poisson_fast(double lam, uint64_t seed, uint32_t image, uint32_t pixel, double* out) {
    RL_FP_STRICT
    if (!(lam > 0.0)) {
        *out = 0.0;
        return true;
    }
    if (lam < 10.0) return false;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const double slam = __builtin_sqrt(lam);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double vr = 0.9277 - 3.6224 / (b - 2.0);
    const Philox4 o = philox4x32_10(pixel, image, 0u, 0x504F4953u, k0, k1);
    const double U = u53(o.x[0], o.x[1]) - 0.5;
    const double V = u53(o.x[2], o.x[3]);
    const double us = 0.5 - (U < 0.0 ? -U : U);
    const double k = __builtin_floor((2.0 * a / us + b) * U + lam + 0.43);
    *out = k;
    return us >= 0.07 && V <= vr;
}

// One PTRS attempt (lam >= 10) with Philox block `blk`: returns true when the candidate *kout is accepted.
// philox_poisson() below is a loop over this function; the device sampler calls it one attempt at a time so
// that the pixels still undecided can be repacked densely between attempts.
RL_HD bool philox_ptrs_attempt(double lam, uint32_t k0, uint32_t k1, uint32_t image, uint32_t pixel, uint32_t blk, double* kout) {
    RL_FP_STRICT
    const double slam = __builtin_sqrt(lam);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double vr = 0.9277 - 3.6224 / (b - 2.0);
    const Philox4 o = philox4x32_10(pixel, image, blk, 0x504F4953u, k0, k1);
    const double U = u53(o.x[0], o.x[1]) - 0.5;
    const double V = u53(o.x[2], o.x[3]);
    const double us = 0.5 - (U < 0.0 ? -U : U);
    const double k = __builtin_floor((2.0 * a / us + b) * U + lam + 0.43);
    *kout = k;
    if (us >= 0.07 && V <= vr) return true;                      // ~87 % of attempts leave here
    if (k < 0.0 || (us < 0.013 && V > us)) return false;
    if (!(V > 0.0)) return true;
    // slow path: the logarithms are evaluated only here (same values as if hoisted)
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    const double loglam = det_log(lam);
    const double lhs = (det_log(V) + det_log(invalpha)) - det_log(a / (us * us) + b);
    const double rhs = (k * loglam - lam) - det_logfact(k);
    return lhs <= rhs;
}

constexpr uint32_t kPoissonMaxBlocks = 64;   // attempts per pixel, then the last candidate is taken

// One Poisson variate for (seed, image, pixel).
RL_HD double philox_poisson(double lam, uint64_t seed, uint32_t image, uint32_t pixel) {
    RL_FP_STRICT
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (!(lam > 0.0)) return 0.0;
    if (lam < 10.0) {
        const double enlam = det_exp(-lam);
        double X = 0.0, prod = 1.0;
        for (uint32_t blk = 0; blk < kPoissonMaxBlocks; ++blk) {
            const Philox4 o = philox4x32_10(pixel, image, blk, 0x504F4953u, k0, k1);
            prod = prod * u53(o.x[0], o.x[1]);
            if (!(prod > enlam)) return X;
            X = X + 1.0;
            prod = prod * u53(o.x[2], o.x[3]);
            if (!(prod > enlam)) return X;
            X = X + 1.0;
        }
        return X;
    }
    double k = 0.0;
    for (uint32_t blk = 0; blk < kPoissonMaxBlocks; ++blk)
        if (philox_ptrs_attempt(lam, k0, k1, image, pixel, blk, &k)) return k;
    return k < 0.0 ? 0.0 : k;
}

}  // namespace rl
