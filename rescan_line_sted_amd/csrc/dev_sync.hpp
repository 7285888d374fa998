// dev_sync.hpp -- the device-side Sync policy of the kernel bodies in conv_kernels.hpp
// (workgroup barrier, wave-private LDS fence, cross-lane exchanges of gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "fft_core.hpp"

namespace rl {

// (Round 4 measured non-temporal (`nt`) loads for the once-used spectrum and image data, to leave the CU's L1 to the twiddle
// tables every wave re-reads: 20.2 k -> 14.9 k frames/s at 512^2, with `nt` stores too 15.7 k.  Removed.)
struct DevSync {
    __device__ __forceinline__ void wg() const { __syncthreads(); }
    // One wave exchanging data with itself through LDS: the hardware completes a
    // wave's LDS operations in issue order, so only compiler reordering has to be
    // prevented (wavefront-scope fences emit no instructions).
    __device__ __forceinline__ void wave() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // Radix-2 exchange stage across the lanes (fft_core.hpp tail_dft8): lanes whose MASK bit is clear end with x + partner,
    // lanes whose bit is set with partner - x (partner = lane ^ MASK; all 64 lanes active).  VALU cross-lane moves of gfx950,
    // no LDS round trip:
    //   MASK 32  v_permlane32_swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b
    //   MASK 16  v_permlane16_swap(a, b): the odd 16-lane rows of a <-> the even rows of b
    // Swapping the REAL part register with the IMAGINARY part register leaves, in the lanes with the bit clear, the real
    // parts of both partners (own, partner's) and in the lanes with the bit set their imaginary parts (partner's, own) --
    // (a, b) = (value of the bit-clear lane, value of the bit-set lane) in every lane.  s = a + b and d = a - b are then both
    // outputs of one component, no lane computes anything that is thrown away, and the same swap of (s, d) puts them back:
    // two swaps + two additions per stage (a copy, a swap and a select per component plus sum, difference and a select before).
    //   MASK 8   no swap of that distance: partner through DPP row_ror:8 folded into the addition, own value sign-flipped
    //            by the lane's bit (v_xor).
    template <int MASK>
    __device__ __forceinline__ void swap_u32(unsigned& a, unsigned& b) const {
        if constexpr (MASK == 32) {
            auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            a = r[0];
            b = r[1];
        } else {
            static_assert(MASK == 16, "swap distance");
            auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
            a = r[0];
            b = r[1];
        }
    }
    template <int MASK>
    __device__ __forceinline__ void bfly(cx<float>& x, int lane) const {
        if constexpr (MASK == 8) {
            const unsigned m = ((unsigned)lane << 28) & 0x80000000u;   // bit 3 of the lane -> sign bit
            const float pr = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x.re), 0x128, 0xf, 0xf, false));
            const float pi = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x.im), 0x128, 0xf, 0xf, false));
            x.re = __uint_as_float(__float_as_uint(x.re) ^ m) + pr;
            x.im = __uint_as_float(__float_as_uint(x.im) ^ m) + pi;
        } else {
            unsigned a = __float_as_uint(x.re), b = __float_as_uint(x.im);
            swap_u32<MASK>(a, b);
            const float fa = __uint_as_float(a), fb = __uint_as_float(b);
            unsigned s = __float_as_uint(fa + fb), d = __float_as_uint(fa - fb);
            swap_u32<MASK>(s, d);
            x.re = __uint_as_float(s);
            x.im = __uint_as_float(d);
        }
    }
    template <int MASK>
    __device__ __forceinline__ void bfly(cx<double>& x, int lane) const {
        if constexpr (MASK == 8) {
            const unsigned long long m = (unsigned long long)(((unsigned)lane << 28) & 0x80000000u) << 32;
            auto part = [](double v) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(v);
                const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x128, 0xf, 0xf, false);
                const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x128, 0xf, 0xf, false);
                return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            };
            const double pr = part(x.re), pi = part(x.im);
            x.re = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(x.re) ^ m)) + pr;
            x.im = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(x.im) ^ m)) + pi;
        } else {
            const unsigned long long br = (unsigned long long)__double_as_longlong(x.re), bi = (unsigned long long)__double_as_longlong(x.im);
            unsigned alo = (unsigned)br, ahi = (unsigned)(br >> 32), blo = (unsigned)bi, bhi = (unsigned)(bi >> 32);
            swap_u32<MASK>(alo, blo);
            swap_u32<MASK>(ahi, bhi);
            const double fa = __longlong_as_double((long long)(((unsigned long long)ahi << 32) | alo));
            const double fb = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
            const unsigned long long s = (unsigned long long)__double_as_longlong(fa + fb), d = (unsigned long long)__double_as_longlong(fa - fb);
            unsigned slo = (unsigned)s, shi = (unsigned)(s >> 32), dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            swap_u32<MASK>(slo, dlo);
            swap_u32<MASK>(shi, dhi);
            x.re = __longlong_as_double((long long)(((unsigned long long)shi << 32) | slo));
            x.im = __longlong_as_double((long long)(((unsigned long long)dhi << 32) | dlo));
        }
    }
};

}  // namespace rl
