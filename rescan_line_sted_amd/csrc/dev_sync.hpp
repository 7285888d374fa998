// dev_sync.hpp -- the device-side Sync policy of the kernel bodies in conv_kernels.hpp
// (workgroup barrier, wave-private LDS fence, cross-lane exchanges of gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace rl {

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
    // value held by lane (this lane ^ MASK), MASK = 8 / 16 / 32; all 64 lanes must be active.
    // VALU cross-lane moves of gfx950 (no LDS round trip, unlike ds_bpermute):
    //   ^32  v_permlane32_swap: swaps the upper half of one register with the lower half of another
    //   ^16  v_permlane16_swap: swaps the odd 16-lane rows of one with the even rows of another
    //   ^8   DPP row_ror:8 (rotate the 16-lane row by half its length)
    template <int MASK>
    __device__ __forceinline__ unsigned shfl_xor_u32(unsigned u) const {
        if constexpr (MASK == 32) {
            auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            return (threadIdx.x & 32) ? r[0] : r[1];
        } else if constexpr (MASK == 16) {
            auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            return (threadIdx.x & 16) ? r[0] : r[1];
        } else {
            static_assert(MASK == 8, "unsupported exchange distance");
            return (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, false);
        }
    }
    template <int MASK>
    __device__ __forceinline__ float shfl_xor(float v) const {
        return __uint_as_float(shfl_xor_u32<MASK>(__float_as_uint(v)));
    }
    template <int MASK>
    __device__ __forceinline__ double shfl_xor(double v) const {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = shfl_xor_u32<MASK>((unsigned)b), hi = shfl_xor_u32<MASK>((unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
};

}  // namespace rl
