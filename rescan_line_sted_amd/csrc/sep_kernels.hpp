// sep_kernels.hpp -- host-callable launchers of sep_kernels.hip (direct separable stencils for rank-1 PSFs)
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_table.hpp"

namespace rl {
enum { SEP_STORE_ = 0, SEP_RATIO_ = 1, SEP_SUM_ = 2, SEP_UPDATE_ = 3 };   // epilogues of the column pass
// out[img] = row stencil of in[img / in_div] with taps_v[img % V] (px taps); images = number of output images
hipError_t sep_rows(int dtype, const void* in, void* out, const void* taps_v, int images, int ny, int nx, int px, int V,
                    int in_div, hipStream_t s);
// column stencil of tmp with taps_u (py taps) + epilogue `mode`; frames_or_images: frames for SUM / UPDATE, else images
hipError_t sep_cols(int dtype, int mode, const void* tmp, const void* taps_u, const void* aux, const void* norm, void* dst,
                    int frames_or_images, int ny, int nx, int py, int V, hipStream_t s);
// Both passes in one kernel (input tile + halo in LDS).  taps_uf / taps_vf: [V][8*ceil(py/8)] / [V][8*ceil(px/8)],
// FLIPPED (f[k] = taps[n-1-k]) and zero padded.  STORE / RATIO: in [frames], dst [frames*V]; SUM / UPDATE: in
// [frames*V], dst [frames].  sep2d_fits: the tile fits the 160 KB of LDS.
bool sep2d_fits(int dtype, int py, int px, int V);
// the direct 2-D stencil for PSFs that are not rank 1 (sep2d with taps_vf == nullptr, taps_uf = [V][px][8*ceil(py/8)]:
// F[l][k] = p[py-1-k][px-1-l], zero padded along k): the input tile and the taps fit LDS
bool direct2d_fits(int dtype, int py, int px, int V);
// the two-pass form (sep_rows + sep_cols) fits LDS: py up to 609 taps in f32, 289 in f64 (the plan falls back to the FFT path beyond)
bool sep_two_pass_fits(int dtype, int py, int px);
hipError_t sep2d(int dtype, int mode, const void* in, const void* taps_uf, const void* taps_vf, const void* aux, const void* norm,
                 void* dst, int frames, int ny, int nx, int py, int px, int V, hipStream_t s);
}  // namespace rl
