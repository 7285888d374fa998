// aux_kernels.hpp -- host-callable launchers of aux_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_table.hpp"

namespace rl {
hipError_t aux_fill(int dtype, void* p, size_t n, double value, hipStream_t s);
// psf_dev: [n_psf][py][px] float64 on the device.  wx/wy: float64 twiddle tables
// exp(-2 pi i m / L).  s1_dev: scratch [n_psf][py][kx] complex128.
// out: [n_psf][ly][pitch] complex of `dtype` (or [n_psf][kx][ly] when transposed), scaled by 1/(ly*lx).
hipError_t aux_psf_spectrum(int dtype, const double* psf_dev, const void* wx_dev, const void* wy_dev, void* s1_dev,
                            void* out, int n_psf, int py, int px, int ly, int lx, int kx, int pitch, int transposed,
                            hipStream_t s);
// list_ws: device scratch of at least aux_poisson_workspace_bytes(n_pix * n_img) bytes.
// image0: index of the first image in the Philox counter (the draws of an image do not depend on
// which slice of the batch a launch covers).
size_t aux_poisson_workspace_bytes(size_t total_pixels);
// frame_seeds / frame_ids (device arrays, one entry per frame of V views each) override seed / image0:
// image (frame f, view v) then draws with seed frame_seeds[f] and image index frame_ids[f]*V + v.
hipError_t aux_poisson(int dtype, const void* noiseless, void* noisy, unsigned n_pix, unsigned n_img, unsigned image0,
                       unsigned long long seed, int rng_kind, void* list_ws, hipStream_t s,
                       const unsigned long long* frame_seeds = nullptr, const unsigned* frame_ids = nullptr, unsigned V = 1);
// float64 stack [frames][n] (device) -> plan dtype, each frame scaled to sum target[f]
// Every `sums` argument below is device memory of aux_sums_elems(frames) doubles: the frames' sums, then scratch for the partial
// sums of large frames (aux_kernels.hip frame_sums).
constexpr size_t kSumChunks = 64;
inline size_t aux_sums_elems(size_t frames) { return frames * (1 + kSumChunks); }
// (target: device array of `frames` doubles; target == nullptr: no scaling)
// want_sums: fill `sums` even without a target (the caller reads the frames' levels)
hipError_t aux_scale_convert(int dtype, const double* src, void* dst, size_t n, size_t frames, const double* target,
                             double* sums, hipStream_t s, bool want_sums = false);
// *flag = 1 if any of the n values is negative, else 0 (flag: device memory)
hipError_t aux_any_negative(int dtype, const void* src, size_t n, int* flag, hipStream_t s);
// sums[f] = sum of image f of a stack [frames][n] in the plan's dtype (float64 accumulation)
hipError_t aux_image_sums(int dtype, const void* src, size_t n, size_t frames, double* sums, hipStream_t s);
// as aux_scale_convert with frame f taken from staged object idx[f] (n_unique staged objects; sums: [n_unique] scratch)
hipError_t aux_scale_convert_indexed(int dtype, const double* src, const unsigned* idx, size_t n_unique, void* dst, size_t n, size_t frames,
                                     const double* target, double* sums, hipStream_t s);
hipError_t aux_to_f64(int dtype, const void* src, double* dst, size_t total, hipStream_t s);
// dst[i] = (dst type) src[i]: a plan buffer into a result buffer of another arithmetic type (same type: a device copy)
hipError_t aux_cast(int dtype_src, const void* src, int dtype_dst, void* dst, size_t total, hipStream_t s);
// re[i] = z[i].re for n complex values of `dtype`; stats (device, 2 doubles) <- max |im|, max(|re|, |im|)
hipError_t aux_split_real(int dtype, const void* z, size_t n, void* re, double* stats, hipStream_t s);
// out [ny][nx] (plan dtype) = sum_v max(conv_same(ones, psf_v), 0) from the PSFs' float64 integral images
// integral_dev [V][py+1][px+1] (I[a][b] = sum of psf[a' < a][b' < b]): H_t(ones) to float64 rounding, no transform
hipError_t aux_box_norm(int dtype, const double* integral_dev, void* out, int V, int py, int px, int ny, int nx, hipStream_t s);
}  // namespace rl
