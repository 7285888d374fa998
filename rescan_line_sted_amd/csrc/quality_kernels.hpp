// quality_kernels.hpp -- launchers of quality_kernels.hip (device pointers throughout).
#pragma once
#include <hip/hip_runtime.h>

namespace rl {

// out[img][ny][nx] = f(|fftshift(fft2(x[img]))| * scale), f = log(1 + .) when log1p_ != 0.
// wx / wy: exp(-2 pi i m / n) tables of nx / ny entries; s1: nimg * ny * (nx/2 + 1) complex doubles.
hipError_t quality_fft2_magnitude(const double* x, const void* wx, const void* wy, void* s1, double* out, int nimg,
                                  int ny, int nx, double scale, int log1p_, hipStream_t s);
// cubic B-spline evaluation (coefficients already prefiltered) at n points; 0 outside the image
hipError_t quality_spline_sample(const double* coef, int ny, int nx, const double* ys, const double* xs, int n,
                                 double* out, hipStream_t s);
// psf_kernels.hip: in-place mirror-boundary cubic B-spline prefilter along both axes
hipError_t psf_spline_prefilter(double* a, int ny, int nx, hipStream_t s);

}  // namespace rl
