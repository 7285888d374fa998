// psf_kernels.hpp -- host-callable launchers of psf_kernels.hip (all float64, device pointers)
#pragma once
#include <hip/hip_runtime.h>

namespace rl {
hipError_t psf_blur_axis(const double* in, double* out, int nz, int ny, int nx, int axis, const double* w, int radius,
                         hipStream_t s);
hipError_t psf_delta(double* out, int ny, int nx, int kind, hipStream_t s);
// op 0 = max, 1 = sum over in[i*stride], i < count; result to *out (device)
hipError_t psf_reduce(const double* in, int count, int stride, int op, double* out, hipStream_t s);
hipError_t psf_stage1(const double* g, const double* outer, const double* maxes, double exc_b, double* exc,
                      double* dep_raw, int n, hipStream_t s);
hipError_t psf_stage2(const double* exc, double* dep, const double* maxes, double dep_b, double* exc_frac,
                      double* dep_frac, double* sted, int n, hipStream_t s);
hipError_t psf_rescan(const double* sted_row, const double* w, int radius, const double* ry, const double* rx, int ny,
                      int nx, int ratio, double* b0, double* cumu, double* descan, double* rescan, hipStream_t s);
hipError_t psf_spline_rotate(const double* in, double* work, double* out, double* vmax, int ny, int nx,
                             double degrees, hipStream_t s);
}  // namespace rl
