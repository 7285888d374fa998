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
                             double degrees, hipStream_t s, int max_count);
}  // namespace rl

// ---- batched PSF pipeline (rl_psf_report_batch): one launch per stage over all parameter sets ----
namespace rl {
struct PsfSetDesc {          // one parameter set; offsets are in doubles from the batch workspace base
    int type, n, radius, ratio;
    double exc_b, dep_b;
    long long g;             // the set's shape group: g [n*n], outer [n*n] behind it
    long long gmax;          // group maxima: max g, max outer
    long long w;             // Gaussian weights of the group (2 radius + 1)
    long long ryx;           // group impulse responses ry [n], rx [n]
    long long arrays;        // exc, dep, exc_frac, dep_frac, sted, descan, rescan  (7 x n*n)
    long long b0;            // [n]
    long long cumu;          // [n][ratio*n]
    long long scal;          // 16 maxima + 16 sums
};
struct ReduceJob {           // out[] <- max (op 0) / sum (op 1) of base[off + i*stride], i < count
    long long off, out;
    int count, stride, op, pad;
};
hipError_t psf_reduce_jobs(const double* base, double* out_base, const ReduceJob* jobs, int n_jobs, hipStream_t s);
hipError_t psf_batch_stage1(double* base, const PsfSetDesc* sets, int n_sets, int max_n, hipStream_t s);
hipError_t psf_batch_stage2(double* base, const PsfSetDesc* sets, int n_sets, int max_n, hipStream_t s);
hipError_t psf_batch_rescan(double* base, const PsfSetDesc* sets, int n_sets, int max_n, int max_ratio, hipStream_t s);
}  // namespace rl
