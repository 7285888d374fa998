// psf_api.cpp -- C ABI for PSF generation (generate_psfs / psf_report /
// get_width / gaussian_filter of figure_generation/line_sted_tools.py).
// Device work in float64 (psf_kernels.hip); the Gaussian fits between device
// phases run on the host (gauss_fit.cpp).
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "gauss_fit.hpp"
#include "psf_kernels.hpp"

using namespace rl;

namespace {

// scipy gaussian_filter1d weights: radius = int(truncate*sigma + 0.5),
// exp(-0.5/sigma^2 * x^2) normalised to sum 1
std::vector<double> gaussian_weights(double sigma, double truncate, int* radius) {
    const int r = (int)(truncate * sigma + 0.5);
    std::vector<double> w(2 * r + 1);
    const double k = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = -r; i <= r; ++i) {
        w[i + r] = std::exp(k * (double)(i * i));
        sum += w[i + r];
    }
    for (double& v : w) v /= sum;
    *radius = r;
    return w;
}

struct PsfLayout {   // carving of the per-context float64 scratch
    double *w, *ry, *rx, *maxes, *sums, *b0;
    double *delta, *tmp, *g, *outer;
    double *arrays;   // exc, dep, exc_frac, dep_frac, sted, descan, rescan  (7 x ny*nx)
    double *cumu;
};

}  // namespace

extern "C" {

int rl_gauss_fit(const double* y, int n, double* p3, int* info) {
    if (!y || !p3 || n < 3) return fail(RL_ERR_INVALID, "rl_gauss_fit: need y, p3 and n >= 3");
    const int code = gauss_fit_lmdif(y, n, p3);
    if (info) *info = code;
    return RL_OK;
}

int rl_gaussian_filter(rl_ctx* ctx, const double* in, double* out, int nz, int ny, int nx, const double* sigma3,
                       double truncate) {
    if (!ctx || !in || !out || !sigma3) return fail(RL_ERR_INVALID, "NULL argument");
    if (nz < 1 || ny < 1 || nx < 1) return fail(RL_ERR_INVALID, "non-positive shape");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)nz * ny * nx;
    size_t wmax = 0;
    for (int a = 0; a < 3; ++a)
        if (sigma3[a] > 1e-15) wmax = std::max(wmax, (size_t)(2 * (int)(truncate * sigma3[a] + 0.5) + 1));
    double* work = nullptr;
    RL_TRY(ctx->psf_workspace(2 * n + wmax + 8, &work));
    double *a = work, *b = work + n, *w = work + 2 * n;
    HIP_TRY(hipMemcpyAsync(a, in, n * 8, hipMemcpyHostToDevice, ctx->stream));
    for (int axis = 0; axis < 3; ++axis) {
        if (!(sigma3[axis] > 1e-15)) continue;          // scipy skips axes with sigma <= 1e-15
        int radius = 0;
        std::vector<double> hw = gaussian_weights(sigma3[axis], truncate, &radius);
        HIP_TRY(hipMemcpyAsync(w, hw.data(), hw.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));     // hw goes out of scope
        HIP_TRY(psf_blur_axis(a, b, nz, ny, nx, axis, w, radius, ctx->stream));
        std::swap(a, b);
    }
    HIP_TRY(hipMemcpyAsync(out, a, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RL_OK;
}

// rotate of line_sted_figure_2.py:264-272 for one (ny, nx) plane: cubic-spline rotation
// about the centre (scipy.ndimage.rotate, reshape=False), clipped to [0, 1.1*max(in)].
int rl_rotate_psf(rl_ctx* ctx, const double* in, double* out, int ny, int nx, double degrees) {
    if (!ctx || !in || !out) return fail(RL_ERR_INVALID, "NULL argument");
    if (ny < 1 || nx < 1) return fail(RL_ERR_INVALID, "non-positive shape");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)ny * nx;
    double* work = nullptr;
    RL_TRY(ctx->psf_workspace(3 * n + 8, &work));
    HIP_TRY(hipMemcpyAsync(work, in, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(psf_spline_rotate(work, work + n, work + 2 * n, work + 3 * n, ny, nx, degrees, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, work + 2 * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RL_OK;
}

// generate_psfs (:168-363) for shape (1, ny, nx).
//   psf_type      0 = point, 1 = line
//   rescan_ratio  line only: > 0 forces the integer ratio, <= 0 derives it from the
//                 fitted width of the central sted row as the reference does (:252-256)
//   arrays_out    NULL or [5 (point) / 7 (line)][ny][nx]: excitation, depletion,
//                 excitation_fraction, depletion_fraction, sted, descan_sted, rescan_sted
//   rows_out      NULL or [3][nx]: central rows of excitation, sted, rescan_sted
//   scalars_out   [10]: 0 rescan ratio used, 1 ideal (float) ratio, 2-4 area sums of
//                 excitation / depletion / sted, 5-7 central-row sums of the same,
//                 8 = 1 if every central-row maximum equals its array maximum (:105-106,120), 9 reserved
int rl_psf_generate(rl_ctx* ctx, int psf_type, int ny, int nx, double exc_b, double dep_b, double sigma,
                    int rescan_ratio, double* arrays_out, double* rows_out, double* scalars_out) {
    if (!ctx || !scalars_out) return fail(RL_ERR_INVALID, "NULL argument");
    if (psf_type != 0 && psf_type != 1) return fail(RL_ERR_INVALID, "psf_type must be 0 (point) or 1 (line)");
    if (ny < 1 || nx < 1 || !(sigma > 1e-15)) return fail(RL_ERR_INVALID, "bad shape or sigma");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)ny * nx;
    int radius = 0;
    std::vector<double> hw = gaussian_weights(sigma, 4.0, &radius);
    const size_t wlen = hw.size();
    // ratio bound for the scratch: sized after the fit; first carve the fixed part
    const size_t fixed = wlen + ny + nx + 16 + 16 + nx + 4 * n + 7 * n;
    double* work = nullptr;
    const int kRatioReserve = 16;   // scratch for the rescan ring up to this ratio; larger ratios get a temporary
    RL_TRY(ctx->psf_workspace(fixed + 64 + (size_t)kRatioReserve * n, &work));
    PsfLayout L;
    double* p = work;
    L.w = p; p += wlen;
    L.ry = p; p += ny;
    L.rx = p; p += nx;
    L.maxes = p; p += 16;
    L.sums = p; p += 16;
    L.b0 = p; p += nx;
    L.delta = p; p += n;
    L.tmp = p; p += n;
    L.g = p; p += n;
    L.outer = p; p += n;
    L.arrays = p; p += 7 * n;
    double *exc = L.arrays, *dep = exc + n, *excf = dep + n, *depf = excf + n, *sted = depf + n, *descan = sted + n,
           *rescan = descan + n;
    HIP_TRY(hipMemcpyAsync(L.w, hw.data(), wlen * 8, hipMemcpyHostToDevice, s));

    // g = gaussian_filter(delta): point -> all three axes; line -> axis 2 only   (:183-193)
    HIP_TRY(psf_delta(L.delta, ny, nx, psf_type, s));
    auto blur = [&](const double* in, double* out) -> int {
        if (psf_type == 0) {
            HIP_TRY(psf_blur_axis(in, out, 1, ny, nx, 0, L.w, radius, s));
            HIP_TRY(psf_blur_axis(out, L.tmp, 1, ny, nx, 1, L.w, radius, s));
            HIP_TRY(psf_blur_axis(L.tmp, out, 1, ny, nx, 2, L.w, radius, s));
        } else {
            HIP_TRY(psf_blur_axis(in, out, 1, ny, nx, 2, L.w, radius, s));
        }
        return RL_OK;
    };
    RL_TRY(blur(L.delta, L.g));                 // excitation shape == depletion inner (:185,200)
    RL_TRY(blur(L.g, L.outer));                 // depletion outer (:202/213)
    HIP_TRY(psf_reduce(L.g, (int)n, 1, 0, L.maxes + 0, s));
    HIP_TRY(psf_reduce(L.outer, (int)n, 1, 0, L.maxes + 1, s));
    HIP_TRY(psf_stage1(L.g, L.outer, L.maxes, exc_b, exc, dep, (int)n, s));
    HIP_TRY(psf_reduce(dep, (int)n, 1, 0, L.maxes + 2, s));
    HIP_TRY(psf_stage2(exc, dep, L.maxes, dep_b, excf, depf, sted, (int)n, s));

    const int cy = ny / 2;
    double ratio_ideal = 0.0;
    int ratio = 0;
    if (psf_type == 1) {
        if (rescan_ratio > 0) {
            ratio = rescan_ratio;
        } else {   // :252-256  fit the central sted row on the host
            std::vector<double> row(nx);
            HIP_TRY(hipMemcpyAsync(row.data(), sted + (size_t)cy * nx, nx * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            double fit[3];
            gauss_fit_lmdif(row.data(), nx, fit);
            ratio_ideal = (sigma / fit[2]) * (sigma / fit[2]) + 1.0;
            ratio = (int)std::nearbyint(ratio_ideal);       // np.round: half to even
        }
        if (ratio < 1 || ratio > 4096) return fail(RL_ERR_INVALID, "line rescan ratio out of range");
        // responses of the emission blur to a centred impulse along y and along x (:258-260)
        HIP_TRY(psf_delta(L.tmp, ny, 1, 0, s));
        HIP_TRY(psf_blur_axis(L.tmp, L.ry, 1, ny, 1, 1, L.w, radius, s));
        HIP_TRY(psf_delta(L.tmp, 1, nx, 1, s));
        HIP_TRY(psf_blur_axis(L.tmp, L.rx, 1, 1, nx, 2, L.w, radius, s));
        double* big = nullptr;
        L.cumu = work + fixed + 32;
        if (ratio > kRatioReserve) {
            HIP_TRY(hipMalloc((void**)&big, (size_t)ratio * n * sizeof(double)));
            L.cumu = big;
        }
        hipError_t e = psf_rescan(sted + (size_t)cy * nx, L.w, radius, L.ry, L.rx, ny, nx, ratio, L.b0, L.cumu, descan,
                                  rescan, s);
        if (big) {
            hipError_t e2 = hipStreamSynchronize(s);
            (void)hipFree(big);
            HIP_TRY(e2);
        }
        HIP_TRY(e);
    }

    // sums and the exact-equality invariants
    const double* arr3[3] = {exc, dep, sted};
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(psf_reduce(arr3[k], (int)n, 1, 1, L.sums + k, s));                              // area sums
        HIP_TRY(psf_reduce(arr3[k] + (size_t)cy * nx, nx, 1, 1, L.sums + 3 + k, s));           // central row sums
    }
    HIP_TRY(psf_reduce(exc, (int)n, 1, 0, L.maxes + 4, s));
    HIP_TRY(psf_reduce(exc + (size_t)cy * nx, nx, 1, 0, L.maxes + 5, s));
    HIP_TRY(psf_reduce(sted, (int)n, 1, 0, L.maxes + 6, s));
    HIP_TRY(psf_reduce(sted + (size_t)cy * nx, nx, 1, 0, L.maxes + 7, s));
    if (psf_type == 1) {
        HIP_TRY(psf_reduce(rescan, (int)n, 1, 0, L.maxes + 8, s));
        HIP_TRY(psf_reduce(rescan + (size_t)cy * nx, nx, 1, 0, L.maxes + 9, s));
    }
    double hm[16], hs[16];
    HIP_TRY(hipMemcpyAsync(hm, L.maxes, 16 * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hs, L.sums, 16 * 8, hipMemcpyDeviceToHost, s));
    if (arrays_out)
        HIP_TRY(hipMemcpyAsync(arrays_out, L.arrays, (psf_type == 1 ? 7 : 5) * n * 8, hipMemcpyDeviceToHost, s));
    if (rows_out) {
        HIP_TRY(hipMemcpyAsync(rows_out, exc + (size_t)cy * nx, nx * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(rows_out + nx, sted + (size_t)cy * nx, nx * 8, hipMemcpyDeviceToHost, s));
        if (psf_type == 1)
            HIP_TRY(hipMemcpyAsync(rows_out + 2 * nx, rescan + (size_t)cy * nx, nx * 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    scalars_out[0] = (double)ratio;
    scalars_out[1] = ratio_ideal;
    for (int k = 0; k < 6; ++k) scalars_out[2 + k] = hs[k];
    bool peaks = hm[4] == hm[5] && hm[6] == hm[7];
    if (psf_type == 1) peaks = peaks && hm[8] == hm[9];
    scalars_out[8] = peaks ? 1.0 : 0.0;
    scalars_out[9] = 0.0;
    return RL_OK;
}

// psf_report (:75-166) in one call: returns report_out[8] = { resolution_improvement_
// descanned, resolution_improvement_rescanned (NaN for point), excitation_dose,
// depletion_dose, expected_emission, num_steps n, line rescan ratio, peak-invariant flag }.
// arrays_out as in rl_psf_generate (NULL: scalars only, nothing but rows crosses PCIe).
int rl_psf_report(rl_ctx* ctx, int psf_type, double exc_b, double dep_b, double steps_per_excitation_psf_width,
                  double pulses_per_position, double* arrays_out, double* report_out) {
    if (!ctx || !report_out) return fail(RL_ERR_INVALID, "NULL argument");
    const double blur_sigma = steps_per_excitation_psf_width / (2 * std::sqrt(2 * std::log(2.0)));   // :91
    const int n = 1 + 2 * (int)std::nearbyint(5 * blur_sigma);                                        // :92
    if (n < 3 || n > 4096) return fail(RL_ERR_INVALID, "steps_per_excitation_psf_width out of range");
    std::vector<double> rows(3 * (size_t)n);
    double sc[10];
    RL_TRY(rl_psf_generate(ctx, psf_type, n, n, exc_b, dep_b, blur_sigma, 0, arrays_out, rows.data(), sc));
    double fit[3];
    gauss_fit_lmdif(rows.data() + n, n, fit);                      // sted row   :108
    report_out[0] = blur_sigma / fit[2];
    report_out[1] = std::nan("");
    if (psf_type == 1) {
        gauss_fit_lmdif(rows.data() + 2 * n, n, fit);              // rescan row :121
        report_out[1] = blur_sigma / fit[2];
    }
    const int o = psf_type == 0 ? 2 : 5;                           // area sums (:135-137) or row sums (:139-144)
    report_out[2] = pulses_per_position * sc[o];
    report_out[3] = pulses_per_position * sc[o + 1];
    report_out[4] = pulses_per_position * sc[o + 2];
    report_out[5] = (double)n;
    report_out[6] = sc[0];
    report_out[7] = sc[8];
    return RL_OK;
}

}  // extern "C"
