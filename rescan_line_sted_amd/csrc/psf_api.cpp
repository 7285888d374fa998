// psf_api.cpp -- C ABI for PSF generation (generate_psfs / psf_report /
// get_width / gaussian_filter of figure_generation/line_sted_tools.py).
// Device work in float64 (psf_kernels.hip); the Gaussian fits between device
// phases run on the host (gauss_fit.cpp).
#include <cmath>
#include <cstring>
#include <map>
#include <utility>
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
    return rl_rotate_psf_stack(ctx, in, out, 1, ny, nx, degrees);
}

int rl_rotate_psf_stack(rl_ctx* ctx, const double* in, double* out, int nz, int ny, int nx, double degrees) {
    if (!ctx || !in || !out) return fail(RL_ERR_INVALID, "NULL argument");
    if (nz < 1 || ny < 1 || nx < 1) return fail(RL_ERR_INVALID, "non-positive shape");
    const size_t n = (size_t)ny * nx;
    if ((size_t)nz * n > (size_t)0x7fffffff) return fail(RL_ERR_UNSUPPORTED, "stack too large");
    HIP_TRY(hipSetDevice(ctx->device));
    double* work = nullptr;                     // [nz][n] in, [n] spline coefficients, [nz][n] out, the clip level
    RL_TRY(ctx->psf_workspace((2 * (size_t)nz + 1) * n + 8, &work));
    double *d_in = work, *d_coef = work + (size_t)nz * n, *d_out = d_coef + n, *d_max = d_out + (size_t)nz * n;
    HIP_TRY(hipMemcpyAsync(d_in, in, (size_t)nz * n * 8, hipMemcpyHostToDevice, ctx->stream));
    for (int z = 0; z < nz; ++z)                // the clip level is the maximum of the WHOLE array (fig2:271), taken with the first plane
        HIP_TRY(psf_spline_rotate(d_in + (size_t)z * n, d_coef, d_out + (size_t)z * n, d_max, ny, nx, degrees, ctx->stream,
                                  z == 0 ? (int)((size_t)nz * n) : 0));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)nz * n * 8, hipMemcpyDeviceToHost, ctx->stream));
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
// Host buffers of rl_psf_generate_line_extras for the generate call it wraps (this thread's next one)
namespace {
struct LineExtras {
    double* emission = nullptr;   // [ny][nx]
    double* unscaled = nullptr;   // [ny][ratio * nx]
};
thread_local LineExtras g_line_extras;
}  // namespace

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
        const LineExtras extras = g_line_extras;
        g_line_extras = LineExtras();
        if (e == hipSuccess && extras.unscaled)   // rescanned_signal_cumu (:266,298): the ring before the roll and the binning
            e = hipMemcpyAsync(extras.unscaled, L.cumu, (size_t)ratio * n * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && extras.emission) {   // emission_psf = gaussian_filter(point delta, sigma) (:258-260); delta / g / tmp are free by now
            e = psf_delta(L.delta, ny, nx, 0, s);
            if (e == hipSuccess) e = psf_blur_axis(L.delta, L.g, 1, ny, nx, 0, L.w, radius, s);
            if (e == hipSuccess) e = psf_blur_axis(L.g, L.tmp, 1, ny, nx, 1, L.w, radius, s);
            if (e == hipSuccess) e = psf_blur_axis(L.tmp, L.g, 1, ny, nx, 2, L.w, radius, s);
            if (e == hipSuccess) e = hipMemcpyAsync(extras.emission, L.g, n * sizeof(double), hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess && (extras.unscaled || extras.emission)) e = hipStreamSynchronize(s);
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

int rl_psf_generate_line_extras(rl_ctx* ctx, int ny, int nx, double exc_b, double dep_b, double sigma, int rescan_ratio,
                                double* emission_psf_out, double* rescan_unscaled_out) {
    if (!ctx) return fail(RL_ERR_INVALID, "NULL argument");
    if (rescan_ratio < 1) return fail(RL_ERR_INVALID, "rescan_ratio must be the ratio a previous rl_psf_generate reported (scalars_out[0])");
    g_line_extras.emission = emission_psf_out;
    g_line_extras.unscaled = rescan_unscaled_out;
    double scalars[10];
    const int rc = rl_psf_generate(ctx, 1, ny, nx, exc_b, dep_b, sigma, rescan_ratio, nullptr, nullptr, scalars);
    g_line_extras = LineExtras();
    return rc;
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

// psf_report for many parameter sets at once (line_sted_figure_1.py:33-48 sweeps 480 of them,
// line_sted_figure_a1.py:29,64,102,172 a few hundred more): one launch per pipeline stage over all sets
// instead of ~20 launches and two synchronisations per set.  Sets that share (psf_type, sampling) share
// the blurred delta and its double blur, which depend on nothing else (:183-213).  Every set's numbers
// are bit for bit those of rl_psf_report: same per-element arithmetic, same reduction order.
//   params[n_sets][5] = { psf_type (0 point / 1 line), excitation_brightness, depletion_brightness,
//                         steps_per_excitation_psf_width, pulses_per_position }
//   report_out[n_sets][8] as rl_psf_report; arrays_out: NULL, or n_sets pointers (each NULL or
//   [5 | 7][n][n] as rl_psf_generate).
int rl_psf_report_batch(rl_ctx* ctx, int n_sets, const double* params, double* report_out, double* const* arrays_out) {
    if (!ctx || !params || !report_out) return fail(RL_ERR_INVALID, "NULL argument");
    if (n_sets < 1) return fail(RL_ERR_INVALID, "n_sets < 1");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    struct Group { int type, n, radius; double sigma; std::vector<double> w; long long g, gmax, w_off, ryx; };
    std::map<std::pair<int, long long>, int> group_of;     // (type, bits of steps) -> group
    std::vector<Group> groups;
    std::vector<PsfSetDesc> sets(n_sets);
    std::vector<double> sigma(n_sets), pulses(n_sets);
    long long top = 0;     // workspace carving, in doubles
    auto carve = [&](size_t count) { const long long o = top; top += (long long)((count + 1) & ~(size_t)1); return o; };
    int max_n = 0;
    for (int i = 0; i < n_sets; ++i) {
        const double* q = params + 5 * (size_t)i;
        const int type = (int)q[0];
        if ((type != 0 && type != 1) || (double)type != q[0]) return fail(RL_ERR_INVALID, "psf_type must be 0 (point) or 1 (line)");
        const double steps = q[3];
        const double bs = steps / (2 * std::sqrt(2 * std::log(2.0)));                    // :91
        const int n = 1 + 2 * (int)std::nearbyint(5 * bs);                               // :92
        if (!(bs > 1e-15) || n < 3 || n > 4096) return fail(RL_ERR_INVALID, "steps_per_excitation_psf_width out of range");
        long long bits;
        memcpy(&bits, &steps, 8);
        auto key = std::make_pair(type, bits);
        auto it = group_of.find(key);
        if (it == group_of.end()) {
            Group g;
            g.type = type; g.n = n; g.sigma = bs;
            g.w = gaussian_weights(bs, 4.0, &g.radius);
            g.g = carve(2 * (size_t)n * n);
            g.gmax = carve(2);
            g.w_off = carve(g.w.size());
            g.ryx = carve(2 * (size_t)n);
            it = group_of.emplace(key, (int)groups.size()).first;
            groups.push_back(std::move(g));
        }
        const Group& g = groups[it->second];
        PsfSetDesc& d = sets[i];
        d.type = type; d.n = n; d.radius = g.radius; d.ratio = 0;
        d.exc_b = q[1]; d.dep_b = q[2];
        d.g = g.g; d.gmax = g.gmax; d.w = g.w_off; d.ryx = g.ryx;
        d.arrays = carve(7 * (size_t)n * n);
        d.b0 = carve(n);
        d.scal = carve(32);
        d.cumu = 0;
        sigma[i] = bs;
        pulses[i] = q[4];
        max_n = std::max(max_n, n);
    }
    const long long fixed_top = top;
    const long long scratch = carve(3 * (size_t)max_n * max_n);   // delta, tmp of the group blurs
    double* base = nullptr;
    // the rescan rings are sized after the fits; reserve for ratio <= 16 now (grown below if ever needed)
    long long ring_reserve = 0;
    for (const PsfSetDesc& d : sets)
        if (d.type == 1) ring_reserve += 16LL * d.n * d.n;
    RL_TRY(ctx->psf_workspace((size_t)(top + ring_reserve) + 64, &base));

    // ---- per shape group: g = blur(delta), outer = blur(g), their maxima, the impulse responses ----
    for (const Group& g : groups) {
        const int n = g.n;
        double *w = base + g.w_off, *G = base + g.g, *outer = G + (size_t)n * n, *delta = base + scratch, *tmp = delta + (size_t)max_n * max_n;
        HIP_TRY(hipMemcpyAsync(w, g.w.data(), g.w.size() * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(psf_delta(delta, n, n, g.type, s));
        auto blur = [&](const double* in, double* out) -> int {
            if (g.type == 0) {
                HIP_TRY(psf_blur_axis(in, out, 1, n, n, 0, w, g.radius, s));
                HIP_TRY(psf_blur_axis(out, tmp, 1, n, n, 1, w, g.radius, s));
                HIP_TRY(psf_blur_axis(tmp, out, 1, n, n, 2, w, g.radius, s));
            } else {
                HIP_TRY(psf_blur_axis(in, out, 1, n, n, 2, w, g.radius, s));
            }
            return RL_OK;
        };
        RL_TRY(blur(delta, G));
        RL_TRY(blur(G, outer));
        HIP_TRY(psf_reduce(G, n * n, 1, 0, base + g.gmax, s));
        HIP_TRY(psf_reduce(outer, n * n, 1, 0, base + g.gmax + 1, s));
        if (g.type == 1) {
            HIP_TRY(psf_delta(tmp, n, 1, 0, s));
            HIP_TRY(psf_blur_axis(tmp, base + g.ryx, 1, n, 1, 1, w, g.radius, s));
            HIP_TRY(psf_delta(tmp, 1, n, 1, s));
            HIP_TRY(psf_blur_axis(tmp, base + g.ryx + n, 1, 1, n, 2, w, g.radius, s));
        }
    }
    HIP_TRY(hipStreamSynchronize(s));   // the groups' host weight vectors were copied asynchronously

    // ---- all sets: saturation maths ----
    PsfSetDesc* d_sets = nullptr;
    ReduceJob* d_jobs = nullptr;
    std::vector<ReduceJob> jobs;
    HIP_TRY(hipMalloc((void**)&d_sets, sizeof(PsfSetDesc) * n_sets));
    HIP_TRY(hipMalloc((void**)&d_jobs, sizeof(ReduceJob) * 16 * n_sets));
    struct Free { void *a, *b; ~Free() { (void)hipFree(a); (void)hipFree(b); } } free_guard{d_sets, d_jobs};
    auto run_jobs = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(ReduceJob) * jobs.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(psf_reduce_jobs(base, base, d_jobs, (int)jobs.size(), s));
        HIP_TRY(hipStreamSynchronize(s));
        jobs.clear();
        return RL_OK;
    };
    HIP_TRY(hipMemcpyAsync(d_sets, sets.data(), sizeof(PsfSetDesc) * n_sets, hipMemcpyHostToDevice, s));
    HIP_TRY(psf_batch_stage1(base, d_sets, n_sets, max_n, s));
    for (const PsfSetDesc& d : sets) jobs.push_back({d.arrays + (long long)d.n * d.n, d.scal + 2, d.n * d.n, 1, 0, 0});   // max dep_raw
    RL_TRY(run_jobs());
    HIP_TRY(psf_batch_stage2(base, d_sets, n_sets, max_n, s));

    // ---- line sets: central sted rows -> host fits -> integer rescan ratios (:252-256) -> rescan simulation ----
    std::vector<std::vector<double>> sted_rows(n_sets);
    for (int i = 0; i < n_sets; ++i) {
        const PsfSetDesc& d = sets[i];
        sted_rows[i].resize(d.n);
        HIP_TRY(hipMemcpyAsync(sted_rows[i].data(), base + d.arrays + 4 * (size_t)d.n * d.n + (size_t)(d.n / 2) * d.n,
                               d.n * 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<double> ratio_ideal(n_sets, 0.0), sted_fit(n_sets);
    long long ring_top = top;
    int max_ratio = 0;
    bool any_line = false;
    for (int i = 0; i < n_sets; ++i) {
        PsfSetDesc& d = sets[i];
        double fit[3];
        gauss_fit_lmdif(sted_rows[i].data(), d.n, fit);          // the same fit serves :108 and :252
        sted_fit[i] = fit[2];
        if (d.type != 1) continue;
        any_line = true;
        ratio_ideal[i] = (sigma[i] / fit[2]) * (sigma[i] / fit[2]) + 1.0;
        d.ratio = (int)std::nearbyint(ratio_ideal[i]);
        if (d.ratio < 1 || d.ratio > 4096) return fail(RL_ERR_INVALID, "line rescan ratio out of range");
        d.cumu = ring_top;
        ring_top += (long long)d.ratio * d.n * d.n;
        max_ratio = std::max(max_ratio, d.ratio);
    }
    if (any_line) {
        if (ring_top > top + ring_reserve) {     // larger rings than reserved: grow (offsets stay valid: same carving)
            std::vector<double> keep((size_t)fixed_top);
            HIP_TRY(hipMemcpy(keep.data(), base, keep.size() * 8, hipMemcpyDeviceToHost));
            RL_TRY(ctx->psf_workspace((size_t)ring_top + 64, &base));
            HIP_TRY(hipMemcpy(base, keep.data(), keep.size() * 8, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMemcpyAsync(d_sets, sets.data(), sizeof(PsfSetDesc) * n_sets, hipMemcpyHostToDevice, s));
        HIP_TRY(psf_batch_rescan(base, d_sets, n_sets, max_n, max_ratio, s));
    }

    // ---- sums, maxima (:105-106,120,134-144) ----
    for (const PsfSetDesc& d : sets) {
        const long long nn = (long long)d.n * d.n, row = (long long)(d.n / 2) * d.n;
        const long long arr3[3] = {d.arrays, d.arrays + nn, d.arrays + 4 * nn};
        for (int k = 0; k < 3; ++k) {
            jobs.push_back({arr3[k], d.scal + 16 + k, (int)nn, 1, 1, 0});
            jobs.push_back({arr3[k] + row, d.scal + 16 + 3 + k, d.n, 1, 1, 0});
        }
        jobs.push_back({d.arrays, d.scal + 4, (int)nn, 1, 0, 0});
        jobs.push_back({d.arrays + row, d.scal + 5, d.n, 1, 0, 0});
        jobs.push_back({d.arrays + 4 * nn, d.scal + 6, (int)nn, 1, 0, 0});
        jobs.push_back({d.arrays + 4 * nn + row, d.scal + 7, d.n, 1, 0, 0});
        if (d.type == 1) {
            jobs.push_back({d.arrays + 6 * nn, d.scal + 8, (int)nn, 1, 0, 0});
            jobs.push_back({d.arrays + 6 * nn + row, d.scal + 9, d.n, 1, 0, 0});
        }
    }
    RL_TRY(run_jobs());
    std::vector<double> scal(32 * (size_t)n_sets), rescan_row;
    for (int i = 0; i < n_sets; ++i)
        HIP_TRY(hipMemcpyAsync(scal.data() + 32 * (size_t)i, base + sets[i].scal, 32 * 8, hipMemcpyDeviceToHost, s));
    std::vector<std::vector<double>> rescan_rows(n_sets);
    for (int i = 0; i < n_sets; ++i) {
        const PsfSetDesc& d = sets[i];
        if (d.type == 1) {
            rescan_rows[i].resize(d.n);
            HIP_TRY(hipMemcpyAsync(rescan_rows[i].data(), base + d.arrays + 6 * (size_t)d.n * d.n + (size_t)(d.n / 2) * d.n,
                                   d.n * 8, hipMemcpyDeviceToHost, s));
        }
        if (arrays_out && arrays_out[i])
            HIP_TRY(hipMemcpyAsync(arrays_out[i], base + d.arrays, (size_t)(d.type == 1 ? 7 : 5) * d.n * d.n * 8,
                                   hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < n_sets; ++i) {
        const PsfSetDesc& d = sets[i];
        const double* hm = scal.data() + 32 * (size_t)i;
        const double* hs = hm + 16;
        double* r = report_out + 8 * (size_t)i;
        r[0] = sigma[i] / sted_fit[i];
        r[1] = std::nan("");
        if (d.type == 1) {
            double fit[3];
            gauss_fit_lmdif(rescan_rows[i].data(), d.n, fit);      // :121
            r[1] = sigma[i] / fit[2];
        }
        const int o = d.type == 0 ? 0 : 3;
        r[2] = pulses[i] * hs[o];
        r[3] = pulses[i] * hs[o + 1];
        r[4] = pulses[i] * hs[o + 2];
        r[5] = (double)d.n;
        r[6] = (double)d.ratio;
        bool peaks = hm[4] == hm[5] && hm[6] == hm[7];
        if (d.type == 1) peaks = peaks && hm[8] == hm[9];
        r[7] = peaks ? 1.0 : 0.0;
    }
    return RL_OK;
}

}  // extern "C"
