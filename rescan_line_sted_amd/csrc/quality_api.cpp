// quality_api.cpp -- C ABI of the reconstruction-quality metrics (SURVEY 8 f-4 and the
// FT-error history of Deconvolver.record_iteration, row a-11).  Host float64 in and out;
// all arithmetic on the device in float64 (quality_kernels.hip).
#include <vector>

#include "ctx.hpp"
#include "quality_kernels.hpp"

using namespace rl;

namespace {
struct DevBuf {   // scoped device allocation
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int get(size_t bytes) {
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 8));
        return RL_OK;
    }
};
}  // namespace

extern "C" {

int rl_fft2_magnitude(rl_ctx* ctx, const double* x, int n_img, int ny, int nx, double scale, int log1p, double* out) {
    if (!ctx || !x || !out) return fail(RL_ERR_INVALID, "NULL argument");
    if (n_img < 1 || ny < 1 || nx < 1) return fail(RL_ERR_INVALID, "non-positive shape");
    if (n_img > 65535 || ny > 65535) return fail(RL_ERR_UNSUPPORTED, "more than 65535 images or rows");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)n_img * ny * nx, ns = (size_t)n_img * ny * (nx / 2 + 1);
    void *wx = nullptr, *wy = nullptr;
    RL_TRY(ctx->plain_twiddles(nx, &wx));
    RL_TRY(ctx->plain_twiddles(ny, &wy));
    DevBuf in, s1, res;
    RL_TRY(in.get(n * 8));
    RL_TRY(s1.get(ns * 16));
    RL_TRY(res.get(n * 8));
    HIP_TRY(hipMemcpyAsync(in.p, x, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(quality_fft2_magnitude((const double*)in.p, wx, wy, s1.p, (double*)res.p, n_img, ny, nx, scale, log1p,
                                   ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, res.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RL_OK;
}

int rl_spline_sample(rl_ctx* ctx, const double* image, int ny, int nx, const double* ys, const double* xs, int n,
                     double* out) {
    if (!ctx || !image || !ys || !xs || !out) return fail(RL_ERR_INVALID, "NULL argument");
    if (ny < 1 || nx < 1 || n < 0) return fail(RL_ERR_INVALID, "bad shape");
    if (n == 0) return RL_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t ni = (size_t)ny * nx;
    DevBuf buf;
    RL_TRY(buf.get((ni + 3 * (size_t)n) * 8));
    double *coef = (double*)buf.p, *dy = coef + ni, *dx = dy + n, *dz = dx + n;
    HIP_TRY(hipMemcpyAsync(coef, image, ni * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dy, ys, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dx, xs, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(psf_spline_prefilter(coef, ny, nx, ctx->stream));
    HIP_TRY(quality_spline_sample(coef, ny, nx, dy, dx, n, dz, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, dz, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RL_OK;
}

}  // extern "C"
