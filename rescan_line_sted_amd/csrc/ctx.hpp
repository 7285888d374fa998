// ctx.hpp -- internals shared by the C-ABI translation units (not installed)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rlsted.h"
#include "kernel_table.hpp"

namespace rl {
std::string& last_error();                 // thread local
int fail(int code, const std::string& msg);
bool debug_sync();                         // RLSTED_DEBUG_SYNC set: sync + check after every launch
}  // namespace rl

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return rl::fail(RL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)
#define RL_TRY(expr)                \
    do {                            \
        int r_ = (expr);            \
        if (r_ != RL_OK) return r_; \
    } while (0)

using rl::fail;

struct rl_ctx {
    // grow-only device scratch for the PSF pipeline (float64 elements)
    double* psf_work = nullptr;
    size_t psf_work_elems = 0;
    int psf_workspace(size_t elems, double** out) {
        if (elems > psf_work_elems) {
            if (psf_work) (void)hipFree(psf_work);
            psf_work = nullptr;
            psf_work_elems = 0;
            HIP_TRY(hipMalloc((void**)&psf_work, elems * sizeof(double)));
            psf_work_elems = elems;
        }
        *out = psf_work;
        return RL_OK;
    }

    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::pair<int, int>, void*> tw;   // (L, dtype) -> device table
    std::map<int, bool> prepared;

    // plain table exp(-2 pi i m / L), float64 (PSF spectrum DFT)
    int plain_twiddles(int L, void** out) {
        auto key = std::make_pair(-L, (int)RL_F64);
        auto it = tw.find(key);
        if (it != tw.end()) {
            *out = it->second;
            return RL_OK;
        }
        std::vector<double> h(2 * (size_t)L);
        for (int m = 0; m < L; ++m) {
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)L;
            h[2 * m] = (double)cosl(a);
            h[2 * m + 1] = (double)sinl(a);
        }
        void* dev = nullptr;
        HIP_TRY(hipMalloc(&dev, sizeof(double) * 2 * L));
        HIP_TRY(hipMemcpy(dev, h.data(), sizeof(double) * 2 * L, hipMemcpyHostToDevice));
        tw[key] = dev;
        *out = dev;
        return RL_OK;
    }

    // per-pass twiddle table of one transform length (layout: fft_core.hpp PassTw)
    // column == true: the table of the column kernels' geometry (fft_configs.hpp ColCfgFor)
    int twiddles(const rl::KernelTable* t, int dtype, void** out, bool column = false) {
        auto key = std::make_pair(t->L + (column ? (1 << 24) : 0), dtype);
        auto it = tw.find(key);
        if (it != tw.end()) {
            *out = it->second;
            return RL_OK;
        }
        const int count = column ? t->tw_count_col[dtype] : t->tw_count;
        const size_t n = 2 * (size_t)(count > 0 ? count : 1);
        std::vector<double> h(n, 0.0);
        if (count > 0) (column ? t->fill_tw_col[dtype] : t->fill_tw)(h.data());
        void* dev = nullptr;
        if (dtype == RL_F64) {
            HIP_TRY(hipMalloc(&dev, sizeof(double) * n));
            HIP_TRY(hipMemcpy(dev, h.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        } else {
            std::vector<float> f(h.begin(), h.end());
            HIP_TRY(hipMalloc(&dev, sizeof(float) * n));
            HIP_TRY(hipMemcpy(dev, f.data(), sizeof(float) * n, hipMemcpyHostToDevice));
        }
        tw[key] = dev;
        *out = dev;
        return RL_OK;
    }

    int prepare(const rl::KernelTable* t) {
        if (prepared[t->L]) return RL_OK;
        HIP_TRY(t->prepare());
        prepared[t->L] = true;
        return RL_OK;
    }
};

