// stamp_probe.hip -- developer tool (not part of the library): runs the L = 576 column and
// row kernels of the RL iteration on a synthetic resident batch with s_memtime stamps at
// the phase boundaries of the kernel bodies and prints where a wave's lifetime goes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Iinclude \
//         -Irescan_line_sted_amd/csrc tools/stamp_probe.hip -o build/stamp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "conv_kernels.hpp"
#include "fft_configs.hpp"
#include "dev_sync.hpp"

using namespace rl;
constexpr int L = 576, NS = 8;
using CF = CfgFor<L>;
using Cfg = CF::Cfg;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct StampSync : DevSync {
    unsigned long long* buf;   // [wave][NS]
    __device__ __forceinline__ void stamp(int k) const {
        unsigned long long t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        if ((threadIdx.x & 63) == 0) {
            const size_t wave = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x) / 64;
            buf[wave * NS + k] = t;
        }
    }
};
// variant that does not drain outstanding memory operations at the stamp (phase times then
// show where the wave actually stalls, not the latency of what was issued)
struct StampSyncLazy : DevSync {
    unsigned long long* buf;
    __device__ __forceinline__ void stamp(int k) const {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        if ((threadIdx.x & 63) == 0) {
            const size_t wave = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x) / 64;
            buf[wave * NS + k] = t;
        }
    }
};

template <class S, int C>
__global__ void __launch_bounds__(64 * C) k_col(const ColParams<float> p, unsigned long long* buf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    S s; s.buf = buf;
    colconv_wave_body<Cfg, C, COL_PER_IMAGE, float>(p, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, reinterpret_cast<cx<float>*>(smem), s);
}
template <class S, int Q, int MODE>
__global__ void __launch_bounds__(64 * Q) k_row(const RowParams<float> p, unsigned long long* buf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    S s; s.buf = buf;
    rowpass_body<Cfg, Q, MODE, true, float>(p, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y, reinterpret_cast<cx<float>*>(smem), s);
}

#ifndef PROBE_COL_MINW
#define PROBE_COL_MINW 1
#endif
#ifndef PROBE_ROW_MINW
#define PROBE_ROW_MINW 1
#endif
#ifndef PROBE_QS
#define PROBE_QS 8
#endif
struct PlainSync : DevSync {
    unsigned long long* buf;
};
template <class S, int C>
__global__ void __launch_bounds__(64 * C, PROBE_COL_MINW) k_cols(const ColParams<float> p, unsigned long long* buf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    S s; s.buf = buf;
    colstream_body<Cfg, C, float>(p, (int)threadIdx.x, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<cx<float>*>(smem), s);
}
template <class S, int Q, int MODE>
__global__ void __launch_bounds__(64 * Q, PROBE_ROW_MINW) k_rows(const RowParams<float> p, unsigned long long* buf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    S s; s.buf = buf;
    rowstream_body<Cfg, Q, MODE, float>(p, (int)threadIdx.x, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<cx<float>*>(smem), s);
}
template <typename F>
static int resident(F* fn, int threads, size_t lds) {
    int per_cu = 0;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds));
    printf("    (%d workgroups of %d threads per CU, %zu B LDS)\n", per_cu, threads, lds);
    return per_cu * 256;
}

static void report(const char* name, const std::vector<unsigned long long>& h, size_t waves, int ns, float ms) {
    std::vector<double> d(ns, 0.0);
    double life = 0;
    unsigned long long t_min = ~0ull, t_max = 0;
    size_t used = 0;
    for (size_t w = 0; w < waves; ++w) {
        bool ok = true;
        for (int k = 0; k < ns; ++k) ok = ok && h[w * NS + k] != 0;
        if (!ok) continue;   // a wave that skipped a phase (idle column of the last tile)
        ++used;
        for (int k = 1; k < ns; ++k) d[k] += (double)(h[w * NS + k] - h[w * NS + k - 1]);
        life += (double)(h[w * NS + ns - 1] - h[w * NS]);
        t_min = std::min(t_min, h[w * NS]);
        t_max = std::max(t_max, h[w * NS + ns - 1]);
    }
    printf("%-28s %.3f ms  span %llu ticks (%.1f MHz tick)  wave lifetime %.0f ticks; phases:", name, ms,
           (unsigned long long)(t_max - t_min), (double)(t_max - t_min) / (ms * 1e3), life / used);
    for (int k = 1; k < ns; ++k) printf(" %d:%.0f", k, d[k] / used);
    printf("\n");
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int ny = 512, nx = 512, kx = L / 2 + 1, pitch = (kx + 7) / 8 * 8;
    const size_t simg = (size_t)ny * pitch, rimg = (size_t)ny * nx;
    std::vector<double> twd(2 * (size_t)PassTw<Cfg, false, 0>::TOTAL);
    fill_pass_twiddles<Cfg>(twd.data());
    std::vector<float> twf(twd.begin(), twd.end());
    float *tw, *spec_a, *spec_b, *psf_hat, *meas, *est, *norm;
    unsigned long long* stamps;
    CHECK(hipMalloc(&tw, twf.size() * 4));
    CHECK(hipMemcpy(tw, twf.data(), twf.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&spec_a, B * simg * 8 + RL_STREAM_SLACK));
    CHECK(hipMalloc(&spec_b, B * simg * 8 + RL_STREAM_SLACK));
    CHECK(hipMalloc(&psf_hat, (size_t)kx * L * 8));
    CHECK(hipMalloc(&meas, B * rimg * 4 + RL_STREAM_SLACK));
    CHECK(hipMalloc(&est, B * rimg * 4 + RL_STREAM_SLACK));
    CHECK(hipMalloc(&norm, rimg * 4 + RL_STREAM_SLACK));
    {
        std::vector<float> r(B * simg * 2);
        for (auto& x : r) x = 0.5f + (float)rand() / RAND_MAX;
        CHECK(hipMemcpy(spec_a, r.data(), r.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(spec_b, r.data(), r.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(psf_hat, r.data(), (size_t)kx * L * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(meas, r.data(), B * rimg * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(est, r.data(), B * rimg * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(norm, r.data(), rimg * 4, hipMemcpyHostToDevice));
    }
    constexpr int C = CF::C32, Q = CF::Q32;
    const unsigned gxc = (kx + C - 1) / C, gxr = (ny / 2 + Q - 1) / Q;
    const size_t col_waves = (size_t)gxc * B * C, row_waves = (size_t)gxr * B * Q;
    CHECK(hipMalloc(&stamps, std::max(col_waves, row_waves) * NS * 8));
    std::vector<unsigned long long> h(std::max(col_waves, row_waves) * NS);
    ColParams<float> cp;
    cp.in = (const cx<float>*)spec_a; cp.out = (cx<float>*)spec_b; cp.psf_hat = (const cx<float>*)psf_hat;
    cp.tw = (const cx<float>*)tw; cp.ny = ny; cp.kx = kx; cp.pitch = pitch; cp.V = 1; cp.in_sb = 1; cp.in_sv = 0;
    cp.mode = COL_PER_IMAGE; cp.order = 1;
    RowParams<float> rp;
    rp.spec_in = (const cx<float>*)spec_b; rp.spec_out = (cx<float>*)spec_a; rp.src = meas; rp.dst = est; rp.norm = norm;
    rp.scale = nullptr; rp.tw = (const cx<float>*)tw; rp.ny = ny; rp.nx = nx; rp.pitch = pitch; rp.V = 1;
    const size_t ldc = (size_t)C * LdsSlots<Cfg>::value * 8, ldr = (size_t)Q * LdsSlots<Cfg>::value * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char* name, size_t waves, int ns) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(stamps, 0, waves * NS * 8));
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        CHECK(hipMemcpy(h.data(), stamps, waves * NS * 8, hipMemcpyDeviceToHost));
        report(name, h, waves, ns, ms);
    };
    printf("B = %d frames, L = %d, C = %d, Q = %d\n", B, L, C, Q);
    timeit([&] { k_col<StampSyncLazy, C><<<dim3(gxc, B), 64 * C, ldc>>>(cp, stamps); }, "colconv (lazy stamps)", col_waves, 8);
    timeit([&] { k_col<StampSync, C><<<dim3(gxc, B), 64 * C, ldc>>>(cp, stamps); }, "colconv (draining stamps)", col_waves, 8);
    timeit([&] { k_row<StampSyncLazy, Q, ROW_RATIO><<<dim3(gxr, B), 64 * Q, ldr>>>(rp, stamps); }, "row RATIO (lazy)", row_waves, 6);
    timeit([&] { k_row<StampSync, Q, ROW_RATIO><<<dim3(gxr, B), 64 * Q, ldr>>>(rp, stamps); }, "row RATIO (draining)", row_waves, 6);
    timeit([&] { k_row<StampSyncLazy, Q, ROW_UPDATE><<<dim3(gxr, B), 64 * Q, ldr>>>(rp, stamps); }, "row UPDATE (lazy)", row_waves, 6);
    timeit([&] { k_row<StampSync, Q, ROW_UPDATE><<<dim3(gxr, B), 64 * Q, ldr>>>(rp, stamps); }, "row UPDATE (draining)", row_waves, 6);
    // streaming kernels: the stamps of each wave's LAST item survive
    cp.images = B; rp.frames = B;
    constexpr int QS = PROBE_QS;
    const size_t ldcs = ((size_t)C * LdsSlots<Cfg>::value + StreamTw<Cfg>::COUNT) * 8, ldrs = ((size_t)QS * LdsSlots<Cfg>::value + StreamTw<Cfg>::COUNT) * 8;
    auto plain = [&](auto launch, const char* name) {
        float ms = 0, best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-28s %.3f ms (no stamps, best of 4)\n", name, best);
    };
    {
        const int n = resident(k_cols<PlainSync, C>, 64 * C, ldcs);
        plain([&] { k_cols<PlainSync, C><<<n, 64 * C, ldcs>>>(cp, stamps); }, "colstream");
        const int n2 = resident(k_rows<PlainSync, QS, ROW_RATIO>, 64 * QS, ldrs);
        plain([&] { k_rows<PlainSync, QS, ROW_RATIO><<<n2, 64 * QS, ldrs>>>(rp, stamps); }, "rowstream RATIO");
        const int n3 = resident(k_rows<PlainSync, QS, ROW_UPDATE>, 64 * QS, ldrs);
        plain([&] { k_rows<PlainSync, QS, ROW_UPDATE><<<n3, 64 * QS, ldrs>>>(rp, stamps); }, "rowstream UPDATE");
    }
    const int nc = resident(k_cols<StampSyncLazy, C>, 64 * C, ldcs);
    timeit([&] { k_cols<StampSyncLazy, C><<<nc, 64 * C, ldcs>>>(cp, stamps); }, "colstream (lazy)", (size_t)nc * C, 8);
    const int nr = resident(k_rows<StampSyncLazy, QS, ROW_RATIO>, 64 * QS, ldrs);
    timeit([&] { k_rows<StampSyncLazy, QS, ROW_RATIO><<<nr, 64 * QS, ldrs>>>(rp, stamps); }, "rowstream RATIO (lazy)", (size_t)nr * QS, 6);
    const int nu = resident(k_rows<StampSyncLazy, QS, ROW_UPDATE>, 64 * QS, ldrs);
    timeit([&] { k_rows<StampSyncLazy, QS, ROW_UPDATE><<<nu, 64 * QS, ldrs>>>(rp, stamps); }, "rowstream UPDATE (lazy)", (size_t)nu * QS, 6);
    return 0;
}
