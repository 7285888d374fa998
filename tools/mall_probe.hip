// mall_probe.hip -- developer tool: in-place streaming update (read 16 B, write 16 B per lane and
// step) over working sets from 16 MiB to 2 GiB, repeated: does the 256 MiB Infinity Cache raise
// the rate for working sets that fit, and by how much?  Also a read-only and a write-only pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_rw(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i];
        v.x += 1.0f; v.y *= 0.5f; v.z -= 1.0f; v.w += v.x;
        p[i] = v;
    }
}
__global__ void k_r(const float4* p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_w(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

int main() {
    const size_t maxb = (size_t)2 << 30;
    float4* buf; float* out;
    CHECK(hipMalloc(&buf, maxb)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 0, maxb));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t sizes_mb[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048};
    printf("%8s %12s %12s %12s   (GB/s of bytes touched; rw counts read + write)\n", "MiB", "read", "write", "rw in place");
    for (size_t mb : sizes_mb) {
        const size_t n = mb * 1048576 / 16;
        const int reps = (int)(8192 / mb) + 4;
        float ms[3];
        for (int which = 0; which < 3; ++which) {
            for (int r = 0; r < 2; ++r) {   // warm
                if (which == 0) k_r<<<4096, 256>>>(buf, n, out); else if (which == 1) k_w<<<4096, 256>>>(buf, n); else k_rw<<<4096, 256>>>(buf, n);
            }
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) {
                if (which == 0) k_r<<<4096, 256>>>(buf, n, out); else if (which == 1) k_w<<<4096, 256>>>(buf, n); else k_rw<<<4096, 256>>>(buf, n);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms[which], e0, e1));
            ms[which] /= reps;
        }
        const double gb = mb * 1048576.0 / 1e9;
        printf("%8zu %12.0f %12.0f %12.0f\n", mb, gb / (ms[0] * 1e-3), gb / (ms[1] * 1e-3), 2 * gb / (ms[2] * 1e-3));
    }
    return 0;
}
