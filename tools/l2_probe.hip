// l2_probe.hip -- developer tool: does data written by one kernel stay in the writing XCD's L2 for the
// next kernel?  Kernel W writes a buffer, every XCD its own contiguous share (workgroup b -> XCD b % 8,
// observed placement); kernel R reads it back with the same share per XCD, or with the shares rotated
// by one XCD.  If L2 contents survive the kernel boundary, "same" is served from L2 and is faster.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// workgroup b of G: XCD slot x = b % 8 (+rot), index i = b / 8 of G/8; share = n/8 float4 per XCD
__global__ void k_write(float4* p, size_t n, int rot) {
    const size_t share = n / 8, per = share / (gridDim.x / 8);
    const int x = (blockIdx.x + rot) % 8, i = blockIdx.x / 8;
    float4* q = p + x * share + (size_t)i * per;
    for (size_t k = threadIdx.x; k < per; k += blockDim.x) q[k] = make_float4(1.f, 2.f, 3.f, (float)k);
}
__global__ void k_read(const float4* p, size_t n, int rot, float* out) {
    const size_t share = n / 8, per = share / (gridDim.x / 8);
    const int x = (blockIdx.x + rot) % 8, i = blockIdx.x / 8;
    const float4* q = p + x * share + (size_t)i * per;
    float acc = 0.f;
    for (size_t k = threadIdx.x; k < per; k += blockDim.x) { float4 v = q[k]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1234.5f) out[0] = acc;
}

int main() {
    float4* buf; float* out; float4* other;
    const size_t maxb = (size_t)1 << 30;
    CHECK(hipMalloc(&buf, maxb)); CHECK(hipMalloc(&other, maxb)); CHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%8s %14s %14s %14s   (GB/s of the read kernel)\n", "MiB", "same XCD", "rotated XCD", "after a flush");
    for (size_t mb : {4, 8, 16, 24, 32, 64, 128, 512}) {
        const size_t n = mb * 1048576 / 16;
        const int G = 2048;
        float res[3];
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                k_write<<<G, 256>>>(buf, n, 0);
                if (mode == 2) k_write<<<G, 256>>>(other, maxb / 16, 0);   // 1 GiB of other traffic in between
                CHECK(hipEventRecord(e0));
                k_read<<<G, 256>>>(buf, n, mode == 1 ? 1 : 0, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            res[mode] = mb * 1048576.0 / 1e9 / (best * 1e-3);
        }
        printf("%8zu %14.0f %14.0f %14.0f\n", mb, res[0], res[1], res[2]);
    }
    return 0;
}
