// valu_probe.hip -- developer tool: issue rate of plain vs packed f32 VALU ops with several waves
// per SIMD (does v_pk_add_f32 / v_pk_fma_f32 retire two results per lane in the time of one?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));

// Round 4: every block also stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop, so the probe reports
// cycles per instruction in REAL shader cycles and the clock the chip held -- the "2.7-3.1 cycles at 2.4 GHz" of rounds 1-3
// were wall time x an assumed 2.4 GHz.  Modes 6-9: the kernels' own mix (dependent FMA chains, v_cndmask, permlane swaps, DPP adds).
__device__ unsigned long long g_stamps[4 * 4096];
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f, d = 0.5f;
    const float2v pc = {1.0001f, 0.9999f}, pd = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (MODE == 1) { // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
        } else if (MODE == 2) { // 8 independent v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
        } else if (MODE == 3) { // 8 independent v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pd));
        } else if (MODE == 4) { // 8 independent v_add_u32 (integer / address arithmetic)
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
        } else if (MODE == 6) { // ONE dependent chain of 8 v_fma_f32 (what a wave alone can issue)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(c), "v"(d));
        } else if (MODE == 7) { // 8 independent v_cndmask_b32 (VOP3, SGPR-pair condition)
            asm volatile("v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "vcc");
        } else if (MODE == 8) { // 4 v_permlane32_swap + 4 v_permlane16_swap on independent pairs
            asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 9) { // 8 v_add_f32 with a DPP row_ror:8 operand
            asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %3, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %5, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %6, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %7, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %0, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 10) { // 8 v_fma_f32 with a 32-bit literal (v_fmamk_f32) -- the constant-twiddle products
            asm volatile("v_fmamk_f32 %0, %0, 0x3f5db3d7, %8\n v_fmamk_f32 %1, %1, 0x3f5db3d7, %8\n v_fmamk_f32 %2, %2, 0x3f5db3d7, %8\n v_fmamk_f32 %3, %3, 0x3f5db3d7, %8\n"
                         "v_fmamk_f32 %4, %4, 0x3f5db3d7, %8\n v_fmamk_f32 %5, %5, 0x3f5db3d7, %8\n v_fmamk_f32 %6, %6, 0x3f5db3d7, %8\n v_fmamk_f32 %7, %7, 0x3f5db3d7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
        } else {                // packed add with a swapped, half-negated operand: a + i*b for complex pairs
            asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
                         "v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
                         "v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pd));
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 4096) { g_stamps[2 * blockIdx.x] = c1 - c0; g_stamps[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 1024 * 64 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32", "v_add_u32", "v_pk_add_f32 op_sel/neg",
                           "v_fma_f32 dependent chain", "v_cndmask_b32 (+1 v_cmp per 7)", "v_permlane32/16_swap", "v_add_f32_dpp row_ror:8", "v_fmamk_f32 (literal)"};
    for (int wgs_per_cu : {1, 2, 4, 6}) {           // 4 waves per WG -> 1, 2, 4, 6 waves per SIMD
        printf("%d wave(s) per SIMD:\n", wgs_per_cu);
        for (int mode = 0; mode < 11; ++mode) {
            float ms = 0;
            const int grid = 256 * wgs_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (mode) {
                    case 0: k<0><<<grid, 256>>>(out, iters); break;
                    case 1: k<1><<<grid, 256>>>(out, iters); break;
                    case 2: k<2><<<grid, 256>>>(out, iters); break;
                    case 3: k<3><<<grid, 256>>>(out, iters); break;
                    case 4: k<4><<<grid, 256>>>(out, iters); break;
                    case 5: k<5><<<grid, 256>>>(out, iters); break;
                    case 6: k<6><<<grid, 256>>>(out, iters); break;
                    case 7: k<7><<<grid, 256>>>(out, iters); break;
                    case 8: k<8><<<grid, 256>>>(out, iters); break;
                    case 9: k<9><<<grid, 256>>>(out, iters); break;
                    default: k<10><<<grid, 256>>>(out, iters); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double instr = (double)iters * 8 * wgs_per_cu;   // wave-instructions per SIMD
            static unsigned long long st[2 * 4096];
            CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
            double cyc = 0, rt = 0;
            const int nb = grid < 4096 ? grid : 4096;
            for (int b = 0; b < nb; ++b) { cyc += (double)st[2 * b]; rt += (double)st[2 * b + 1]; }
            cyc /= nb; rt /= nb;
            const double ghz = cyc / (rt * 10.0);     // s_memrealtime ticks at 100 MHz: 10 ns each
            printf("   %-32s %.3f ms  %.2f ns per wave-instruction per SIMD = %.2f shader cycles (in-kernel stamps: %.0f cycles per block, clock %.2f GHz)\n",
                   names[mode], ms, ms * 1e6 / instr, cyc / instr, cyc, ghz);
        }
    }
    return 0;
}
