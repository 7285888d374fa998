// kernel_table.hpp -- type-erased launch table, one per supported FFT length.
// Each table is produced by compiling fft_kernels.hip with -DRL_CFG_L=<L>.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace rl {

enum DType { DT_F32 = 0, DT_F64 = 1 };

// Timing events for the NEXT kernel launch of this thread (rl_deconv_time_cycle): the launch helpers of
// fft_kernels.hip pass them to hipExtLaunchKernelGGL, which stamps the kernel's begin and end.
struct LaunchTiming {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchTiming& launch_timing();   // thread local (rlsted.cpp)

struct KernelTable {
    int L;            // transform length
    int T;            // threads per transform
    int C[2];         // spectrum columns per workgroup in the column kernel, per dtype
    int Q[2];         // row pairs per workgroup in the row kernels, per dtype
    // per dtype (the f32 and f64 column kernels of a length may be different kernels):
    int psf_transposed[2];   // column kernel reads psf_hat as [view][Kx][L] (one wave per column)
    int col_multi[2];        // fused multi-view column modes the plan should use: bit 0 COL_H_MULTI, bit 1 COL_HT_SUM
    int tw_count;         // complex entries of the per-pass twiddle table (fft_core.hpp PassTw), row kernels
    void (*fill_tw)(double* out);   // host: writes 2*tw_count doubles (re, im interleaved)
    int tw_count_col[2];  // the same for the column kernels (their geometry may differ: fft_configs.hpp ColCfgFor)
    void (*fill_tw_col[2])(double* out);
    // params: pointer to ColParams<T> / RowParams<T> of the matching dtype
    hipError_t (*launch_col)(int dtype, const void* params, unsigned grid_x, unsigned grid_y, hipStream_t s);
    hipError_t (*launch_row)(int dtype, int mode, const void* params, unsigned grid_x, unsigned grid_y, hipStream_t s);
    // one-time attribute setup (dynamic LDS above the default limit)
    hipError_t (*prepare)(void);
    // Frame-pair row kernels (conv_kernels.hpp rowpair_body: two frames in one complex image, spectra [ny][L]; modes
    // ROW_FWD / ROW_RATIO / ROW_UPDATE, one view): grid (ceil(ny / Q), pairs), RowParams::frames = frames covered.
    // nullptr when the length's row transform is not wave private.
    hipError_t (*launch_row_pair)(int dtype, int mode, const void* params, unsigned grid_y_pairs, hipStream_t s);
    // The split column pass of the long f32 transforms (conv_kernels.hpp COL_SPLIT_FWD / COL_SPLIT_INV / COL_SPLIT_INV_SUM through
    // launch_col): complex elements of one C-column tile's spectra in register-slot order -- an image's parked spectra take
    // ceil(kx / C[0]) * split_tile_elems -- or 0 where the length has no such kernels.
    size_t split_tile_elems;
};

const KernelTable* table_64();
const KernelTable* table_192();
const KernelTable* table_256();
const KernelTable* table_576();
const KernelTable* table_1152();
const KernelTable* table_2304();
const KernelTable* table_4608();

}  // namespace rl
