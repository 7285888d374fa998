// rlsted.cpp -- C ABI (include/rlsted.h) over the gfx950 kernels.
// Host orchestration only: buffer ownership, kernel sequencing on one HIP
// stream per context, host<->device staging.  No arithmetic of the hot path
// runs on the host.
#include "../../include/rlsted.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <chrono>
#include <vector>

#include "aux_kernels.hpp"
#include "conv_kernels.hpp"
#include "kernel_table.hpp"
#include "ctx.hpp"
#include "sep_kernels.hpp"

using namespace rl;

namespace rl {
std::string& last_error() {
    thread_local std::string e;
    return e;
}
int fail(int code, const std::string& msg) {
    last_error() = msg;
    return code;
}
LaunchTiming& launch_timing() {
    thread_local LaunchTiming t;
    return t;
}
bool debug_sync() {
    static const bool v = getenv("RLSTED_DEBUG_SYNC") != nullptr;
    return v;
}
}  // namespace rl

namespace {

const KernelTable* table_for(int L) {
    switch (L) {
        case 64: return table_64();
        case 192: return table_192();
        case 256: return table_256();
        case 576: return table_576();
        case 1152: return table_1152();
        case 2304: return table_2304();
        case 4608: return table_4608();
    }
    return nullptr;
}
const int kLengths[] = {64, 192, 256, 576, 1152, 2304, 4608};

size_t esize(int dtype) { return dtype == RL_F32 ? 4 : 8; }

}  // namespace

struct rl_deconv {
    rl_ctx* ctx = nullptr;
    int V = 0, py = 0, px = 0, B = 0, ny = 0, nx = 0, dtype = RL_F32;
    int ly = 0, lx = 0, kx = 0, pitch = 0;
    const KernelTable *ty = nullptr, *tx = nullptr;
    void *twy = nullptr, *twx = nullptr;
    // device buffers (element type = dtype)
    void* psf_hat = nullptr;   // [V][ly][pitch] complex
    void* psf_hat_re = nullptr;   // real parts, when the PSF spectrum is real (point-symmetric PSFs) and the column
                                  // transform is wave private: the column kernels then multiply by a real array
    double psf_hat_imag_ratio = 0;   // max |im| / max |z| of the PSF spectrum
    void* spec_a = nullptr;    // [B] spectrum images (layout: conv_kernels.hpp spec_off)
    void* spec_b = nullptr;    // [B*V] spectrum images
    void* spec_x = nullptr;    // [B*V] column spectra in register-slot order between the halves of the split column pass (col_split() plans)
    void* spec_ones = nullptr; // [V] column-transformed spectra of H(estimate = 1): the same for every frame (ref:522)
    // storage-precision study builds (conv_kernels.hpp RL_SPEC_QUANT): powers of two that bring the DC term of an
    // estimate-type / ratio-type spectrum to 2^14 (RLSTED_Q_EXP_EST / RLSTED_Q_EXP_RATIO = log2 of the DC bound)
    float q_est = 1.0f, q_ratio = 1.0f;
    bool ones_shortcut = true; // first iteration reads spec_ones instead of transforming a frame of ones (RLSTED_ONES_SHORTCUT=0: off)
    // f32 plans (conv_kernels.hpp rl_ratio): the normaliser H_t(ones) from the PSFs' integral images instead of the f32 transform
    // path (RLSTED_EXACT_NORM=0: off), and -- non-negative PSFs -- the second half of every iteration on `ratio - 1`
    // (RLSTED_SUB_ONE=0: off).  Both shrink f32 rounding error, neither changes the arithmetic in exact terms.
    bool exact_norm = false, sub_one = false;
    // `ratio - 1` clamps the SUM of the views' back-projections where the reference clamps each view's (ref:587).  The two agree
    // whenever no view's term is negative -- always for one view, and for several as long as the measurement has no negative
    // pixel (PSFs >= 0 is a condition of sub_one).  A multi-view measurement WITH negative pixels (background-subtracted data
    // through rl_deconv_set_measurement) therefore runs the plain arithmetic with the per-view clamp; so does RLSTED_FUSE_VIEWS=0,
    // the switch that asks for the reference's per-view clamp.
    bool meas_negative = false;
    // lanes of the ROW_RATIO launches that met a prediction H(est) <= 0 inside the image (conv_kernels.hpp rl_ratio: such a pixel is
    // neutral); device counter, read by rl_deconv_unresolved.  Zero on data whose predictions the plan's arithmetic resolves.
    unsigned long long* unresolved = nullptr;
    bool in_rl_loop = false;   // set by iterate_chunk: only there do the H_t column launches carry `ratio - 1` (rl_adjoint's input is an image)
    bool sub() const { return sub_one && !(V > 1 && meas_negative); }
    void* obj = nullptr;       // [B][ny][nx]
    void* noiseless = nullptr; // [B*V][ny][nx]
    void* meas = nullptr;      // [B*V][ny][nx]
    void* est = nullptr;       // [B][ny][nx]
    void* norm = nullptr;      // [ny][nx]
    void* scratch = nullptr;   // [B*V][ny][nx] staging for rl_forward / rl_adjoint
    size_t bytes = 0;
    bool have_obj = false, have_meas = false;
    // H_t views summed before the inverse transforms (one clamp of the sum instead of one per
    // view, ref:587): default for f32 plans, off for f64 (faithful); RLSTED_FUSE_VIEWS=0/1 overrides
    bool fuse_views = false;
    // Slices of the batch are independent: they are iterated on `lanes` HIP streams at once so that
    // the tail of one slice's kernel (the last, partly filled round of workgroups) overlaps another
    // slice's kernels.  RLSTED_LANES=1: one slice after the other on the context's stream.
    static constexpr int kMaxLanes = 4;
    int lanes = 2;
    hipStream_t lane_stream[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_done[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr;
    hipStream_t active = nullptr;                     // stream the kernel launch helpers use
    bool defer_join = false, lanes_open = false;      // rl_deconv_bench_cycles: no lane join between its cycles
    void* slice_ws = nullptr;                         // per-slice Poisson work lists of run_cycle()
    void *key_seeds = nullptr, *key_ids = nullptr;    // per-frame Philox keys of rl_deconv_simulate_keyed
    size_t slice_ws_bytes = 0, slice_ws_stride = 0;
    // ---- rl_batch_submit: the tasks of a chunk -- objects, brightness targets, Philox keys -- are staged in one page-locked
    // block, uploaded on a copy stream of the plan's own and consumed on the context's stream; two blocks, so that chunk i + 1
    // is staged and uploaded while chunk i iterates.  Block layout (host and device): header -- [B] float64 targets, [B] uint64
    // seeds, [B] uint32 image ids, [B] uint32 object index -- then the chunk's DISTINCT objects, [<= B][n_img] float64 (tasks that
    // share an object pointer -- a sweep's seeds -- are staged and uploaded once; only the used prefix of the block crosses PCIe);
    // the device block is followed by the objects' float64 sums and their scratch (aux_sums_elems(B)).
    struct BatchSlot {
        char* host = nullptr;
        char* dev = nullptr;
        hipEvent_t uploaded = nullptr, freed = nullptr;
        bool used = false;
    };
    BatchSlot bslot[2];
    void* batch_out = nullptr;       // rl_batch_run's result buffer (grow only)
    size_t batch_out_bytes = 0;
    hipStream_t copy_stream = nullptr;
    unsigned long batch_chunks = 0;
    const unsigned long long* run_key_seeds = nullptr;   // keyed Poisson draws of run_slices (device, [B]); nullptr: one seed
    const unsigned* run_key_ids = nullptr;
    size_t slot_objects_bytes() const { return (size_t)B * n_img() * sizeof(double); }
    size_t slot_header_bytes() const { return ((size_t)B * (8 + 8 + 4 + 4) + 15) / 16 * 16; }
    size_t slot_host_bytes() const { return slot_header_bytes() + slot_objects_bytes(); }
    bool batch_slots_ready = false;
    int ensure_batch_slots() {
        if (batch_slots_ready) return RL_OK;
        // (a call that failed half way is taken up where it stopped: every resource is created once, the destructor frees what exists)
        if (!copy_stream) HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        for (BatchSlot& sl : bslot) {
            if (!sl.host) HIP_TRY(hipHostMalloc((void**)&sl.host, slot_host_bytes(), hipHostMallocDefault));
            if (!sl.dev) {
                HIP_TRY(hipMalloc((void**)&sl.dev, slot_host_bytes() + 8 + aux_sums_elems((size_t)B) * sizeof(double)));
                bytes += slot_host_bytes() + 8 + aux_sums_elems((size_t)B) * sizeof(double);
            }
            if (!sl.uploaded) HIP_TRY(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
            if (!sl.freed) HIP_TRY(hipEventCreateWithFlags(&sl.freed, hipEventDisableTiming));
        }
        batch_slots_ready = true;
        return RL_OK;
    }
    hipStream_t cur() const { return active ? active : ctx->stream; }
    // column kernel work order: images per block of the tile order (fft_kernels.hip k_colconv):
    // 1 image-major ... >= images per launch: tile-major (RLSTED_COL_ORDER).  Measured at 512^2, 32-frame
    // slices: 1: 16.80 k, 2: 16.88 k, 4: 16.98 k, 8: 16.81 k, 32: 16.56 k frames/s.
    int col_order = 4;
    bool inplace = true;     // single-view RL iterations entirely in spec_a (RLSTED_INPLACE=0: spec_a -> spec_b -> spec_a)
    bool est_ready = false;    // est holds a valid estimate
    bool spec_valid = false;   // spec_a holds rowFFT(est)
    long iterations = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_iter_ms = 0, last_sim_ms = 0;

    enum ColKind { COL_H, COL_HT_VIEW, COL_HT_FUSED };
    // ---- frame pairs (conv_kernels.hpp rowpair_body; RLSTED_PAIR): two frames ride through one complex image, the
    // Richardson-Lucy loop of a single-view plan then runs on spectra [pairs][ny][lx] -- no Hermitian packing /
    // splitting around the row transforms.  The simulation and the H / H_t calls keep the per-frame layout.
    bool pair = false;          // the loop that runs: pair_layout && the batch's partners are of comparable brightness (choose_loop)
    bool pair_layout = false;   // the plan holds the pair buffers (psf_hat_pair, spec_ones_pair)
    // A pair's two frames share one complex transform, so f32 rounding error scales with the BRIGHTER partner: a dim frame
    // next to one 1e5 times brighter would carry ~1e5 times its own error through H.  Frames are paired only while every
    // pair's levels (sums of the object / measurement images) are within kPairMaxRatio of each other
    // (RLSTED_PAIR_MAX_RATIO); otherwise the plan runs its per-frame loop, which every pair plan also holds.
    double pair_max_ratio = 4.0;
    std::vector<double> obj_level, meas_level;   // per frame (host): sum of the object / of the measurement over its views
    bool levels_ok(const std::vector<double>& lv) const {
        if ((int)lv.size() != B) return true;    // nothing known yet
        for (int f = 0; f + 1 < B; f += 2) {
            const double lo = std::min(lv[f], lv[f + 1]), hi = std::max(lv[f], lv[f + 1]);
            if (lo == 0.0 && hi == 0.0) continue;   // two empty frames
            if (!(lo > 0.0) || !std::isfinite(hi) || hi > pair_max_ratio * lo) return false;
        }
        return true;
    }
    void choose_loop(const std::vector<double>& lv) {
        const bool want = pair_layout && levels_ok(lv);
        if (want != pair) {
            pair = want;
            spec_valid = false;   // spec_a holds the other layout's spectra
        }
    }
    // The measurement buffer was handed out (rl_deconv_device_ptr which = 1) and may have been written on the device: the per-frame
    // sums are recomputed there before the next run decides between the pair loop and the per-frame loop.
    bool meas_external = false;
    int refresh_meas_levels() {
        if (!meas_external) return RL_OK;
        RL_TRY(ensure_stage());
        std::vector<double> sums((size_t)B * V);
        HIP_TRY(aux_image_sums(dtype, meas, n_img(), (size_t)B * V, stage_sums, ctx->stream));
        HIP_TRY(hipMemcpyAsync(sums.data(), stage_sums, sums.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        meas_level.assign((size_t)B, 0.0);
        for (size_t i = 0; i < sums.size(); ++i) meas_level[i / V] += sums[i];
        choose_loop(meas_level);
        RL_TRY(scan_meas_negative());
        meas_external = false;
        return RL_OK;
    }
    // does the measurement on the device hold a negative pixel?  (multi-view plans only: see sub())
    int scan_meas_negative() {
        meas_negative = false;
        if (V < 2 || !sub_one) return RL_OK;
        RL_TRY(ensure_stage());
        int flag = 0;
        HIP_TRY(aux_any_negative(dtype, meas, (size_t)B * V * n_img(), (int*)stage_aux, ctx->stream));
        HIP_TRY(hipMemcpyAsync(&flag, stage_aux, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        meas_negative = flag != 0;
        return RL_OK;
    }
    bool keep_last_spectrum = false;   // RLSTED_KEEP_LAST_SPECTRUM=1: every iteration ends with rowFFT(estimate)
    bool drop_last_spectrum = false;   // set by run_slices for the last iteration of a long run: ROW_UPDATE skips rowFFT(estimate)
    void *psf_hat_pair = nullptr, *psf_hat_pair_re = nullptr;   // psf_hat at full width: [lx][ly] (transposed layout)
    void* spec_ones_pair = nullptr;                              // column-transformed spectrum of a pair of ones frames
    // Row pitch of a pair spectrum (complex elements): lx, or lx + 32 for the long rows (lx >= 1152).  A pitch of exactly lx
    // makes the row stride (lx * 8 bytes in f32: 18432 / 36864) an EVEN multiple of 256 bytes, and the column kernels' tile
    // rows -- one 64-byte segment per row -- then fall on a fraction of the memory channels; + 256 bytes makes it an odd
    // multiple (RLSTED_PAIR_PAD columns).  Worth 2-4 % on the long rows, costs 3 % at lx = 576 (4608-byte rows: left alone).
    int pair_pitch = 0;
    size_t n_spec_pair() const { return spec_image_elems(ny, pair_pitch); }
    void* pair_spec(int f0) const { return (char*)spec_a + (size_t)(f0 / 2) * n_spec_pair() * 2 * esize(dtype); }
    // kind: COL_H (pair spectrum -> V images; in place when V == 1), COL_HT_VIEW (V == 1, in place) or COL_HT_FUSED
    // (V images summed in the Fourier domain -> pair spectrum)
    template <typename T>
    int col_pair_t(const void* in, void* out, int pairs, ColKind kind) {
        ColParams<T> p;
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.psf_hat = (const cx<T>*)psf_hat_pair;
        p.psf_hat_re = (const T*)psf_hat_pair_re;
        p.qscale = kind == COL_H ? q_est : q_ratio;
        p.tw = (const cx<T>*)twy;
        p.ny = ny; p.kx = lx; p.pitch = pair_pitch; p.V = V;
        p.mode = V == 1 ? COL_PER_IMAGE : (kind == COL_H ? COL_H_MULTI : COL_HT_SUM);
        p.in_sb = 1; p.in_sv = 0;
        p.images = pairs; p.order = col_order;
        p.residual = (kind != COL_H && sub() && in_rl_loop) ? 1 : 0;   // H_t inside a `ratio - 1` iteration: the spectrum of a residual
        const int C = ty->C[dtype];
        TimedScope t(this, kind == COL_H ? TK_COL_H : TK_COL_HT);
        HIP_TRY(ty->launch_col(dtype, &p, (unsigned)((lx + C - 1) / C), (unsigned)pairs, cur()));
        return RL_OK;
    }
    int col_pair(const void* in, void* out, int pairs, ColKind kind) {
        const size_t sp = n_spec_pair() * 2 * esize(dtype);
        const size_t in_per = kind == COL_H ? 1 : (size_t)V, out_per = kind == COL_H ? (size_t)V : 1;
        for (int p0 = 0; p0 < pairs; p0 += kMaxGridY) {
            const int np = std::min((int)kMaxGridY, pairs - p0);
            const void* i = (const char*)in + (size_t)p0 * in_per * sp;
            void* o = (char*)out + (size_t)p0 * out_per * sp;
            RL_TRY(dtype == RL_F32 ? col_pair_t<float>(i, o, np, kind) : col_pair_t<double>(i, o, np, kind));
        }
        return RL_OK;
    }
    int col_pair(void* io, int pairs, bool h_mode) { return col_pair(io, io, pairs, h_mode ? COL_H : COL_HT_VIEW); }
    // frames: images covered (even, or the batch's last odd one); spectra and images start at the launch's first pair
    template <typename T>
    int row_pair_t(int mode, int frames, const void* spec_in, void* spec_out, const void* src, void* dst, const void* nrm, int in_mod, int views) {
        RowParams<T> p;
        p.in_mod = in_mod;
        p.sub_one = sub() ? 1 : 0;
        p.unresolved = unresolved;
        p.qscale = mode == ROW_RATIO ? q_ratio : q_est;
        p.spec_in = (const cx<T>*)spec_in;
        p.spec_out = (cx<T>*)spec_out;
        p.src = (const T*)src;
        p.dst = (T*)dst;
        p.norm = (const T*)nrm;
        p.scale = nullptr;
        p.tw = (const cx<T>*)twx;
        p.ny = ny; p.nx = nx; p.pitch = pair_pitch; p.V = views;
        p.frames = frames;
        TimedScope t(this, mode == ROW_RATIO ? TK_RATIO : mode == ROW_UPDATE ? TK_UPDATE : TK_FWD);
        HIP_TRY(tx->launch_row_pair(dtype, mode, &p, (unsigned)((frames + 1) / 2 * views), cur()));
        return RL_OK;
    }
    // views > 1 (ROW_RATIO of a multi-view plan): one launch image per (pair, view); spectra [pair][view], images [frame][view]
    int row_pair(int mode, int frames, const void* spec_in, void* spec_out, const void* src, void* dst, const void* nrm, int in_mod = 0,
                 int views = 1) {
        const size_t sp = n_spec_pair() * 2 * esize(dtype), im = n_img() * esize(dtype);
        const int step = 2 * ((int)kMaxGridY / views);
        for (int f0 = 0; f0 < frames; f0 += step) {
            const int nf = std::min(step, frames - f0);
            const void* si = spec_in ? (const char*)spec_in + (in_mod > 0 ? 0 : (size_t)(f0 / 2) * views * sp) : nullptr;
            void* so = spec_out ? (char*)spec_out + (size_t)(f0 / 2) * views * sp : nullptr;
            const void* sr = src ? (const char*)src + (size_t)f0 * views * im : nullptr;
            void* ds = dst ? (char*)dst + (size_t)f0 * im : nullptr;
            RL_TRY(dtype == RL_F32 ? row_pair_t<float>(mode, nf, si, so, sr, ds, nrm, in_mod, views)
                                   : row_pair_t<double>(mode, nf, si, so, sr, ds, nrm, in_mod, views));
        }
        return RL_OK;
    }

    // ---- separable strategy (sep_kernels.hip): every view rank 1 (p = u v^T) and small -> direct row + column stencils
    // instead of the FFT path.  RLSTED_SEP: 0 never, 1 (default) when py + px <= RLSTED_SEP_MAX_TAPS (16: the measured
    // crossover, profiles/r02/separable_vs_fft.json -- the FFT path's cost does not depend on the PSF size), 2 whenever rank 1.
    bool sep = false;
    void *sep_u = nullptr, *sep_v = nullptr;   // [V][py], [V][px] in the plan's dtype
    void *sep_uf = nullptr, *sep_vf = nullptr; // flipped, zero padded to multiples of 8: the one-kernel form's taps
    bool sep_one = false;                      // both passes in one kernel (RLSTED_SEP_ONE, see deconv_build)
    bool sep_direct = false;                   // the one-kernel form as a direct 2-D stencil: PSFs that are not rank 1 (RLSTED_DIRECT)
    int sep2d_(int mode, const void* in, const void* aux, const void* nrm, void* dst, int frames) {
        const bool multi = mode == SEP_SUM_ || mode == SEP_UPDATE_;
        for (int f0 = 0; f0 < frames; f0 += 65535) {
            const int n = std::min(65535, frames - f0);
            const size_t img = n_img() * esize(dtype), in_off = (size_t)f0 * (multi ? V : 1) * img, out_off = (size_t)f0 * (multi ? 1 : V) * img;
            HIP_TRY(sep2d(dtype, mode, (const char*)in + in_off, sep_uf, sep_vf, aux ? (const char*)aux + out_off : nullptr, nrm,
                          (char*)dst + out_off, n, ny, nx, py, px, V, cur()));
        }
        return RL_OK;
    }
    void* sep_tmp() const { return spec_b; }    // row-pass results [B*V][ny][nx] (the spectrum buffer is free in this mode)
    int sep_rows_(const void* in, void* out, int images, int in_div) {
        for (int i0 = 0; i0 < images; i0 += 65535 / V * V) {   // grid.z pieces of whole frames
            const int n = std::min(65535 / V * V, images - i0);
            HIP_TRY(sep_rows(dtype, (const char*)in + (size_t)(i0 / in_div) * n_img() * esize(dtype),
                             (char*)out + (size_t)i0 * n_img() * esize(dtype), sep_v, n, ny, nx, px, V, in_div, cur()));
        }
        return RL_OK;
    }
    // mode, tmp [count*(multi ? V : 1)] -> dst [count]; aux: the measurement for SEP_RATIO_
    int sep_cols_(int mode, const void* tmp, const void* aux, const void* nrm, void* dst, int count) {
        const bool multi = mode == SEP_SUM_ || mode == SEP_UPDATE_;
        const int step = multi ? 65535 : 65535 / V * V;
        for (int i0 = 0; i0 < count; i0 += step) {
            const int n = std::min(step, count - i0);
            const size_t in_off = (size_t)i0 * (multi ? V : 1) * n_img() * esize(dtype), out_off = (size_t)i0 * n_img() * esize(dtype);
            HIP_TRY(sep_cols(dtype, mode, (const char*)tmp + in_off, sep_u, aux ? (const char*)aux + out_off : nullptr, nrm,
                             (char*)dst + out_off, n, ny, nx, py, V, cur()));
        }
        return RL_OK;
    }
    // H of nf frames: x [nf] -> out [nf*V] (clamped)
    int sep_forward(const void* x, void* out, int nf) {
        if (sep_one) return sep2d_(SEP_STORE_, x, nullptr, nullptr, out, nf);
        RL_TRY(sep_rows_(x, sep_tmp(), nf * V, V));
        return sep_cols_(SEP_STORE_, sep_tmp(), nullptr, nullptr, out, nf * V);
    }
    int sep_iterate(int f0, int nf) {   // ref:520-531
        void* e = off(est, (size_t)f0 * n_img());
        void* ratio = off(scratch, (size_t)f0 * V * n_img());
        void* tmp = off(sep_tmp(), (size_t)f0 * V * n_img());
        if (sep_one) {
            RL_TRY(sep2d_(SEP_RATIO_, e, off(meas, (size_t)f0 * V * n_img()), nullptr, ratio, nf));
            return sep2d_(SEP_UPDATE_, ratio, nullptr, norm, e, nf);
        }
        RL_TRY(sep_rows_(e, tmp, nf * V, V));
        RL_TRY(sep_cols_(SEP_RATIO_, tmp, off(meas, (size_t)f0 * V * n_img()), nullptr, ratio, nf * V));
        RL_TRY(sep_rows_(ratio, tmp, nf * V, 1));
        return sep_cols_(SEP_UPDATE_, tmp, nullptr, norm, e, nf);
    }

    // ---- in-situ kernel timing (rl_deconv_time_cycle): an event pair around every launch of one whole
    // cycle, on the stream the launch goes to, with the slice streams overlapping as in production
    enum TimedKind { TK_COL_H = 0, TK_RATIO, TK_COL_HT, TK_UPDATE, TK_FWD, TK_INV, TK_POISSON, TK_COUNT };
    struct TimedLaunch { int kind; hipEvent_t a, b; bool cont; };   // cont: second launch of one pass (its time adds to the pass)
    bool timing = false;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    hipEvent_t pool_event() {
        if (events_used == event_pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            event_pool.push_back(e);
        }
        return event_pool[events_used++];
    }
    // usage: { TimedScope t(this, kind); launch; }.  ext: the launch goes through fft_kernels.hip's
    // rl_launch, which stamps the kernel's own begin / end on the events (what a kernel trace shows);
    // otherwise the events are recorded on the stream around the launch(es).
    struct TimedScope {
        rl_deconv* h; hipEvent_t a = nullptr, b = nullptr; int kind; bool ext; bool cont = false;
        TimedScope(rl_deconv* h_, int kind_, bool ext_ = true, bool cont_ = false) : h(h_), kind(kind_), ext(ext_), cont(cont_) {
            if (!h->timing || !(a = h->pool_event()) || !(b = h->pool_event())) return;
            if (ext) {
                launch_timing().start = a;
                launch_timing().stop = b;
            } else {
                (void)hipEventRecord(a, h->cur());
            }
        }
        ~TimedScope() {
            if (!a || !b) return;
            if (ext) {
                const bool used = launch_timing().start == nullptr;   // rl_launch consumed them
                launch_timing().start = launch_timing().stop = nullptr;
                if (!used) return;
            } else {
                (void)hipEventRecord(b, h->cur());
            }
            h->timed.push_back({kind, a, b, cont});
        }
    };

    size_t n_img() const { return (size_t)ny * nx; }
    size_t n_spec() const { return spec_image_elems(ny, pitch); }   // complex elements of one spectrum image

    bool col_multi = true;   // RLSTED_COL_MULTI=0 (A/B knob): V per-image column launches even where the multi-view modes exist
    // KernelTable::col_multi: bit 0 COL_H_MULTI, bit 1 COL_HT_SUM.  wave_private_y(): the Fourier-domain view sum exists (and is wanted)
    bool wave_private_y() const { return col_multi && (ty->col_multi[dtype] & 2) != 0; }
    bool h_multi() const { return col_multi && (ty->col_multi[dtype] & 1) != 0; }
    // The split column pass (conv_kernels.hpp COL_SPLIT_*; f32 multi-view plans on the long column transforms): H transforms a frame's
    // spectrum once for its V views, H_t sums the views' products before one inverse transform (RLSTED_COL_SPLIT=0: A/B knob)
    bool split_wanted = true;
    bool split_ht = true;    // H_t through the split pass (RLSTED_SPLIT_HT=0: V whole-pass launches + the pre-summed update; measured)
    bool split_h = true;     // H through the split pass too (RLSTED_SPLIT_H=0: H_t only -- measured in round 4)
    bool col_split() const { return split_wanted && dtype == RL_F32 && V > 1 && ty->split_tile_elems > 0; }
    size_t n_spec_x() const { return (size_t)((kx + ty->C[RL_F32] - 1) / ty->C[RL_F32]) * ty->split_tile_elems; }
    // one half of the split pass over `images` launch rows
    int col_split_launch(int mode, const void* in, void* out, const void* xs_in, void* xs_out, int images, int in_sb, int in_sv, ColKind kind,
                         bool cont) {
        ColParams<float> p;
        p.in = (const cx<float>*)in;
        p.out = (cx<float>*)out;
        p.xs_in = (const cx<float>*)xs_in;
        p.xs_out = (cx<float>*)xs_out;
        p.psf_hat = (const cx<float>*)psf_hat;
        p.psf_hat_re = (const float*)psf_hat_re;
        p.qscale = kind == COL_H ? q_est : q_ratio;
        p.tw = (const cx<float>*)twy;
        p.ny = ny; p.kx = kx; p.pitch = pitch; p.V = V;
        p.mode = mode;
        p.in_sb = in_sb; p.in_sv = in_sv;
        p.images = images;
        p.order = 1;
        const int C = ty->C[dtype];
        {
            TimedScope t(this, kind == COL_H ? TK_COL_H : TK_COL_HT, true, cont);
            HIP_TRY(ty->launch_col(dtype, &p, (unsigned)((kx + C - 1) / C), (unsigned)images, cur()));
        }
        if (rl::debug_sync()) {
            hipError_t e = hipStreamSynchronize(cur());
            if (e != hipSuccess) return fail(RL_ERR_HIP, "split column kernel mode " + std::to_string(mode) + ": " + hipGetErrorString(e));
        }
        return RL_OK;
    }
    // kind COL_H: `in` = the frames' spectra -> out = frames * V images; COL_HT_FUSED: in = frames * V images -> out = frames
    // xs: this slice's part of spec_x (frames * V images of n_spec_x() elements)
    int col_split_pass(const void* in, void* out, void* xs, int frames, ColKind kind) {
        const size_t sp = n_spec() * 2 * sizeof(float), sx = n_spec_x() * 2 * sizeof(float);
        const int step = std::max(1, kMaxGridY / V);
        for (int f0 = 0; f0 < frames; f0 += step) {
            const int nf = std::min(step, frames - f0);
            void* x = (char*)xs + (size_t)f0 * V * sx;
            if (kind == COL_H) {
                RL_TRY(col_split_launch(COL_SPLIT_FWD, (const char*)in + (size_t)f0 * sp, nullptr, nullptr, x, nf, 1, 0, kind, false));
                RL_TRY(col_split_launch(COL_SPLIT_INV, nullptr, (char*)out + (size_t)f0 * V * sp, x, nullptr, nf * V, 1, 0, kind, true));
            } else {
                RL_TRY(col_split_launch(COL_SPLIT_FWD, (const char*)in + (size_t)f0 * V * sp, nullptr, nullptr, x, nf * V, 1, 0, kind, false));
                RL_TRY(col_split_launch(COL_SPLIT_INV_SUM, nullptr, (char*)out + (size_t)f0 * sp, x, nullptr, nf, 1, 0, kind, true));
            }
        }
        return RL_OK;
    }
    bool psf_transposed() const { return ty->psf_transposed[dtype] != 0; }   // psf_hat is [view][Kx][Ly]
    template <typename T>
    int col_t(const void* in, void* out, int frames, ColKind kind) {
        ColParams<T> p;
        p.in = (const cx<T>*)in;
        p.out = (cx<T>*)out;
        p.psf_hat = (const cx<T>*)psf_hat;
        p.psf_hat_re = (const T*)psf_hat_re;
        p.qscale = kind == COL_H ? q_est : q_ratio;
        p.tw = (const cx<T>*)twy;
        p.ny = ny; p.kx = kx; p.pitch = pitch; p.V = V;
        unsigned gy = (unsigned)(frames * V);
        p.mode = COL_PER_IMAGE;
        p.in_sb = kind == COL_H ? 1 : V;
        p.in_sv = kind == COL_H ? 0 : 1;
        if (V > 1 && kind == COL_H && h_multi()) {
            p.mode = COL_H_MULTI;
            gy = (unsigned)frames;
        } else if (V > 1 && kind == COL_HT_FUSED && wave_private_y()) {
            p.mode = COL_HT_SUM;
            gy = (unsigned)frames;
        } else if (kind == COL_HT_FUSED && V > 1) {
            return fail(RL_ERR_STATE, "internal: fused H_t needs a wave-private column transform");
        }
        const int C = ty->C[dtype];
        const unsigned gx = (unsigned)((kx + C - 1) / C);
        p.images = (int)gy;
        p.order = col_order;
        p.residual = (kind != COL_H && sub() && in_rl_loop) ? 1 : 0;   // H_t inside a `ratio - 1` iteration: the spectra of residuals
        {
            TimedScope t(this, kind == COL_H ? TK_COL_H : TK_COL_HT);
            HIP_TRY(ty->launch_col(dtype, &p, gx, gy, cur()));
        }
        if (rl::debug_sync()) {
            hipError_t e = hipStreamSynchronize(cur());
            if (e != hipSuccess)
                return fail(RL_ERR_HIP, "column kernel L=" + std::to_string(ly) + " grid " + std::to_string(gx) + "x" +
                                            std::to_string(gy) + ": " + hipGetErrorString(e));
        }
        return RL_OK;
    }
    // grid.y carries the image index of a launch: at most kMaxGridY images per launch, larger batches
    // are launched in pieces (whole frames each) with the pointers moved on
    static constexpr int kMaxGridY = 65535;
    int col(const void* in, void* out, int frames, ColKind kind) {
        const size_t sp = n_spec() * 2 * esize(dtype);   // bytes of one spectrum image
        const size_t in_per = kind == COL_H ? 1 : (size_t)V, out_per = kind == COL_HT_FUSED ? 1 : (size_t)V;
        const int step = std::max(1, kMaxGridY / V);
        for (int f0 = 0; f0 < frames; f0 += step) {
            const int nf = std::min(step, frames - f0);
            const void* i = (const char*)in + (size_t)f0 * in_per * sp;
            void* o = (char*)out + (size_t)f0 * out_per * sp;
            RL_TRY(dtype == RL_F32 ? col_t<float>(i, o, nf, kind) : col_t<double>(i, o, nf, kind));
        }
        return RL_OK;
    }
    int col(const void* in, void* out, int frames, bool h_mode) { return col(in, out, frames, h_mode ? COL_H : COL_HT_VIEW); }
    template <typename T>
    int row_t(int mode, unsigned gy, const void* spec_in, void* spec_out, const void* src, void* dst, const void* nrm,
              const void* scale, int views, int in_mod = 0) {
        RowParams<T> p;
        p.in_mod = in_mod;
        p.sub_one = sub() ? 1 : 0;
        p.unresolved = unresolved;
        p.qscale = mode == ROW_RATIO ? q_ratio : q_est;
        p.spec_in = (const cx<T>*)spec_in;
        p.spec_out = (cx<T>*)spec_out;
        p.src = (const T*)src;
        p.dst = (T*)dst;
        p.norm = (const T*)nrm;
        p.scale = (const T*)scale;
        p.tw = (const cx<T>*)twx;
        p.ny = ny; p.nx = nx; p.pitch = pitch; p.V = views;
        const int Q = tx->Q[dtype];
        const unsigned pairs = (unsigned)((ny + 1) / 2);
        p.frames = (int)gy;
        {
            TimedScope t(this, mode == ROW_RATIO ? TK_RATIO : mode == ROW_UPDATE ? TK_UPDATE : mode == ROW_FWD ? TK_FWD : TK_INV);
            HIP_TRY(tx->launch_row(dtype, mode, &p, (pairs + Q - 1) / Q, gy, cur()));
        }
        if (rl::debug_sync()) {
            hipError_t e = hipStreamSynchronize(cur());
            if (e != hipSuccess)
                return fail(RL_ERR_HIP, "row kernel mode " + std::to_string(mode) + " L=" + std::to_string(lx) + " grid " +
                                            std::to_string((pairs + Q - 1) / Q) + "x" + std::to_string(gy) + ": " +
                                            hipGetErrorString(e));
        }
        return RL_OK;
    }
    int row(int mode, unsigned gy, const void* spec_in, void* spec_out, const void* src, void* dst, const void* nrm,
            const void* scale = nullptr, int views = -1, int in_mod = 0) {
        if (views < 0) views = V;
        const size_t sp = n_spec() * 2 * esize(dtype), im = n_img() * esize(dtype);
        const bool multi = mode == ROW_UPDATE || mode == ROW_ADJ;   // `views` input spectra per image
        const unsigned piece = in_mod > 0 ? (unsigned)(kMaxGridY / in_mod * in_mod) : (unsigned)kMaxGridY;
        for (unsigned g0 = 0; g0 < gy; g0 += piece) {
            const unsigned ng = std::min(piece, gy - g0);
            // (kMaxGridY is a multiple of every in_mod in use only by accident: a shared input is not moved on,
            // and the image index restarts at 0 in each piece -- so pieces must start on a multiple of in_mod)
            const void* si = spec_in ? (const char*)spec_in + (in_mod > 0 ? 0 : (size_t)g0 * (multi ? (size_t)views : 1) * sp) : nullptr;
            void* so = spec_out ? (char*)spec_out + (size_t)g0 * sp : nullptr;
            const void* sr = src ? (const char*)src + (size_t)g0 * im : nullptr;
            void* ds = dst ? (char*)dst + (size_t)g0 * im : nullptr;
            const void* sc = scale ? (const char*)scale + (size_t)g0 * esize(dtype) : nullptr;
            RL_TRY(dtype == RL_F32 ? row_t<float>(mode, ng, si, so, sr, ds, nrm, sc, views, in_mod)
                                   : row_t<double>(mode, ng, si, so, sr, ds, nrm, sc, views, in_mod));
        }
        return RL_OK;
    }

    // Host float64 <-> plan dtype.  The conversion (and the brightness scaling) runs on the
    // device: the host array goes over PCIe as it is, in slices of at most kStageElems
    // doubles, through a device staging buffer.
    static constexpr size_t kStageMax = (size_t)16 << 20;   // at most 128 MiB of float64
    size_t kStageElems = 0;        // elements of the staging buffer: the plan's largest transfer, capped (a 128-square sweep plan: 4 MB)
    double* stage_dev = nullptr;   // [kStageElems] + per-frame sums / targets
    double* stage_aux = nullptr;   // [B] per-frame targets
    double* stage_sums = nullptr;  // [B * V] per-image sums (+ scratch: aux_sums_elems)
    int ensure_stage() {
        if (stage_dev) return RL_OK;
        kStageElems = std::max(n_img(), std::min(kStageMax, (size_t)B * V * n_img()));
        HIP_TRY(hipMalloc((void**)&stage_dev, kStageElems * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&stage_aux, (size_t)B * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&stage_sums, aux_sums_elems((size_t)B * V) * sizeof(double)));
        bytes += kStageElems * sizeof(double);
        return RL_OK;
    }
    // images: `count` images of n_img() pixels; target (host, per image) may be nullptr
    // sums_out (host, optional): the images' sums as uploaded (before any scaling)
    int upload_images(const double* src, void* dst, size_t count, const double* target, std::vector<double>* sums_out = nullptr) {
        RL_TRY(ensure_stage());
        const size_t n = n_img();
        if (n > kStageElems) return fail(RL_ERR_UNSUPPORTED, "image larger than the staging buffer");   // (cannot happen: the buffer holds an image at least)
        const size_t per = kStageElems / n;
        if (sums_out) sums_out->assign(count, 0.0);
        if (target) HIP_TRY(hipMemcpyAsync(stage_aux, target, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        for (size_t f0 = 0; f0 < count; f0 += per) {
            const size_t nf = f0 + per <= count ? per : count - f0;
            HIP_TRY(hipMemcpyAsync(stage_dev, src + f0 * n, nf * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(aux_scale_convert(dtype, stage_dev, (char*)dst + f0 * n * esize(dtype), n, nf,
                                      target ? stage_aux + f0 : nullptr, stage_sums + f0, ctx->stream, target != nullptr || sums_out != nullptr));
            if (sums_out) HIP_TRY(hipMemcpyAsync(sums_out->data() + f0, stage_sums + f0, nf * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by the next slice
        }
        return RL_OK;
    }
    int upload(const double* src, void* dst, size_t n) {
        if (n % n_img() == 0) return upload_images(src, dst, n / n_img(), nullptr);
        return fail(RL_ERR_INVALID, "internal: upload of a partial image");
    }
    int download(const void* src, double* dst, size_t n) {
        if (dtype == RL_F64) {
            HIP_TRY(hipMemcpyAsync(dst, src, n * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            return RL_OK;
        }
        RL_TRY(ensure_stage());
        for (size_t o = 0; o < n; o += kStageElems) {
            const size_t m = o + kStageElems <= n ? kStageElems : n - o;
            HIP_TRY(aux_to_f64(dtype, (const char*)src + o * esize(dtype), stage_dev, m, ctx->stream));
            HIP_TRY(hipMemcpyAsync(dst + o, stage_dev, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        return RL_OK;
    }

    // noiseless = H(obj) on the device
    int forward_object() {
        if (sep) return sep_forward(obj, noiseless, B);
        RL_TRY(row(ROW_FWD, (unsigned)B, nullptr, spec_a, obj, nullptr, nullptr));
        RL_TRY(col(spec_a, spec_b, B, true));
        RL_TRY(row(ROW_INV, (unsigned)(B * V), spec_b, nullptr, nullptr, noiseless, nullptr));
        return RL_OK;
    }
    // ---- frame chunks: run the whole K-iteration loop on a slice of the batch whose
    // working set (spectra + measurement + estimate) fits the 256 MiB Infinity Cache,
    // so the inter-kernel traffic is served on die instead of from HBM.
    char* off(void* base, size_t elems) const { return (char*)base + elems * esize(dtype); }
    int chunk_frames() const {
        // per slice; `lanes` slices are in flight at once, together about the 256 MiB Infinity Cache
        const double budget_mb = getenv("RLSTED_CHUNK_MB") ? atof(getenv("RLSTED_CHUNK_MB")) : (lanes > 1 ? 108.0 : 288.0);
        const double specs = (V == 1 && inplace) ? 1.0 : 1.0 + V;   // spectra alive in an iteration
        const double per_frame = (specs * 2.0 * n_spec() + (1.0 + V) * n_img()) * esize(dtype);
        // Frames of 32 MB and more (2048^2 up) do not live in the Infinity Cache whatever the slice: there the slice
        // only has to fill the chip -- ~1 GB per slice measured best at 2048^2 (point 658 -> 713, 4 views 180 -> 192
        // frames/s over 2-frame / 1-frame slices).
        int c = (int)((!getenv("RLSTED_CHUNK_MB") && per_frame >= 32.0 * 1048576.0 ? 1024.0 : budget_mb) * 1048576.0 / per_frame);
        // many views: at least 8 frames per slice when no budget was given -- fewer leave the column
        // kernels (37 workgroups per 512^2 frame) too small to fill the chip (6 / 8 views: +10 % / +7 %)
        if (!getenv("RLSTED_CHUNK_MB") && c < 8 && per_frame * 8.0 <= 300.0 * 1048576.0) c = 8;
        // ... and enough frames for the column launches to fill the chip once (two 8-wave workgroups per CU): 512^2 has 37 column
        // tiles per frame, so 16 frames -- measured 3 / 4 views 6968 -> 7826 / 5840 -> 6100 frames/s over 8-frame slices
        if (!getenv("RLSTED_CHUNK_MB") && V > 1) {
            const int cw = ty->C[dtype] > 0 ? ty->C[dtype] : 8, tiles = (kx + cw - 1) / cw;
            const int need = ((512 + tiles - 1) / tiles + 7) / 8 * 8;
            if (c < need && per_frame * need <= 300.0 * 1048576.0) c = need;
        }
        if (c < 1) c = 1;
        if (c >= B) return B;
        // equal slices (a short last slice would run its 4 launches per iteration nearly empty; slices BALANCED to within one frame --
        // 6 6 5 5 5 5 instead of 6 6 6 6 6 2 for 32 frames -- measured +1 % at 2048^2 x 4 views and -1 % at 512^2 x 4 views, 250 frames: not done)
        const int fit = c;                       // frames the budget holds
        const int slices = (B + c - 1) / c;
        c = (B + slices - 1) / slices;
        if (c > 8) {                               // whole groups of 8 frames: up if that still fits the budget, down otherwise
            const int up = (c + 7) / 8 * 8;
            c = up <= fit ? up : std::max(8, fit / 8 * 8);
        }
        if (pair && (c & 1)) ++c;   // slices of whole frame pairs
        return c >= B ? B : c;
    }
    // estimate = 1 (ref:522).  with_spectrum: also spec_a = rowFFT(estimate); the first iteration does not need
    // it when it takes H(1) from spec_ones (iterate_chunk(first = true)).
    int start_estimate_chunk(int f0, int nf, bool with_spectrum = true) {
        HIP_TRY(aux_fill(dtype, off(est, (size_t)f0 * n_img()), (size_t)nf * n_img(), 1.0, cur()));
        if (with_spectrum && pair)
            RL_TRY(row_pair(ROW_FWD, nf, nullptr, pair_spec(f0), off(est, (size_t)f0 * n_img()), nullptr, nullptr));
        else if (with_spectrum && !sep)
            RL_TRY(row(ROW_FWD, (unsigned)nf, nullptr, off(spec_a, (size_t)f0 * n_spec() * 2), off(est, (size_t)f0 * n_img()),
                       nullptr, nullptr));
        return RL_OK;
    }
    int ensure_lanes() {
        if (fork) return RL_OK;
        HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        // (equal priorities: lane 0 at the highest stream priority and lane 1 at the lowest measured 17.2 k against 18.1 k frames/s)
        for (int l = 0; l < lanes; ++l) {
            HIP_TRY(hipStreamCreateWithFlags(&lane_stream[l], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&lane_done[l], hipEventDisableTiming));
        }
        return RL_OK;
    }
    // first: the iteration starts from estimate = 1 (just filled): H(estimate) is spec_ones for every frame, so the
    // column pass of H is skipped and ROW_RATIO reads the shared spectra (bit for bit what the pass would write)
    int iterate_chunk(int f0, int nf, bool first = false, bool from_ones = false) {
        // (study builds: the ratio of the iteration that starts from estimate = 1 is measurement / H(1), data scale)
        struct ScaleGuard {
            float& q; float keep;
            ~ScaleGuard() { q = keep; }
        } scale_guard{q_ratio, q_ratio};
        struct LoopGuard {
            bool& f;
            ~LoopGuard() { f = false; }
        } loop_guard{in_rl_loop};
        in_rl_loop = true;
        if (from_ones) q_ratio = q_est;
        if (sep) return sep_iterate(f0, nf);
        if (pair) {   // the whole iteration in the pair spectra, in place
            void* sp = pair_spec(f0);
            const int np = (nf + 1) / 2;
            if (first) {
                RL_TRY(row_pair(ROW_RATIO, nf, spec_ones_pair, sp, off(meas, (size_t)f0 * n_img()), nullptr, nullptr, 1));
            } else {
                RL_TRY(col_pair(sp, np, true));
                RL_TRY(row_pair(ROW_RATIO, nf, sp, sp, off(meas, (size_t)f0 * n_img()), nullptr, nullptr));
            }
            RL_TRY(col_pair(sp, np, false));
            RL_TRY(row_pair(ROW_UPDATE, nf, sp, drop_last_spectrum ? nullptr : sp, nullptr, off(est, (size_t)f0 * n_img()), norm));
            return RL_OK;
        }
        void* sa = off(spec_a, (size_t)f0 * n_spec() * 2);
        void* sb = off(spec_b, (size_t)f0 * V * n_spec() * 2);
        if (first && V == 1 && inplace) {
            RL_TRY(row(ROW_RATIO, (unsigned)nf, spec_ones, sa, off(meas, (size_t)f0 * n_img()), nullptr, nullptr, nullptr, -1, 1));
            RL_TRY(col(sa, sa, nf, false));
            RL_TRY(row(ROW_UPDATE, (unsigned)nf, sa, sa, nullptr, off(est, (size_t)f0 * n_img()), norm));
            return RL_OK;
        }
        if (V == 1 && inplace) {
            // one view: every pass maps a spectrum onto itself (a column tile / a row pair is read
            // completely before it is written), so the whole iteration runs in spec_a -- a third
            // less working set per frame for the Infinity Cache, and stores that hit lines just read
            RL_TRY(col(sa, sa, nf, true));
            RL_TRY(row(ROW_RATIO, (unsigned)nf, sa, sa, off(meas, (size_t)f0 * n_img()), nullptr, nullptr));
            RL_TRY(col(sa, sa, nf, false));
            RL_TRY(row(ROW_UPDATE, (unsigned)nf, sa, sa, nullptr, off(est, (size_t)f0 * n_img()), norm));
            return RL_OK;
        }
        if (col_split()) {
            // the split column pass: 1 + V and V + 1 column transforms per frame instead of 2 V and (fused) V + 1 on one register set
            void* sx = off(spec_x, (size_t)f0 * V * n_spec_x() * 2);
            if (first) {
                RL_TRY(row(ROW_RATIO, (unsigned)(nf * V), spec_ones, sb, off(meas, (size_t)f0 * V * n_img()), nullptr, nullptr, nullptr, -1, V));
            } else {
                if (split_h) RL_TRY(col_split_pass(sa, sb, sx, nf, COL_H));
                else RL_TRY(col(sa, sb, nf, true));          // (RLSTED_SPLIT_H=0: V whole-pass launches, nothing parked on this side)
                RL_TRY(row(ROW_RATIO, (unsigned)(nf * V), sb, sb, off(meas, (size_t)f0 * V * n_img()), nullptr, nullptr));
            }
            if (fuse_views && split_ht) {
                RL_TRY(col_split_pass(sb, sa, sx, nf, COL_HT_FUSED));
                RL_TRY(row(ROW_UPDATE, (unsigned)nf, sa, sa, nullptr, off(est, (size_t)f0 * n_img()), norm, nullptr, 1));
            } else {
                RL_TRY(col(sb, sb, nf, false));
                RL_TRY(row(ROW_UPDATE, (unsigned)nf, sb, sa, nullptr, off(est, (size_t)f0 * n_img()), norm));
            }
            return RL_OK;
        }
        if (first) {
            RL_TRY(row(ROW_RATIO, (unsigned)(nf * V), spec_ones, sb, off(meas, (size_t)f0 * V * n_img()), nullptr, nullptr, nullptr, -1, V));
        } else {
            RL_TRY(col(sa, sb, nf, true));                                                                   // H(est), column part
            RL_TRY(row(ROW_RATIO, (unsigned)(nf * V), sb, sb, off(meas, (size_t)f0 * V * n_img()), nullptr, nullptr));   // meas / H(est)
        }
        if (fuse_views && V > 1 && wave_private_y()) {
            // views summed in the Fourier domain: one inverse column + one inverse row transform per frame
            RL_TRY(col(sb, sa, nf, COL_HT_FUSED));
            RL_TRY(row(ROW_UPDATE, (unsigned)nf, sa, sa, nullptr, off(est, (size_t)f0 * n_img()), norm, nullptr, 1));
        } else {
            RL_TRY(col(sb, sb, nf, false));                                                                  // H_t, column part
            RL_TRY(row(ROW_UPDATE, (unsigned)nf, sb, sa, nullptr, off(est, (size_t)f0 * n_img()), norm));   // est *= H_t / norm
        }
        return RL_OK;
    }
    int start_estimate() {
        RL_TRY(start_estimate_chunk(0, B));
        est_ready = true;
        spec_valid = true;
        iterations = 0;
        return RL_OK;
    }
    int iterate_once() {
        RL_TRY(iterate_chunk(0, B));
        ++iterations;
        return RL_OK;
    }
    // noiseless = H(obj) on a slice
    int forward_slice(int f0, int nf) {
        if (sep) {
            const size_t o = (size_t)f0 * V * n_img();
            if (sep_one) return sep2d_(SEP_STORE_, off(obj, (size_t)f0 * n_img()), nullptr, nullptr, off(noiseless, o), nf);
            RL_TRY(sep_rows_(off(obj, (size_t)f0 * n_img()), off(sep_tmp(), o), nf * V, V));
            return sep_cols_(SEP_STORE_, off(sep_tmp(), o), nullptr, nullptr, off(noiseless, o), nf * V);
        }
        void* sb = off(spec_b, (size_t)f0 * V * n_spec() * 2);
        // (frame pairs: the other lane's slice iterates in spec_a in the pair layout, whose slice boundaries are not
        // this layout's -- the simulation then stays in spec_b, in place)
        void* sa = !pair ? off(spec_a, (size_t)f0 * n_spec() * 2) : sb;
        RL_TRY(row(ROW_FWD, (unsigned)nf, nullptr, sa, off(obj, (size_t)f0 * n_img()), nullptr, nullptr));
        if (col_split()) RL_TRY(col_split_pass(sa, sb, off(spec_x, (size_t)f0 * V * n_spec_x() * 2), nf, COL_H));   // (the same values as the whole pass)
        else RL_TRY(col(sa, sb, nf, true));
        RL_TRY(row(ROW_INV, (unsigned)(nf * V), sb, nullptr, nullptr, off(noiseless, (size_t)f0 * V * n_img()), nullptr));
        return RL_OK;
    }
    // one whole simulate + deconvolve cycle, slice by slice
    int run_cycle(int k, int rng_kind, uint64_t seed) { return run_slices(k, true, true, rng_kind, seed); }
    int join_open_lanes() {   // after a failed cycle between deferred joins
        if (!lanes_open) return RL_OK;
        lanes_open = false;
        for (int l = 0; l < kMaxLanes; ++l) {
            if (!lane_stream[l]) continue;
            HIP_TRY(hipEventRecord(lane_done[l], lane_stream[l]));
            HIP_TRY(hipStreamWaitEvent(ctx->stream, lane_done[l], 0));
        }
        return RL_OK;
    }
    // (optionally restart from est = 1 and) run k iterations, slice by slice
    int run_iterations(int k, bool restart) { return run_slices(k, restart, false, 0, 0); }
    int run_slices(int k, bool restart, bool simulate, int rng_kind, uint64_t seed) {
        const int cf = chunk_frames();
        const int slices = (B + cf - 1) / cf;
        const int nl = slices < lanes ? slices : lanes;
        // Lanes stay open between the back-to-back cycles of rl_deconv_bench_cycles (defer_join): slice s of every cycle
        // goes to the same lane, so stream order alone keeps each slice's buffers consistent and the lanes need not meet.
        const bool keep_open = defer_join && nl > 1;
        if (nl > 1 && !lanes_open) {
            RL_TRY(ensure_lanes());
            HIP_TRY(hipEventRecord(fork, ctx->stream));
            for (int l = 0; l < nl; ++l) HIP_TRY(hipStreamWaitEvent(lane_stream[l], fork, 0));
        }
        if (simulate) {   // one Poisson work list per slice: slices on different lanes run at the same time
            const size_t stride = (aux_poisson_workspace_bytes((size_t)cf * V * n_img()) + 255) / 256 * 256;
            if (slice_ws_bytes < stride * slices) {
                HIP_TRY(hipDeviceSynchronize());
                if (slice_ws) HIP_TRY(hipFree(slice_ws));
                slice_ws = nullptr;
                HIP_TRY(hipMalloc(&slice_ws, stride * slices));
                bytes += stride * slices - slice_ws_bytes;
                slice_ws_bytes = stride * slices;
            }
            slice_ws_stride = stride;
        }
        int rc = RL_OK;
        struct ActiveGuard {   // whatever path leaves this function, the launch helpers are back on the context's stream
            hipStream_t& a;
            ~ActiveGuard() { a = nullptr; }
        } active_guard{active};
        auto simulate_slice = [&](int sl, int f0, int nf) -> int {
            RL_TRY(forward_slice(f0, nf));
            void* ws = (char*)slice_ws + (size_t)sl * slice_ws_stride;   // this slice's Poisson work list
            TimedScope t(this, TK_POISSON, false);
            // (rl_batch_submit: a Philox key per frame, device arrays of B entries)
            hipError_t e = aux_poisson(dtype, off(noiseless, (size_t)f0 * V * n_img()), off(meas, (size_t)f0 * V * n_img()),
                                       (unsigned)n_img(), (unsigned)(nf * V), (unsigned)(f0 * V), seed, rng_kind, ws, cur(),
                                       run_key_seeds ? run_key_seeds + f0 : nullptr, run_key_ids ? run_key_ids + f0 : nullptr, (unsigned)V);
            if (e != hipSuccess) return fail(RL_ERR_HIP, std::string("Poisson kernels: ") + hipGetErrorString(e));
            return RL_OK;
        };
        // (Round 1 also tried the simulation of all slices ahead of the RL lanes on a stream of its own: 16.5 k against 17.2 k
        // frames/s -- 2.3 GB streamed through the memory system while the first slices iterate push their working sets out of the
        // Infinity Cache.  Slice by slice on the RL lanes the same overlap happens between lanes.)
        for (int sl = 0, f0 = 0; f0 < B && rc == RL_OK; f0 += cf, ++sl) {
            const int nf = f0 + cf <= B ? cf : B - f0;
            active = nl > 1 ? lane_stream[sl % nl] : nullptr;
            if (simulate) rc = simulate_slice(sl, f0, nf);
            const bool shortcut = restart && ones_shortcut && spec_ones && k > 0 && !sep;
            if (restart && rc == RL_OK) rc = start_estimate_chunk(f0, nf, !shortcut);
            // Frame pairs: the last iteration of a run of >= 4 does not transform its estimate forward again (1 of 2 row
            // transforms of that launch, the spectrum store); an rl_deconv_iterate that continues rebuilds it with one ROW_FWD.
            const bool drop = pair && k >= 4 && !keep_last_spectrum;
            for (int i = 0; i < k && rc == RL_OK; ++i) {
                drop_last_spectrum = drop && i == k - 1;
                rc = iterate_chunk(f0, nf, shortcut && i == 0, restart && i == 0);
            }
            drop_last_spectrum = false;
        }
        active = nullptr;
        if (nl > 1 && (!keep_open || rc != RL_OK)) {   // join, also on errors: the context's stream continues after every lane
            for (int l = 0; l < nl; ++l) {
                HIP_TRY(hipEventRecord(lane_done[l], lane_stream[l]));
                HIP_TRY(hipStreamWaitEvent(ctx->stream, lane_done[l], 0));
            }
            lanes_open = false;
        } else if (nl > 1) {
            lanes_open = true;
        }
        RL_TRY(rc);
        if (restart) {
            est_ready = true;
            spec_valid = true;
            iterations = 0;
        }
        if (pair && k >= 4 && !keep_last_spectrum) spec_valid = false;   // the last iteration left no spectrum behind
        iterations += k;
        return RL_OK;
    }
};

extern "C" {

const char* rl_last_error(void) { return rl::last_error().c_str(); }
int rl_version(void) { return 100; }

int rl_device_count(int* count) {
    if (!count) return fail(RL_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(RL_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return RL_OK;
}

int rl_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(RL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (bytes == 0) return fail(RL_ERR_INVALID, "bytes == 0");
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return RL_OK;
}

int rl_host_free(void* p) {
    if (p) HIP_TRY(hipHostFree(p));
    return RL_OK;
}

int rl_fft_length_for(int n) {
    for (int L : kLengths)
        if (L >= n) return L;
    return 0;
}

// A sweep deals its plan groups to several contexts of a GPU (sweep.py), each with a stream of its own plus its plans' slice and copy
// streams.  The HIP runtime maps a process's streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise; with 8 the
// small launches of neighbouring groups overlap further (BASELINE config 4, 1152 tasks: 29.7 -> 26.4-26.8 ms on 4-6 contexts,
// profiles/r04/sweep_hw_queues.log).  The runtime reads the variable when it initialises -- at this process's first HIP call --
// so the default is set when the library is loaded, and only if the user has not set it.
__attribute__((constructor)) static void rl_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

int rl_ctx_create(int device, rl_ctx** out) {
    if (!out) return fail(RL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RL_ERR_INVALID, "no such device");
    HIP_TRY(hipSetDevice(device));
    rl_ctx* c = new rl_ctx;
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(RL_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = c;
    return RL_OK;
}

int rl_ctx_destroy(rl_ctx* c) {
    if (!c) return RL_OK;
    (void)hipSetDevice(c->device);
    for (auto& kv : c->tw) (void)hipFree(kv.second);
    if (c->psf_work) (void)hipFree(c->psf_work);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RL_OK;
}

int rl_ctx_synchronize(rl_ctx* c) {
    if (!c) return fail(RL_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RL_OK;
}

int rl_deconv_destroy(rl_deconv* h) {
    if (!h) return RL_OK;
    (void)hipSetDevice(h->ctx->device);
    // nothing of this plan may still be running when its buffers go away (slice streams included)
    (void)hipStreamSynchronize(h->ctx->stream);
    for (int l = 0; l < rl_deconv::kMaxLanes; ++l)
        if (h->lane_stream[l]) (void)hipStreamSynchronize(h->lane_stream[l]);
    void* bufs[] = {h->psf_hat_pair, h->psf_hat_pair_re, h->spec_ones_pair, h->sep_u, h->sep_v, h->sep_uf, h->sep_vf, h->spec_ones, h->psf_hat_re, h->psf_hat, h->spec_a, h->spec_b, h->spec_x, h->obj, h->noiseless, h->meas, h->est, h->norm, h->scratch,
                    h->stage_dev, h->stage_aux, h->stage_sums, h->slice_ws, h->key_seeds, h->key_ids, h->unresolved};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    for (rl_deconv::BatchSlot& sl : h->bslot) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.freed) (void)hipEventDestroy(sl.freed);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->batch_out) (void)hipFree(h->batch_out);
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->fork) (void)hipEventDestroy(h->fork);
    for (int l = 0; l < rl_deconv::kMaxLanes; ++l) {
        if (h->lane_done[l]) (void)hipEventDestroy(h->lane_done[l]);
        if (h->lane_stream[l]) (void)hipStreamDestroy(h->lane_stream[l]);
    }
    delete h;
    return RL_OK;
}

static int deconv_build(rl_deconv* h, const double* psfs) {
    rl_ctx* ctx = h->ctx;
    const int cy_lo = (h->py - 1) / 2, cy_hi = h->py - 1 - cy_lo;
    const int cx_lo = (h->px - 1) / 2, cx_hi = h->px - 1 - cx_lo;
    // wrap-free sizes: L >= n + max(offset below, offset above the PSF centre)
    h->ly = rl_fft_length_for(h->ny + (cy_lo > cy_hi ? cy_lo : cy_hi));
    h->lx = rl_fft_length_for(h->nx + (cx_lo > cx_hi ? cx_lo : cx_hi));
    if (!h->ly || !h->lx)
        return fail(RL_ERR_UNSUPPORTED, "image + PSF half width exceeds the largest built transform length (4608)");
    h->ty = table_for(h->ly);
    h->tx = table_for(h->lx);
    h->kx = h->lx / 2 + 1;
    h->pitch = (h->kx + 7) / 8 * 8;
    RL_TRY(ctx->prepare(h->ty));
    RL_TRY(ctx->prepare(h->tx));
    RL_TRY(ctx->twiddles(h->ty, h->dtype, &h->twy, true));
    RL_TRY(ctx->twiddles(h->tx, h->dtype, &h->twx));
    const size_t es = esize(h->dtype), B = (size_t)h->B, V = (size_t)h->V;
    // (measured, pitch lx against lx + 32: 512^2 18.7 k against 18.1 k frames/s, 2048^2 690 against 716, 4096^2 K = 100 28.0 against 28.6)
    h->pair_pitch = h->lx + (getenv("RLSTED_PAIR_PAD") ? std::max(0, atoi(getenv("RLSTED_PAIR_PAD"))) / 8 * 8 : (h->lx >= 1152 ? 32 : 0));
    struct Req { void** p; size_t n; };
    const Req reqs[] = {
        {&h->psf_hat, V * h->ly * h->pitch * 2 * es},   // (spec_a: the frames' half spectra or the pairs' full ones)
        {&h->spec_a, std::max(B * h->n_spec(), (B + 1) / 2 * h->n_spec_pair()) * 2 * es},
        {&h->spec_b, B * V * h->n_spec() * 2 * es},   {&h->obj, B * h->n_img() * es},
        {&h->noiseless, B * V * h->n_img() * es},     {&h->meas, B * V * h->n_img() * es},
        {&h->est, B * h->n_img() * es},               {&h->norm, h->n_img() * es},
        {&h->scratch, std::max(B * V * h->n_img() * es, aux_poisson_workspace_bytes(B * V * h->n_img()))}};   // also the Poisson work list
    for (const Req& r : reqs) {
        // RL_STREAM_SLACK: the streaming row kernels load whole 64-lane segments without clamping;
        // lanes past the end of the last row of a buffer read (and discard) these bytes
        HIP_TRY(hipMalloc(r.p, r.n + RL_STREAM_SLACK));
        HIP_TRY(hipMemsetAsync(static_cast<char*>(*r.p) + r.n, 0, RL_STREAM_SLACK, ctx->stream));
        h->bytes += r.n + RL_STREAM_SLACK;
    }
    HIP_TRY(hipMalloc((void**)&h->unresolved, sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(h->unresolved, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));

    // PSF spectra: direct DFT of the (py x px) support in float64 on the device
    {
        void *wy = nullptr, *wx = nullptr, *psf_dev = nullptr, *s1 = nullptr;
        RL_TRY(ctx->plain_twiddles(h->ly, &wy));
        RL_TRY(ctx->plain_twiddles(h->lx, &wx));
        const size_t np = V * h->py * h->px;
        HIP_TRY(hipMalloc(&psf_dev, np * 8));
        HIP_TRY(hipMalloc(&s1, V * h->py * h->kx * 16));
        HIP_TRY(hipMemcpyAsync(psf_dev, psfs, np * 8, hipMemcpyHostToDevice, ctx->stream));
        hipError_t e = aux_psf_spectrum(h->dtype, (const double*)psf_dev, wx, wy, s1, h->psf_hat, h->V, h->py, h->px,
                                        h->ly, h->lx, h->kx, h->pitch, h->ty->psf_transposed[h->dtype], ctx->stream);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(psf_dev);
        (void)hipFree(s1);
        HIP_TRY(e);
        HIP_TRY(e2);
    }
    // A point-symmetric PSF (every PSF of the reference: symmetric "to 1e-15") has a real spectrum.
    // Where the imaginary parts are rounding noise (<= 1e-12 of the largest value; in f32 they vanish against
    // the real parts' own rounding) the wave-private column kernels multiply by the real parts alone: half the
    // multiplier bytes per column launch.  RLSTED_REAL_PSF=0 keeps the complex multiplier.
    if (h->psf_transposed() && !(getenv("RLSTED_REAL_PSF") && atoi(getenv("RLSTED_REAL_PSF")) == 0)) {
        const size_t nz = V * (size_t)h->kx * h->ly;
        void* re = nullptr;
        double* stats = nullptr;
        HIP_TRY(hipMalloc(&re, nz * es + RL_STREAM_SLACK));
        hipError_t e = hipMalloc((void**)&stats, 2 * sizeof(double));
        double hs[2] = {1.0, 1.0};
        if (e == hipSuccess) e = aux_split_real(h->dtype, h->psf_hat, nz, re, stats, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hs, stats, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (stats) (void)hipFree(stats);
        if (e != hipSuccess) {
            (void)hipFree(re);
            HIP_TRY(e);
        }
        h->psf_hat_imag_ratio = hs[1] > 0 ? hs[0] / hs[1] : 0.0;
        if (h->psf_hat_imag_ratio <= 1e-12) {
            h->psf_hat_re = re;
            h->bytes += nz * es + RL_STREAM_SLACK;
        } else {
            (void)hipFree(re);
        }
    }
    // The split column pass parks B * V column spectra (24 MB per 2048^2 image).  The set-up below runs it on ONE frame (H(1)); the
    // full allocation follows the strategy selection further down: a plan that ends on the separable stencils never touches it.
    if (h->col_split()) {
        const size_t n = V * h->n_spec_x() * 2 * es;
        HIP_TRY(hipMalloc(&h->spec_x, n));
        HIP_TRY(hipMemsetAsync(h->spec_x, 0, n, ctx->stream));
    }
    // H_t(ones), line_sted_tools.py:589-592: sum_v clamp(conv(ones, psf_v))
    HIP_TRY(aux_fill(h->dtype, h->est, h->n_img(), 1.0, ctx->stream));
    {
        const int keepB = h->B;
        h->B = 1;
        // (study builds: the spectra of a frame of ones have DC = pixels x sum(psf), not the data's)
        const float keep_q = h->q_est;
        double psf_sum = 1.0;
        for (size_t v = 0; v < V; ++v) {
            double t = 0.0;
            for (size_t i = 0; i < (size_t)h->py * h->px; ++i) t += psfs[v * h->py * h->px + i];
            psf_sum = std::max(psf_sum, t);
        }
        if (getenv("RLSTED_Q_EXP_EST")) h->q_est = std::ldexp(1.0f, 14 - (int)std::ceil(std::log2((double)h->n_img() * psf_sum)) - 1);
        int r = h->row(ROW_FWD, 1, nullptr, h->spec_a, h->est, nullptr, nullptr);
        if (r == RL_OK) r = h->col(h->spec_a, h->spec_b, 1, true);
        if (r == RL_OK) r = h->row(ROW_ADJ, 1, h->spec_b, nullptr, nullptr, h->norm, nullptr);
        h->B = keepB;
        h->q_est = keep_q;
        RL_TRY(r);
        // spec_b now holds the column-transformed spectra of H(1), one per view: every frame's first iteration
        const size_t ones_bytes = V * h->n_spec() * 2 * es;
        HIP_TRY(hipMalloc(&h->spec_ones, ones_bytes + RL_STREAM_SLACK));
        HIP_TRY(hipMemsetAsync(static_cast<char*>(h->spec_ones) + ones_bytes, 0, RL_STREAM_SLACK, ctx->stream));
        HIP_TRY(hipMemcpyAsync(h->spec_ones, h->spec_b, ones_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        h->bytes += ones_bytes + RL_STREAM_SLACK;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // ---- f32 plans: the normaliser to float64 rounding from the PSFs' integral images (aux_box_norm) ----
    if (h->exact_norm) {
        const size_t py = h->py, px = h->px, stride = (py + 1) * (px + 1);
        std::vector<double> integ(V * stride, 0.0);
        for (size_t v = 0; v < V; ++v)
            for (size_t a = 0; a < py; ++a) {
                double row = 0.0;   // running sum of PSF row a
                for (size_t b = 0; b < px; ++b) {
                    row += psfs[(v * py + a) * px + b];
                    integ[v * stride + (a + 1) * (px + 1) + (b + 1)] = integ[v * stride + a * (px + 1) + (b + 1)] + row;
                }
            }
        double* integ_dev = nullptr;
        HIP_TRY(hipMalloc((void**)&integ_dev, integ.size() * sizeof(double)));
        hipError_t e = hipMemcpyAsync(integ_dev, integ.data(), integ.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = aux_box_norm(h->dtype, integ_dev, h->norm, h->V, h->py, h->px, h->ny, h->nx, ctx->stream);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(integ_dev);
        HIP_TRY(e);
        HIP_TRY(e2);
    }
    // ---- strategy selection: direct separable stencils when every view is rank 1 and small ----
    const int sep_mode = getenv("RLSTED_SEP") ? atoi(getenv("RLSTED_SEP")) : 1;
    const int max_taps = getenv("RLSTED_SEP_MAX_TAPS") ? atoi(getenv("RLSTED_SEP_MAX_TAPS")) : 16;
    if (sep_mode > 0 && (sep_mode > 1 || h->py + h->px <= max_taps) && sep_two_pass_fits(h->dtype, h->py, h->px)) {
        const size_t py = h->py, px = h->px;
        std::vector<double> u(V * py), vv(V * px);
        bool rank1 = true;
        for (size_t v = 0; v < V && rank1; ++v) {
            const double* p = psfs + v * py * px;
            size_t a0 = 0, b0 = 0;
            double pmax = 0.0;
            for (size_t a = 0; a < py; ++a)
                for (size_t b = 0; b < px; ++b)
                    if (std::fabs(p[a * px + b]) > pmax) { pmax = std::fabs(p[a * px + b]); a0 = a; b0 = b; }
            if (!(pmax > 0.0)) { rank1 = false; break; }
            // cross approximation through the largest element: exact for a rank-1 matrix
            for (size_t a = 0; a < py; ++a) u[v * py + a] = p[a * px + b0];
            for (size_t b = 0; b < px; ++b) vv[v * px + b] = p[a0 * px + b] / p[a0 * px + b0];
            for (size_t a = 0; a < py && rank1; ++a)
                for (size_t b = 0; b < px; ++b)
                    if (std::fabs(p[a * px + b] - u[v * py + a] * vv[v * px + b]) > 1e-12 * pmax) { rank1 = false; break; }
        }
        if (rank1) {
            HIP_TRY(hipMalloc(&h->sep_u, V * py * es));
            HIP_TRY(hipMalloc(&h->sep_v, V * px * es));
            if (h->dtype == RL_F64) {
                HIP_TRY(hipMemcpy(h->sep_u, u.data(), V * py * 8, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(h->sep_v, vv.data(), V * px * 8, hipMemcpyHostToDevice));
            } else {
                std::vector<float> uf(u.begin(), u.end()), vf(vv.begin(), vv.end());
                HIP_TRY(hipMemcpy(h->sep_u, uf.data(), V * py * 4, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(h->sep_v, vf.data(), V * px * 4, hipMemcpyHostToDevice));
            }
            h->sep = true;
            // RLSTED_SEP_ONE: 0 two passes, 1 (default) one kernel up to 24 taps a side (beyond, the two-pass form is
            // faster: profiles/r02/separable_vs_fft.json), 2 one kernel whenever the tile fits LDS
            const int one_mode = getenv("RLSTED_SEP_ONE") ? atoi(getenv("RLSTED_SEP_ONE")) : 1;
            const bool want_one = one_mode >= 2 || (one_mode == 1 && std::max(py, px) <= 24);
            if (want_one && sep2d_fits(h->dtype, h->py, h->px, (int)V)) {
                const size_t pyp = (py + 7) / 8 * 8, pxp = (px + 7) / 8 * 8;
                std::vector<double> uf(V * pyp, 0.0), vf(V * pxp, 0.0);
                for (size_t v = 0; v < V; ++v) {
                    for (size_t k = 0; k < py; ++k) uf[v * pyp + k] = u[v * py + (py - 1 - k)];
                    for (size_t k = 0; k < px; ++k) vf[v * pxp + k] = vv[v * px + (px - 1 - k)];
                }
                HIP_TRY(hipMalloc(&h->sep_uf, V * pyp * es));
                HIP_TRY(hipMalloc(&h->sep_vf, V * pxp * es));
                if (h->dtype == RL_F64) {
                    HIP_TRY(hipMemcpy(h->sep_uf, uf.data(), V * pyp * 8, hipMemcpyHostToDevice));
                    HIP_TRY(hipMemcpy(h->sep_vf, vf.data(), V * pxp * 8, hipMemcpyHostToDevice));
                } else {
                    std::vector<float> a(uf.begin(), uf.end()), b(vf.begin(), vf.end());
                    HIP_TRY(hipMemcpy(h->sep_uf, a.data(), V * pyp * 4, hipMemcpyHostToDevice));
                    HIP_TRY(hipMemcpy(h->sep_vf, b.data(), V * pxp * 4, hipMemcpyHostToDevice));
                }
                h->sep_one = true;
            }
            // H_t(ones) through the same stencils (ref:589-592); the V copies of ones live in scratch
            HIP_TRY(aux_fill(h->dtype, h->scratch, V * h->n_img(), 1.0, ctx->stream));
            if (h->sep_one) {
                RL_TRY(h->sep2d_(SEP_SUM_, h->scratch, nullptr, nullptr, h->norm, 1));
            } else {
                RL_TRY(h->sep_rows_(h->scratch, h->sep_tmp(), (int)V, 1));
                RL_TRY(h->sep_cols_(SEP_SUM_, h->sep_tmp(), nullptr, nullptr, h->norm, 1));
            }
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    // ---- ... or a direct 2-D stencil when the views are small but not rank 1 (sep_kernels.hip k_sep2d DIRECT): py * px multiply-adds per
    // pixel and view, all of one sign for a non-negative PSF -- the relative accuracy an FFT convolution cannot give a dark region
    // (DESIGN.md section 3b).  RLSTED_DIRECT: 0 never, 1 (default) up to RLSTED_DIRECT_MAX_TAPS = 49 taps (where it is also the faster
    // path), 2 whenever the tile fits LDS (f32 plans on sparse samples with PSFs up to ~15 x 15: 2x the FFT path's time at 11 x 11).
    {
        const int direct_mode = getenv("RLSTED_DIRECT") ? atoi(getenv("RLSTED_DIRECT")) : 1;
        const int direct_max = getenv("RLSTED_DIRECT_MAX_TAPS") ? atoi(getenv("RLSTED_DIRECT_MAX_TAPS")) : 49;
        if (!h->sep && sep_mode > 0 && direct_mode > 0 && (direct_mode > 1 || h->py * h->px <= direct_max) && h->py * h->px <= 1024 &&
            direct2d_fits(h->dtype, h->py, h->px, (int)V)) {
            const size_t py = h->py, px = h->px, pyp = (py + 7) / 8 * 8;
            std::vector<double> f(V * px * pyp, 0.0);
            for (size_t v = 0; v < V; ++v)
                for (size_t l = 0; l < px; ++l)
                    for (size_t k = 0; k < py; ++k) f[(v * px + l) * pyp + k] = psfs[(v * py + (py - 1 - k)) * px + (px - 1 - l)];
            HIP_TRY(hipMalloc(&h->sep_uf, f.size() * es));
            if (h->dtype == RL_F64) {
                HIP_TRY(hipMemcpy(h->sep_uf, f.data(), f.size() * 8, hipMemcpyHostToDevice));
            } else {
                std::vector<float> ff(f.begin(), f.end());
                HIP_TRY(hipMemcpy(h->sep_uf, ff.data(), ff.size() * 4, hipMemcpyHostToDevice));
            }
            h->sep = h->sep_one = h->sep_direct = true;      // (sep_vf stays null: that is how sep2d tells the two forms apart)
            HIP_TRY(aux_fill(h->dtype, h->scratch, V * h->n_img(), 1.0, ctx->stream));
            RL_TRY(h->sep2d_(SEP_SUM_, h->scratch, nullptr, nullptr, h->norm, 1));      // H_t(ones) through the same stencil (ref:589-592)
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    if (h->col_split()) {   // (the one-frame parking space of the set-up -> the plan's, unless the stencils took over)
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(h->spec_x));
        h->spec_x = nullptr;
        if (!h->sep) {
            const size_t n = B * V * h->n_spec_x() * 2 * es;
            HIP_TRY(hipMalloc(&h->spec_x, n));
            HIP_TRY(hipMemsetAsync(h->spec_x, 0, n, ctx->stream));
            h->bytes += n;
        }
    }
    // ---- frame pairs for the Richardson-Lucy loop (single view, even batch, wave-private lengths) ----
    // Default: f32 plans (the throughput mode).  f64 plans keep every frame's arithmetic independent of its neighbour in the
    // batch (a pair's two frames share rounding errors: 1e-16-level differences with the partner frame).
    // Multi-view plans: built and tested (RLSTED_PAIR=1), no gain (512^2, 4 views: 30.0 ms per 64 frames x 20 iterations either
    // way -- the multi-view column kernels set the pace), so the default pairs single-view plans only.
    // Long transforms (L >= 1152, one workgroup-synchronous row transform per workgroup): built and tested too, 2048^2 -3.5 %,
    // 4096^2 -12 % time per iteration -- but the per-frame loop's Hermitian split averages the two mirrored halves of every
    // row spectrum, the pair loop does not, and where f32 has no margin left that shows: white noise at 2048^2, K = 20,
    // 8.8e-6 -> 1.05e-5 against the f64 plan.  Default there: per frame (RLSTED_PAIR=1 pairs them).
    const bool want_pair = getenv("RLSTED_PAIR") ? atoi(getenv("RLSTED_PAIR")) != 0 : (h->dtype == RL_F32 && V == 1);
    const bool pair_views_ok = V == 1 && h->inplace && h->B >= 2;   // (a single frame has no partner -- and the set-up below fills two frames of ones)
    // an odd batch leaves its last pair half empty: allowed where the pair spectra still fit the per-frame buffers
    if (want_pair && !h->sep && pair_views_ok && h->tx->launch_row_pair && h->psf_transposed()) {
        const size_t nz = V * (size_t)h->lx * h->ly;
        void *wy = nullptr, *wx = nullptr, *psf_dev = nullptr, *s1 = nullptr;
        RL_TRY(ctx->plain_twiddles(h->ly, &wy));
        RL_TRY(ctx->plain_twiddles(h->lx, &wx));
        HIP_TRY(hipMalloc(&h->psf_hat_pair, nz * 2 * es + RL_STREAM_SLACK));
        HIP_TRY(hipMalloc(&psf_dev, V * (size_t)h->py * h->px * 8));
        HIP_TRY(hipMalloc(&s1, V * (size_t)h->py * h->lx * 16));
        HIP_TRY(hipMemcpyAsync(psf_dev, psfs, V * (size_t)h->py * h->px * 8, hipMemcpyHostToDevice, ctx->stream));
        hipError_t e = aux_psf_spectrum(h->dtype, (const double*)psf_dev, wx, wy, s1, h->psf_hat_pair, h->V, h->py, h->px, h->ly, h->lx,
                                        h->lx, h->lx, 1, ctx->stream);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(psf_dev);
        (void)hipFree(s1);
        HIP_TRY(e);
        HIP_TRY(e2);
        h->bytes += nz * 2 * es + RL_STREAM_SLACK;
        if (h->psf_hat_re) {   // the half-width spectra are real: so are the full ones
            double* stats = nullptr;
            HIP_TRY(hipMalloc(&h->psf_hat_pair_re, nz * es + RL_STREAM_SLACK));
            HIP_TRY(hipMalloc((void**)&stats, 2 * sizeof(double)));
            e = aux_split_real(h->dtype, h->psf_hat_pair, nz, h->psf_hat_pair_re, stats, ctx->stream);
            e2 = hipStreamSynchronize(ctx->stream);
            (void)hipFree(stats);
            HIP_TRY(e);
            HIP_TRY(e2);
            h->bytes += nz * es + RL_STREAM_SLACK;
        }
        h->pair = h->pair_layout = true;
        if (getenv("RLSTED_PAIR_MAX_RATIO")) h->pair_max_ratio = atof(getenv("RLSTED_PAIR_MAX_RATIO"));
        // H(1 + i) of a pair of ones frames, column part (one spectrum per view): what every pair's first iteration reads
        const size_t ones_bytes = V * h->n_spec_pair() * 2 * es;
        HIP_TRY(hipMalloc(&h->spec_ones_pair, ones_bytes + RL_STREAM_SLACK));
        HIP_TRY(aux_fill(h->dtype, h->est, 2 * h->n_img(), 1.0, ctx->stream));
        RL_TRY(h->row_pair(ROW_FWD, 2, nullptr, h->spec_a, h->est, nullptr, nullptr));
        RL_TRY(h->col_pair(h->spec_a, V == 1 ? h->spec_a : h->spec_b, 1, rl_deconv::COL_H));
        HIP_TRY(hipMemcpyAsync(h->spec_ones_pair, V == 1 ? h->spec_a : h->spec_b, ones_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemsetAsync(static_cast<char*>(h->spec_ones_pair) + ones_bytes, 0, RL_STREAM_SLACK, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        h->bytes += ones_bytes + RL_STREAM_SLACK;
    }
    return RL_OK;
}

int rl_deconv_create(rl_ctx* ctx, const double* psfs, int n_psf, int py, int px, int batch, int ny, int nx, int dtype,
                     rl_deconv** out) {
    if (!ctx || !psfs || !out) return fail(RL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_psf < 1 || py < 1 || px < 1 || batch < 1 || ny < 1 || nx < 1) return fail(RL_ERR_INVALID, "non-positive size");
    if (dtype != RL_F32 && dtype != RL_F64) return fail(RL_ERR_INVALID, "dtype must be RL_F32 or RL_F64");
    if (n_psf > rl_deconv::kMaxGridY) return fail(RL_ERR_UNSUPPORTED, "more than 65535 views");
    HIP_TRY(hipSetDevice(ctx->device));
    rl_deconv* h = new rl_deconv;
    h->ctx = ctx;
    h->V = n_psf; h->py = py; h->px = px; h->B = batch; h->ny = ny; h->nx = nx; h->dtype = dtype;
    if (getenv("RLSTED_INPLACE")) h->inplace = atoi(getenv("RLSTED_INPLACE")) != 0;
    if (getenv("RLSTED_COL_ORDER")) h->col_order = atoi(getenv("RLSTED_COL_ORDER")) < 1 ? 1 : atoi(getenv("RLSTED_COL_ORDER"));
    if (getenv("RLSTED_Q_EXP_EST")) h->q_est = std::ldexp(1.0f, 14 - atoi(getenv("RLSTED_Q_EXP_EST")));
    if (getenv("RLSTED_Q_EXP_RATIO")) h->q_ratio = std::ldexp(1.0f, 14 - atoi(getenv("RLSTED_Q_EXP_RATIO")));
    if (getenv("RLSTED_ONES_SHORTCUT")) h->ones_shortcut = atoi(getenv("RLSTED_ONES_SHORTCUT")) != 0;
    if (getenv("RLSTED_KEEP_LAST_SPECTRUM")) h->keep_last_spectrum = atoi(getenv("RLSTED_KEEP_LAST_SPECTRUM")) != 0;
    if (getenv("RLSTED_LANES")) {
        h->lanes = atoi(getenv("RLSTED_LANES"));
        if (h->lanes < 1) h->lanes = 1;
        if (h->lanes > rl_deconv::kMaxLanes) h->lanes = rl_deconv::kMaxLanes;
    }
    if (getenv("RLSTED_COL_SPLIT")) h->split_wanted = atoi(getenv("RLSTED_COL_SPLIT")) != 0;
    if (getenv("RLSTED_SPLIT_HT")) h->split_ht = atoi(getenv("RLSTED_SPLIT_HT")) != 0;
    // H through the split pass from three views on: with two the forward half saved (1 + V against 2 V column transforms) does not pay
    // for parking the spectrum -- measured, 2048^2: 2 views 429 (split) against 449 frames/s, 4 views 269 against 258
    h->split_h = getenv("RLSTED_SPLIT_H") ? atoi(getenv("RLSTED_SPLIT_H")) != 0 : n_psf >= 3;
    if (getenv("RLSTED_COL_MULTI")) h->col_multi = atoi(getenv("RLSTED_COL_MULTI")) != 0;
    h->fuse_views = getenv("RLSTED_FUSE_VIEWS") ? atoi(getenv("RLSTED_FUSE_VIEWS")) != 0 : (dtype == RL_F32);
    h->exact_norm = getenv("RLSTED_EXACT_NORM") ? atoi(getenv("RLSTED_EXACT_NORM")) != 0 : (dtype == RL_F32);
    {   // ratio - 1 needs H_t(ones) == the normaliser: every PSF value >= 0 (and not the 16-bit storage study, whose scales assume ratio spectra)
        bool nonneg = true;
        for (size_t i = 0; i < (size_t)n_psf * py * px && nonneg; ++i) nonneg = psfs[i] >= 0.0;
        const bool want = getenv("RLSTED_SUB_ONE") ? atoi(getenv("RLSTED_SUB_ONE")) != 0 : (dtype == RL_F32 && RL_SPEC_QUANT == 0);
        // (RLSTED_FUSE_VIEWS=0 selects the reference's per-view clamp in H_t: `ratio - 1` clamps the view sum, so it is off then)
        const bool per_view_clamp = n_psf > 1 && getenv("RLSTED_FUSE_VIEWS") && atoi(getenv("RLSTED_FUSE_VIEWS")) == 0;
        h->sub_one = want && nonneg && !per_view_clamp;
    }
    int r = deconv_build(h, psfs);
    if (r != RL_OK) {
        std::string keep = rl::last_error();
        rl_deconv_destroy(h);
        rl::last_error() = keep;
        return r;
    }
    *out = h;
    return RL_OK;
}

int rl_deconv_info(const rl_deconv* h, int* ly, int* lx, int* pitch, size_t* device_bytes) {
    if (!h) return fail(RL_ERR_INVALID, "handle is NULL");
    if (ly) *ly = h->ly;
    if (lx) *lx = h->lx;
    if (pitch) *pitch = h->pitch;
    if (device_bytes) *device_bytes = h->bytes;
    return RL_OK;
}

int rl_deconv_set_object(rl_deconv* h, const double* obj, const double* total_brightness) {
    if (!h || !obj) return fail(RL_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    // :505-506  obj *= total_brightness / obj.sum(), per frame, on the device
    RL_TRY(h->upload_images(obj, h->obj, (size_t)h->B, total_brightness, &h->obj_level));
    if (total_brightness) h->obj_level.assign(total_brightness, total_brightness + h->B);   // the frames' sums after scaling
    HIP_TRY(hipEventRecord(h->ev0, h->ctx->stream));
    RL_TRY(h->forward_object());
    HIP_TRY(hipEventRecord(h->ev1, h->ctx->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_sim_ms = ms;
    h->have_obj = true;
    h->spec_valid = false;
    return RL_OK;
}

int rl_deconv_simulate(rl_deconv* h, int rng_kind, uint64_t seed) {
    if (!h) return fail(RL_ERR_INVALID, "handle is NULL");
    if (!h->have_obj) return fail(RL_ERR_STATE, "rl_deconv_set_object has not been called");
    if (rng_kind != RL_RNG_NONE && rng_kind != RL_RNG_PHILOX) return fail(RL_ERR_INVALID, "unknown rng_kind");
    HIP_TRY(hipSetDevice(h->ctx->device));
    HIP_TRY(aux_poisson(h->dtype, h->noiseless, h->meas, (unsigned)h->n_img(), (unsigned)(h->B * h->V), 0, seed, rng_kind,
                        h->scratch, h->ctx->stream));
    HIP_TRY(hipStreamSynchronize(h->ctx->stream));
    h->meas_level = h->obj_level;   // the measurement's level follows the object's
    h->choose_loop(h->meas_level);
    h->have_meas = true;
    h->meas_external = false;
    h->est_ready = false;
    return RL_OK;
}

int rl_deconv_simulate_keyed(rl_deconv* h, int rng_kind, const uint64_t* seeds, const uint32_t* image_ids) {
    if (!h || !seeds || !image_ids) return fail(RL_ERR_INVALID, "NULL argument");
    if (!h->have_obj) return fail(RL_ERR_STATE, "rl_deconv_set_object has not been called");
    if (rng_kind != RL_RNG_NONE && rng_kind != RL_RNG_PHILOX) return fail(RL_ERR_INVALID, "unknown rng_kind");
    for (int f = 0; f < h->B; ++f)   // image index = id * V + view must fit the 32-bit Philox counter word
        if ((uint64_t)image_ids[f] * (uint64_t)h->V + (uint64_t)h->V > 0xffffffffull) return fail(RL_ERR_INVALID, "image id too large");
    HIP_TRY(hipSetDevice(h->ctx->device));
    if (!h->key_seeds) {
        HIP_TRY(hipMalloc(&h->key_seeds, (size_t)h->B * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&h->key_ids, (size_t)h->B * sizeof(uint32_t)));
    }
    HIP_TRY(hipMemcpyAsync(h->key_seeds, seeds, (size_t)h->B * sizeof(uint64_t), hipMemcpyHostToDevice, h->ctx->stream));
    HIP_TRY(hipMemcpyAsync(h->key_ids, image_ids, (size_t)h->B * sizeof(uint32_t), hipMemcpyHostToDevice, h->ctx->stream));
    HIP_TRY(aux_poisson(h->dtype, h->noiseless, h->meas, (unsigned)h->n_img(), (unsigned)(h->B * h->V), 0, 0, rng_kind,
                        h->scratch, h->ctx->stream, (const unsigned long long*)h->key_seeds, (const unsigned*)h->key_ids,
                        (unsigned)h->V));
    HIP_TRY(hipStreamSynchronize(h->ctx->stream));   // the host arrays may go away
    h->meas_level = h->obj_level;
    h->choose_loop(h->meas_level);
    h->meas_external = false;
    h->meas_negative = false;   // (Poisson draws + 1e-9)
    h->have_meas = true;
    h->est_ready = false;
    return RL_OK;
}

int rl_deconv_set_measurement(rl_deconv* h, const double* noisy) {
    if (!h || !noisy) return fail(RL_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    {
        std::vector<double> sums;
        RL_TRY(h->upload_images(noisy, h->meas, (size_t)h->B * h->V, nullptr, &sums));
        h->meas_level.assign((size_t)h->B, 0.0);
        for (size_t i = 0; i < sums.size(); ++i) h->meas_level[i / h->V] += sums[i];
        h->choose_loop(h->meas_level);
        RL_TRY(h->scan_meas_negative());      // (an uploaded measurement may hold negative pixels: sub())
    }
    h->meas_external = false;
    h->have_meas = true;
    h->est_ready = false;
    return RL_OK;
}

int rl_deconv_set_estimate(rl_deconv* h, const double* estimate) {
    if (!h || !estimate) return fail(RL_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    RL_TRY(h->upload(estimate, h->est, (size_t)h->B * h->n_img()));
    h->est_ready = true;
    h->spec_valid = false;   // the next iterate rebuilds rowFFT(estimate)
    return RL_OK;
}

int rl_deconv_reset_estimate(rl_deconv* h) {
    if (!h) return fail(RL_ERR_INVALID, "handle is NULL");
    h->est_ready = false;
    return RL_OK;
}

int rl_deconv_iterate(rl_deconv* h, int k) {
    if (!h) return fail(RL_ERR_INVALID, "handle is NULL");
    if (k < 0) return fail(RL_ERR_INVALID, "k < 0");
    if (!h->have_meas) return fail(RL_ERR_STATE, "no measurement: call rl_deconv_simulate or rl_deconv_set_measurement");
    HIP_TRY(hipSetDevice(h->ctx->device));
    RL_TRY(h->refresh_meas_levels());
    HIP_TRY(hipEventRecord(h->ev0, h->ctx->stream));
    bool restart = !h->est_ready;
    if (!restart && !h->spec_valid && h->pair) {
        RL_TRY(h->row_pair(ROW_FWD, h->B, nullptr, h->spec_a, h->est, nullptr, nullptr));
        h->spec_valid = true;
    }
    if (!restart && !h->spec_valid && !h->sep) {   // H / H_t were called in between: rebuild rowFFT(est)
        RL_TRY(h->row(ROW_FWD, (unsigned)h->B, nullptr, h->spec_a, h->est, nullptr, nullptr));
        h->spec_valid = true;
    }
    RL_TRY(h->run_iterations(k, restart));
    HIP_TRY(hipEventRecord(h->ev1, h->ctx->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_iter_ms = ms;
    return RL_OK;
}

#define RL_GETTER(name, buf, count, need, what)                                   \
    int name(rl_deconv* h, double* out) {                                         \
        if (!h || !out) return fail(RL_ERR_INVALID, "NULL argument");             \
        if (!(need)) return fail(RL_ERR_STATE, what " is not available yet");     \
        HIP_TRY(hipSetDevice(h->ctx->device));                                    \
        return h->download(h->buf, out, (count));                                 \
    }
RL_GETTER(rl_deconv_get_object, obj, (size_t)h->B* h->n_img(), h->have_obj, "object")
RL_GETTER(rl_deconv_get_noiseless, noiseless, (size_t)h->B* h->V* h->n_img(), h->have_obj, "noiseless measurement")
RL_GETTER(rl_deconv_get_measurement, meas, (size_t)h->B* h->V* h->n_img(), h->have_meas, "measurement")
RL_GETTER(rl_deconv_get_estimate, est, (size_t)h->B* h->n_img(), h->est_ready, "estimate")
RL_GETTER(rl_deconv_get_normalization, norm, h->n_img(), true, "normalization")

int rl_forward(rl_deconv* h, const double* x, double* out) {
    if (!h || !x || !out) return fail(RL_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    // overwrites spec_a / spec_b; the estimate itself is kept (spec_a is rebuilt by the next iterate)
    void* xin = h->scratch;   // first B images of scratch
    RL_TRY(h->upload(x, xin, (size_t)h->B * h->n_img()));
    if (h->sep) {
        // xin lives in scratch.  Two passes: the row pass has consumed xin before the column pass writes scratch.
        // One kernel: its workgroups read halos of xin while others store -- the result goes to the (idle) spectrum buffer.
        void* res = h->sep_one ? h->sep_tmp() : h->scratch;
        RL_TRY(h->sep_forward(xin, res, h->B));
        return h->download(res, out, (size_t)h->B * h->V * h->n_img());
    }
    RL_TRY(h->row(ROW_FWD, (unsigned)h->B, nullptr, h->spec_a, xin, nullptr, nullptr));
    RL_TRY(h->col(h->spec_a, h->spec_b, h->B, true));
    RL_TRY(h->row(ROW_INV, (unsigned)(h->B * h->V), h->spec_b, nullptr, nullptr, h->scratch, nullptr));
    h->spec_valid = false;
    return h->download(h->scratch, out, (size_t)h->B * h->V * h->n_img());
}

int rl_adjoint(rl_deconv* h, const double* y, double* out, int normalize) {
    if (!h || !y || !out) return fail(RL_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    RL_TRY(h->upload(y, h->scratch, (size_t)h->B * h->V * h->n_img()));
    if (h->sep && h->sep_one) {
        RL_TRY(h->sep2d_(SEP_SUM_, h->scratch, nullptr, normalize ? h->norm : nullptr, h->spec_a, h->B));
        return h->download(h->spec_a, out, (size_t)h->B * h->n_img());
    }
    if (h->sep) {
        RL_TRY(h->sep_rows_(h->scratch, h->sep_tmp(), h->B * h->V, 1));
        RL_TRY(h->sep_cols_(SEP_SUM_, h->sep_tmp(), nullptr, normalize ? h->norm : nullptr, h->spec_a, h->B));
        return h->download(h->spec_a, out, (size_t)h->B * h->n_img());
    }
    // row transform of every view image: run ROW_FWD with V folded into the frame index
    RL_TRY(h->row(ROW_FWD, (unsigned)(h->B * h->V), nullptr, h->spec_b, h->scratch, nullptr, nullptr));
    RL_TRY(h->col(h->spec_b, h->spec_b, h->B, false));
    RL_TRY(h->row(ROW_ADJ, (unsigned)h->B, h->spec_b, nullptr, nullptr, h->scratch, normalize ? h->norm : nullptr));
    h->spec_valid = false;
    return h->download(h->scratch, out, (size_t)h->B * h->n_img());
}

int rl_deconv_last_ms(const rl_deconv* h, double* iterate_ms, double* simulate_ms) {
    if (!h) return fail(RL_ERR_INVALID, "handle is NULL");
    if (iterate_ms) *iterate_ms = h->last_iter_ms;
    if (simulate_ms) *simulate_ms = h->last_sim_ms;
    return RL_OK;
}

int rl_deconv_bench_cycles(rl_deconv* h, int k, int reps, int rng_kind, uint64_t seed, double* total_ms) {
    if (!h || !total_ms) return fail(RL_ERR_INVALID, "NULL argument");
    if (!h->have_obj) return fail(RL_ERR_STATE, "rl_deconv_set_object has not been called");
    if (k < 0 || reps < 1) return fail(RL_ERR_INVALID, "bad k / reps");
    HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    h->meas_level = h->obj_level;   // every cycle draws its measurement from the object
    h->choose_loop(h->meas_level);
    const auto t_start = std::chrono::steady_clock::now();
    HIP_TRY(hipEventRecord(h->ev0, s));
    for (int r = 0; r < reps; ++r) {
        // per slice of the batch: noiseless = H(obj), noisy = Poisson(noiseless) + 1e-9, est = 1,
        // k iterations -- the same values as rl_deconv_simulate + rl_deconv_iterate over the batch
        h->defer_join = r + 1 < reps;
        const int rc = h->run_cycle(k, rng_kind, seed + (uint64_t)r);
        h->defer_join = false;
        if (rc != RL_OK) {
            const std::string keep = rl::last_error();
            h->join_open_lanes();
            rl::last_error() = keep;
            return rc;
        }
        h->have_meas = true;
    }
    HIP_TRY(hipEventRecord(h->ev1, s));
    const auto t_enq = std::chrono::steady_clock::now();
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *total_ms = ms;
    if (getenv("RLSTED_DEBUG_ENQUEUE"))   // how far the host runs ahead of the device: enqueue time against device time
        fprintf(stderr, "rl_deconv_bench_cycles: host enqueue %.3f ms, device %.3f ms\n",
                std::chrono::duration<double, std::milli>(t_enq - t_start).count(), (double)ms);
    return RL_OK;
}


int rl_deconv_strategy(const rl_deconv* h, int* separable, int* real_psf_spectrum, int* split_column_pass, int* frame_pairs) {
    if (!h) return fail(RL_ERR_INVALID, "handle is NULL");
    if (frame_pairs) *frame_pairs = h->pair ? 1 : 0;
    if (separable) *separable = h->sep ? (h->sep_direct ? 2 : 1) : 0;
    if (real_psf_spectrum) *real_psf_spectrum = h->psf_hat_re ? 1 : 0;
    if (split_column_pass) *split_column_pass = !h->sep && !h->pair && h->col_split() ? 1 : 0;
    return RL_OK;
}

int rl_deconv_unresolved(rl_deconv* h, unsigned long long* count, int reset) {
    if (!h || !count) return fail(RL_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->ctx->device));
    // (the slices of a run are joined into the context's stream before a run returns: everything counted so far is ordered before this copy)
    HIP_TRY(hipMemcpyAsync(count, h->unresolved, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->ctx->stream));
    if (reset) HIP_TRY(hipMemsetAsync(h->unresolved, 0, sizeof(unsigned long long), h->ctx->stream));
    HIP_TRY(hipStreamSynchronize(h->ctx->stream));
    return RL_OK;
}

int rl_deconv_dims(const rl_deconv* h, int* batch, int* n_psf, int* ny, int* nx) {
    if (!h) return fail(RL_ERR_INVALID, "handle is NULL");
    if (batch) *batch = h->B;
    if (n_psf) *n_psf = h->V;
    if (ny) *ny = h->ny;
    if (nx) *nx = h->nx;
    return RL_OK;
}

int rl_batch_submit(rl_deconv* h, const rl_task* tasks, int n_tasks, int k_iters, int rng_kind, void* dev_out, int out_dtype) {
    if (!h || (!tasks && n_tasks > 0)) return fail(RL_ERR_INVALID, "NULL argument");
    if (n_tasks < 0 || k_iters < 0) return fail(RL_ERR_INVALID, "negative count");
    if (rng_kind != RL_RNG_NONE && rng_kind != RL_RNG_PHILOX) return fail(RL_ERR_INVALID, "unknown rng_kind");
    if (dev_out && out_dtype != RL_F32 && out_dtype != RL_F64) return fail(RL_ERR_INVALID, "out_dtype must be RL_F32 or RL_F64");
    for (int t = 0; t < n_tasks; ++t) {
        if (!tasks[t].object) return fail(RL_ERR_INVALID, "task without an object");
        if ((uint64_t)tasks[t].image_id * (uint64_t)h->V + (uint64_t)h->V > 0xffffffffull) return fail(RL_ERR_INVALID, "image id too large");
    }
    HIP_TRY(hipSetDevice(h->ctx->device));
    RL_TRY(h->ensure_batch_slots());
    const int B = h->B;
    const size_t n = h->n_img();
    hipStream_t s = h->ctx->stream;
    for (int t0 = 0; t0 < n_tasks; t0 += B) {
        const int nt = std::min(B, n_tasks - t0);
        rl_deconv::BatchSlot& sl = h->bslot[h->batch_chunks++ % 2];
        if (sl.used) HIP_TRY(hipEventSynchronize(sl.freed));   // the chunk before last has consumed this block
        double* tb = (double*)sl.host;
        uint64_t* seeds = (uint64_t*)(tb + B);
        uint32_t* ids = (uint32_t*)(seeds + B);
        uint32_t* idx = ids + B;
        double* objs = (double*)(sl.host + h->slot_header_bytes());
        bool scaled = true;
        std::vector<double> level((size_t)B, 0.0);
        std::vector<const double*> uniq;
        for (int f = 0; f < B; ++f) {   // a short last chunk repeats its last task (the plan's batch is fixed)
            const rl_task& t = tasks[t0 + std::min(f, nt - 1)];
            size_t u = 0;
            while (u < uniq.size() && uniq[u] != t.object) ++u;     // (a handful of distinct objects per chunk)
            if (u == uniq.size()) {
                uniq.push_back(t.object);
                memcpy(objs + u * n, t.object, n * sizeof(double));
            }
            idx[f] = (uint32_t)u;
            tb[f] = t.total_brightness;
            scaled = scaled && t.total_brightness > 0;
            seeds[f] = t.seed;
            ids[f] = t.image_id;
        }
        // the frames' levels (what pairs frames of comparable brightness, rl_deconv::choose_loop) are known on the host: the
        // targets, or -- unscaled objects -- their sums
        if (scaled) {
            for (int f = 0; f < B; ++f) level[f] = tb[f];
        } else {
            std::vector<double> usum(uniq.size(), 0.0);
            for (size_t u = 0; u < uniq.size(); ++u)
                for (size_t i = 0; i < n; ++i) usum[u] += objs[u * n + i];
            for (int f = 0; f < B; ++f) level[f] = usum[idx[f]];
        }
        const size_t used = h->slot_header_bytes() + uniq.size() * n * sizeof(double);
        HIP_TRY(hipMemcpyAsync(sl.dev, sl.host, used, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(hipEventRecord(sl.uploaded, h->copy_stream));
        HIP_TRY(hipStreamWaitEvent(s, sl.uploaded, 0));
        const double* d_tb = (const double*)sl.dev;
        const unsigned long long* d_seeds = (const unsigned long long*)(d_tb + B);
        const unsigned* d_ids = (const unsigned*)(d_seeds + B);
        const unsigned* d_idx = d_ids + B;
        const double* d_objs = (const double*)(sl.dev + h->slot_header_bytes());
        double* d_sums = (double*)(sl.dev + (h->slot_host_bytes() + 7) / 8 * 8);
        // :505-506  obj *= total_brightness / obj.sum(), per frame, on the device
        HIP_TRY(aux_scale_convert_indexed(h->dtype, d_objs, d_idx, uniq.size(), h->obj, n, (size_t)B, scaled ? d_tb : nullptr, d_sums, s));
        h->obj_level = level;
        h->meas_level = level;
        h->choose_loop(h->meas_level);
        h->have_obj = true;
        h->run_key_seeds = d_seeds;
        h->run_key_ids = d_ids;
        const int rc = h->run_cycle(k_iters, rng_kind, 0);   // per slice: H(obj), keyed Poisson draws, estimate = 1, k iterations
        h->run_key_seeds = nullptr;
        h->run_key_ids = nullptr;
        HIP_TRY(hipEventRecord(sl.freed, s));
        sl.used = true;
        RL_TRY(rc);
        h->have_meas = true;
        if (dev_out)
            HIP_TRY(aux_cast(h->dtype, h->est, out_dtype, (char*)dev_out + (size_t)t0 * n * esize(out_dtype), (size_t)nt * n, s));
    }
    return RL_OK;
}

int rl_batch_run(rl_deconv* h, const rl_task* tasks, int n_tasks, int k_iters, int rng_kind, double* estimates_out) {
    if (!h || (!tasks && n_tasks > 0)) return fail(RL_ERR_INVALID, "NULL argument");
    if (n_tasks < 0 || k_iters < 0) return fail(RL_ERR_INVALID, "negative count");
    HIP_TRY(hipSetDevice(h->ctx->device));
    const size_t need = (size_t)n_tasks * h->n_img() * esize(h->dtype);
    if (estimates_out && need > h->batch_out_bytes) {
        HIP_TRY(hipStreamSynchronize(h->ctx->stream));
        if (h->batch_out) HIP_TRY(hipFree(h->batch_out));
        h->batch_out = nullptr;
        h->batch_out_bytes = 0;
        HIP_TRY(hipMalloc(&h->batch_out, need));
        h->batch_out_bytes = need;
    }
    RL_TRY(rl_batch_submit(h, tasks, n_tasks, k_iters, rng_kind, estimates_out ? h->batch_out : nullptr, h->dtype));
    if (estimates_out && n_tasks > 0) return h->download(h->batch_out, estimates_out, (size_t)n_tasks * h->n_img());   // ONE download
    HIP_TRY(hipStreamSynchronize(h->ctx->stream));
    return RL_OK;
}

int rl_device_alloc(rl_ctx* ctx, size_t bytes, void** dev_out) {
    if (!ctx || !dev_out) return fail(RL_ERR_INVALID, "NULL argument");
    *dev_out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(dev_out, bytes ? bytes : 1));
    return RL_OK;
}

int rl_device_free(rl_ctx* ctx, void* dev) {
    if (!ctx) return fail(RL_ERR_INVALID, "ctx is NULL");
    if (!dev) return RL_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // nothing queued may still write it
    HIP_TRY(hipFree(dev));
    return RL_OK;
}

int rl_device_upload(rl_ctx* ctx, void* dev, int dtype, size_t n_elements, const double* host) {
    if (!ctx || (n_elements && (!dev || !host))) return fail(RL_ERR_INVALID, "NULL argument");
    if (dtype != RL_F32 && dtype != RL_F64) return fail(RL_ERR_INVALID, "dtype must be RL_F32 or RL_F64");
    HIP_TRY(hipSetDevice(ctx->device));
    if (dtype == RL_F64) {
        HIP_TRY(hipMemcpyAsync(dev, host, n_elements * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return RL_OK;
    }
    const size_t piece = std::min(n_elements, (size_t)16 << 20);
    double* w = nullptr;
    RL_TRY(ctx->psf_workspace(piece, &w));
    for (size_t o = 0; o < n_elements; o += piece) {
        const size_t m = std::min(piece, n_elements - o);
        HIP_TRY(hipMemcpyAsync(w, host + o, m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(aux_cast(RL_F64, w, RL_F32, (char*)dev + o * 4, m, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return RL_OK;
}

int rl_device_download(rl_ctx* ctx, const void* dev, int dtype, size_t n_elements, double* host_out) {
    if (!ctx || (n_elements && (!dev || !host_out))) return fail(RL_ERR_INVALID, "NULL argument");
    if (dtype != RL_F32 && dtype != RL_F64) return fail(RL_ERR_INVALID, "dtype must be RL_F32 or RL_F64");
    HIP_TRY(hipSetDevice(ctx->device));
    if (dtype == RL_F64) {
        HIP_TRY(hipMemcpyAsync(host_out, dev, n_elements * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return RL_OK;
    }
    // f32: widened on the device through the context's float64 scratch, in pieces of at most 128 MiB
    const size_t piece = std::min(n_elements, (size_t)16 << 20);
    double* w = nullptr;
    RL_TRY(ctx->psf_workspace(piece, &w));
    for (size_t o = 0; o < n_elements; o += piece) {
        const size_t m = std::min(piece, n_elements - o);
        HIP_TRY(aux_to_f64(RL_F32, (const char*)dev + o * 4, w, m, ctx->stream));
        HIP_TRY(hipMemcpyAsync(host_out + o, w, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return RL_OK;
}

int rl_deconv_device_ptr(rl_deconv* h, int which, void** ptr, size_t* n_elements, int* dtype) {
    if (!h || !ptr) return fail(RL_ERR_INVALID, "NULL argument");
    void* p = nullptr;
    size_t n = 0;
    switch (which) {
        case 0: p = h->est; n = (size_t)h->B * h->n_img(); break;
        case 1:
            p = h->meas;
            n = (size_t)h->B * h->V * h->n_img();
            // The caller may write a measurement straight into this buffer: what the host knows about the frames' levels (the
            // pairing guard, choose_loop) is then void -- the next run recomputes them from the buffer (refresh_meas_levels).
            h->meas_level.clear();
            h->meas_external = true;
            break;
        case 2: p = h->noiseless; n = (size_t)h->B * h->V * h->n_img(); break;
        case 3: p = h->obj; n = (size_t)h->B * h->n_img(); break;
        default: return fail(RL_ERR_INVALID, "which must be 0..3");
    }
    *ptr = p;
    if (n_elements) *n_elements = n;
    if (dtype) *dtype = h->dtype;
    return RL_OK;
}

int rl_deconv_time_cycle(rl_deconv* h, int k, int rng_kind, uint64_t seed, double* avg_ms, double* launches, double* frames_per_launch) {
    if (!h || !avg_ms) return fail(RL_ERR_INVALID, "NULL argument");
    if (!h->have_obj) return fail(RL_ERR_STATE, "rl_deconv_set_object has not been called");
    if (k < 0) return fail(RL_ERR_INVALID, "k < 0");
    if (h->sep) return fail(RL_ERR_UNSUPPORTED, "per-kernel timing covers the FFT strategy; this plan runs the separable stencils");
    HIP_TRY(hipSetDevice(h->ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    h->meas_level = h->obj_level;
    h->choose_loop(h->meas_level);
    h->timed.clear();
    h->events_used = 0;
    h->timing = true;
    int rc = h->run_cycle(k, rng_kind, seed);
    h->timing = false;
    hipError_t e = hipDeviceSynchronize();
    RL_TRY(rc);
    HIP_TRY(e);
    h->have_meas = true;
    double sum[rl_deconv::TK_COUNT] = {0}, cnt[rl_deconv::TK_COUNT] = {0};
    for (const auto& t : h->timed) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
        sum[t.kind] += ms;
        if (!t.cont) cnt[t.kind] += 1;
    }
    for (int i = 0; i < rl_deconv::TK_COUNT; ++i) {
        avg_ms[i] = cnt[i] > 0 ? sum[i] / cnt[i] : 0.0;
        if (launches) launches[i] = cnt[i];
    }
    avg_ms[7] = 0.0;   // (was the fused loop's slot; the arrays keep their 8 entries)
    if (launches) launches[7] = 0.0;
    if (frames_per_launch) *frames_per_launch = (double)h->chunk_frames();
    return RL_OK;
}

int rl_deconv_time_kernels(rl_deconv* h, int reps, double* avg_ms) {
    if (!h || !avg_ms) return fail(RL_ERR_INVALID, "NULL argument");
    if (!h->have_meas) return fail(RL_ERR_STATE, "no measurement");
    if (reps < 1) return fail(RL_ERR_INVALID, "reps < 1");
    if (h->sep) return fail(RL_ERR_UNSUPPORTED, "per-kernel timing covers the FFT strategy; this plan runs the separable stencils");
    HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    if (!h->est_ready) RL_TRY(h->start_estimate());
    // one untimed iteration so that every buffer holds realistic data
    RL_TRY(h->iterate_once());
    // the RL kernels are timed on the launch shape the RL loop uses: one slice of the batch
    const int nf = h->chunk_frames();
    avg_ms[6] = (double)nf;
    for (int which = 0; which < 6; ++which) {
        HIP_TRY(hipEventRecord(h->ev0, s));
        for (int r = 0; r < reps; ++r) {
            const bool one_buffer = h->V == 1 && h->inplace;   // as iterate_chunk(): everything in spec_a
            switch (which) {
                case 0:
                    if (h->col_split()) RL_TRY(h->col_split_pass(h->spec_a, h->spec_b, h->spec_x, nf, rl_deconv::COL_H));
                    else RL_TRY(h->col(h->spec_a, one_buffer ? h->spec_a : h->spec_b, nf, true));
                    break;
                case 1:
                    if (one_buffer) RL_TRY(h->row(ROW_RATIO, (unsigned)nf, h->spec_a, h->spec_a, h->meas, nullptr, nullptr));
                    else RL_TRY(h->row(ROW_RATIO, (unsigned)(nf * h->V), h->spec_b, h->spec_b, h->meas, nullptr, nullptr));
                    break;
                case 2:   // as iterate_chunk(): in place, fused (Fourier-domain view sum) or per view
                    if (one_buffer) RL_TRY(h->col(h->spec_a, h->spec_a, nf, false));
                    else if (h->col_split() && h->fuse_views) RL_TRY(h->col_split_pass(h->spec_b, h->spec_a, h->spec_x, nf, rl_deconv::COL_HT_FUSED));
                    else if (h->fuse_views && h->V > 1 && h->wave_private_y()) RL_TRY(h->col(h->spec_b, h->spec_a, nf, rl_deconv::COL_HT_FUSED));
                    else RL_TRY(h->col(h->spec_b, h->spec_b, nf, false));
                    break;
                case 3:
                    if (one_buffer) RL_TRY(h->row(ROW_UPDATE, (unsigned)nf, h->spec_a, h->spec_a, nullptr, h->est, h->norm));
                    else if (h->fuse_views && h->V > 1 && (h->wave_private_y() || h->col_split()))
                        RL_TRY(h->row(ROW_UPDATE, (unsigned)nf, h->spec_a, h->spec_a, nullptr, h->est, h->norm, nullptr, 1));
                    else RL_TRY(h->row(ROW_UPDATE, (unsigned)nf, h->spec_b, h->spec_a, nullptr, h->est, h->norm));
                    break;
                case 4: RL_TRY(h->row(ROW_FWD, (unsigned)h->B, nullptr, h->spec_a, h->obj, nullptr, nullptr)); break;
                case 5: HIP_TRY(aux_poisson(h->dtype, h->noiseless, h->meas, (unsigned)h->n_img(), (unsigned)(h->B * h->V), 0, 1, RL_RNG_PHILOX, h->scratch, s)); break;
            }
        }
        HIP_TRY(hipEventRecord(h->ev1, s));
        HIP_TRY(hipEventSynchronize(h->ev1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        avg_ms[which] = (double)ms / reps;
    }
    // the repeated launches trashed the RL state on purpose; force a clean restart
    h->est_ready = false;
    h->spec_valid = false;
    return RL_OK;
}

}  // extern "C"

