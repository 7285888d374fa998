/* rlsted.h -- C ABI of librlsted.so: MI355X (gfx950) implementation of the
 * rescan line-STED image-formation + Richardson-Lucy hot path.
 *
 * The reference (AndrewGYork/rescan_line_sted) has no FFI or plugin interface
 * for this path: its boundary is the Python module namespace
 * figure_generation/line_sted_tools.py.  Each entry point below states which
 * reference function (file:line) it stands in for; the Python mirror of that
 * module (rescan_line_sted_amd/line_sted_tools.py) binds them with ctypes (see
 * INTEGRATION.md).
 *
 * Conventions: every function returns 0 on success and a negative code on
 * failure, after which rl_last_error() describes the failure (thread local).
 * No exceptions cross the boundary.  Host buffers are caller owned, row major,
 * C contiguous.  Handles are opaque.  One call in flight per context.
 *
 * All image-like host arrays are float64 (the reference's dtype); the
 * arithmetic type on the device is chosen per plan (RL_F32 / RL_F64).
 */
#ifndef RLSTED_H
#define RLSTED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RL_OK 0
#define RL_ERR_INVALID -1     /* bad argument */
#define RL_ERR_HIP -2         /* HIP runtime error (no device, launch failure, OOM...) */
#define RL_ERR_UNSUPPORTED -3 /* size / mode not built */
#define RL_ERR_STATE -4       /* call sequence error (e.g. iterate before data) */

#define RL_F32 0
#define RL_F64 1

/* Poisson generator selection for rl_deconv_simulate */
#define RL_RNG_NONE 0      /* noisy = noiseless + 1e-9 (no noise)                       */
#define RL_RNG_PHILOX 1    /* device Philox4x32-10 counter RNG, bit-exact vs oracle twin */

typedef struct rl_ctx rl_ctx;
typedef struct rl_deconv rl_deconv;
typedef struct rl_comm rl_comm;

const char* rl_last_error(void);
int rl_version(void);
int rl_device_count(int* count);

/* Page-locked host memory for the buffers handed to the calls below: the float64 arrays of the reference's
 * call surface (line_sted_tools.py:496-531: obj, noisy_measurement, estimate are host numpy arrays) then cross
 * PCIe by direct DMA instead of through the runtime's pageable-copy staging.
 * Purely optional: every entry point accepts ordinary host pointers.                                        */
int rl_host_alloc(size_t bytes, void** out);
int rl_host_free(void* p);

/* One context per GPU: owns the stream and the per-length twiddle tables. */
int rl_ctx_create(int device, rl_ctx** out);
int rl_ctx_destroy(rl_ctx* ctx);
int rl_ctx_synchronize(rl_ctx* ctx);

/* Smallest supported transform length >= n (0 if none). */
int rl_fft_length_for(int n);

/* ---- Deconvolver: line_sted_tools.py:478-594 -------------------------------
 * A plan holds `batch` independent frames that share one PSF set (`n_psf`
 * views of shape (1, py, px)) and one image shape (ny, nx).  Replaces
 * Deconvolver.__init__ (:479-494) plus the lazily cached H_t normalisation
 * (:589-592), which is computed here once.                                   */
int rl_deconv_create(rl_ctx* ctx, const double* psfs, int n_psf, int py, int px,
                     int batch, int ny, int nx, int dtype, rl_deconv** out);
int rl_deconv_destroy(rl_deconv* h);

/* Geometry chosen by the plan: ly, lx transform lengths, pitch of spectra. */
int rl_deconv_info(const rl_deconv* h, int* ly, int* lx, int* pitch, size_t* device_bytes);

/* create_data_from_object (:496-512), first half: copy obj [batch][ny][nx],
 * scale each frame so that its sum is total_brightness[f] (NULL: no scaling),
 * noiseless = H(obj).                                                        */
int rl_deconv_set_object(rl_deconv* h, const double* obj, const double* total_brightness);
/* create_data_from_object (:508-511), second half: noisy = Poisson(noiseless)
 * + 1e-9 drawn on the device.  Counter layout: see DESIGN.md "Device Poisson". */
int rl_deconv_simulate(rl_deconv* h, int rng_kind, uint64_t seed);

/* As rl_deconv_simulate, with a Philox key per frame: frame f (all its views v) draws with seed
 * seeds[f] and image index image_ids[f] * n_psf + v (host arrays of n_frames entries).  A frame's
 * noise then depends on (seed, image id, pixel) only -- not on the batch it is simulated in, which
 * is what lets a parameter sweep pack tasks with different seeds into one plan.  With seeds[f] = s
 * and image_ids[f] = f this is rl_deconv_simulate(h, rng_kind, s).                              */
int rl_deconv_simulate_keyed(rl_deconv* h, int rng_kind, const uint64_t* seeds, const uint32_t* image_ids);
/* Inject a measurement [batch][n_psf][ny][nx] instead (load_data_from_tif
 * :514-518, or noise drawn on the host with numpy for figure reproduction).
 * New data (this call, rl_deconv_simulate, rl_deconv_simulate_keyed) starts a new Richardson-Lucy run: the
 * next rl_deconv_iterate begins from ones and rl_deconv_get_estimate fails (RL_ERR_STATE) until then.  The
 * reference keeps its estimate across create_data_from_object (:496-531, it only resets at num_iterations
 * == 0); a caller that wants that reads the estimate before the new data and hands it back with
 * rl_deconv_set_estimate afterwards -- the Deconvolver mirror does (line_sted_tools.py).                 */
int rl_deconv_set_measurement(rl_deconv* h, const double* noisy);

/* iterate (:520-531) K times; the first call starts from estimate = 1.       */
int rl_deconv_iterate(rl_deconv* h, int k);
int rl_deconv_reset_estimate(rl_deconv* h);
/* Deconvolver.estimate is a plain attribute in the reference (:522,530): assigning to it, or replacing
 * the measurement after some iterations (create_data_from_object called again, :496-512), continues
 * from that estimate.  Uploads estimate [batch][ny][nx]; the next rl_deconv_iterate continues from it. */
int rl_deconv_set_estimate(rl_deconv* h, const double* estimate);

int rl_deconv_get_object(rl_deconv* h, double* out);        /* [batch][ny][nx]        */
int rl_deconv_get_noiseless(rl_deconv* h, double* out);     /* [batch][n_psf][ny][nx] */
int rl_deconv_get_measurement(rl_deconv* h, double* out);   /* [batch][n_psf][ny][nx] */
int rl_deconv_get_estimate(rl_deconv* h, double* out);      /* [batch][ny][nx]        */
int rl_deconv_get_normalization(rl_deconv* h, double* out); /* [ny][nx]               */

/* H (:567-577): x [batch][ny][nx] -> out [batch][n_psf][ny][nx].             */
int rl_forward(rl_deconv* h, const double* x, double* out);
/* H_t (:579-594): y [batch][n_psf][ny][nx] -> out [batch][ny][nx].           */
int rl_adjoint(rl_deconv* h, const double* y, double* out, int normalize);

/* Device time (hipEvents on the plan's stream) of the last rl_deconv_iterate /
 * rl_deconv_set_object+simulate call, in milliseconds.                       */
int rl_deconv_last_ms(const rl_deconv* h, double* iterate_ms, double* simulate_ms);

/* Benchmark entry: run `reps` x (simulate + k iterations) on device-resident
 * data with no host transfers inside the timed region; returns total device ms
 * measured with hipEvents on the plan's stream.                              */
int rl_deconv_bench_cycles(rl_deconv* h, int k, int reps, int rng_kind, uint64_t seed, double* total_ms);

/* Device address of a plan buffer for zero-copy hand-off (e.g. the RCCL gather
 * of final estimates): which = 0 estimate [batch][ny][nx], 1 measurement, 2
 * noiseless [batch][n_psf][ny][nx], 3 object.  The buffer stays owned by the
 * plan; *dtype = RL_F32 / RL_F64 element type.  Synchronise the context first.
 * The allocation extends 16 KiB past n_elements (zeroed slack that the row kernels' unconditional 64-lane
 * loads may read and discard): never write there, never assume the next buffer starts right behind. */
int rl_deconv_device_ptr(rl_deconv* h, int which, void** ptr, size_t* n_elements, int* dtype);

/* Convolution strategy the plan chose for its PSF set: separable == 1: every view
 * is rank 1 (p = u v^T; the 0 / 90 degree line PSFs) and small, H / H_t run as direct row + column stencils;
 * separable == 2: the views are small but not rank 1 and H / H_t run as a direct 2-D stencil (py * px multiply-adds
 * per pixel; RLSTED_DIRECT) -- like the separable form it keeps the relative accuracy of a dark region's prediction;
 * otherwise the FFT path, with real_psf_spectrum != 0 when the (point-symmetric) PSFs' spectra are real and
 * the column kernels multiply by their real parts alone; split_column_pass != 0: a multi-view f32 plan on the long
 * column transforms (L = 2304, 4608), whose column passes are two launches each -- forward half, inverse half, the
 * column spectra parked between them (this slot reported the removed fused kernel of round 2 and was always 0);
 * frame_pairs != 0: the Richardson-Lucy loop transforms frames 2p and 2p+1 as the real and imaginary part of one
 * complex image (single-view f32 plans by default: RLSTED_PAIR) -- a frame's estimate then depends on its partner
 * at f32 rounding level (~1e-7 of the brighter partner's scale).  The answer is the loop that will run on the
 * CURRENT data: a plan built with pairs runs its per-frame loop while any pair's frames differ in level (sum of the
 * object / measurement) by more than a factor of 4 (RLSTED_PAIR_MAX_RATIO), because a dim frame would inherit
 * the rounding error of a bright partner.  Any of the pointers may be NULL.                                  */
int rl_deconv_strategy(const rl_deconv* h, int* separable, int* real_psf_spectrum, int* split_column_pass, int* frame_pairs);

/* Predictions the plan could not resolve.  iterate divides the measurement by H(estimate) clamped at 0 (line_sted_tools.py:575,
 * 524); in exact arithmetic that prediction is positive, but a transform resolves a value to eps * the frame's maximum only, so
 * on sparse emitters over a black background the predictions of the dark region are rounding noise of either sign -- below 1e-7
 * of the maximum for an f32 plan, 1e-16 for a float64 plan (and for the reference, which then divides by zero: inf, nan, the
 * frame lost).  The kernels treat a pixel whose prediction is not positive as neutral (ratio 1) and count it: *count = the lanes
 * of the FFT path's ratio launches that met such a pixel inside the image since the plan was created or the counter last reset
 * (reset != 0 clears it).  0 on data the plan's arithmetic resolves; an f32 plan that counts should be a float64 plan -- its
 * estimates stay finite and non-negative but are no longer within 1e-5 of the float64 result.  The count is a sufficient sign, not
 * a necessary one: rounding noise that happens to be positive everywhere is not counted (a frame of isolated photons, well below
 * one per pixel, can pass with 0).  Synchronises the plan's stream.                                                              */
int rl_deconv_unresolved(rl_deconv* h, unsigned long long* count, int reset);

/* Plan geometry: frames per plan, views per frame, image shape.                */
int rl_deconv_dims(const rl_deconv* h, int* batch, int* n_psf, int* ny, int* nx);

/* ---- parameter sweeps: line_sted_figure_2.py:39-57 ---------------------------
 * The reference's figure script builds one Deconvolver per (PSF set, test image) and runs
 * create_data_from_object + N x iterate on each.  A task is one such simulation for a plan's PSF
 * set and image shape: an object [ny][nx] (host, float64), its total brightness (:505-506; <= 0:
 * no scaling -- then for every task of the call), and the Philox key (seed, image_id) of its noise
 * (rl_deconv_simulate_keyed).  rl_batch_run works through n_tasks tasks in chunks of the plan's
 * batch: objects -> H -> Poisson -> k_iters Richardson-Lucy iterations from estimate = 1, and
 * writes the estimates [n_tasks][ny][nx] to estimates_out (NULL: the last chunk stays in the
 * plan's buffers for rl_gather).  A task's noiseless and noisy measurements do not depend on its position in the
 * list (per-task Philox key); its estimate does not either in f64 plans, and in f32 plans only at rounding level
 * (~1e-7: frame pairs share a transform with their neighbour in the batch -- and only with a neighbour of
 * comparable level, see rl_deconv_strategy).                                                                  */
typedef struct rl_task {
    const double* object;
    double total_brightness;
    uint64_t seed;
    uint32_t image_id;
} rl_task;
int rl_batch_run(rl_deconv* h, const rl_task* tasks, int n_tasks, int k_iters, int rng_kind, double* estimates_out);

/* The same work, ENQUEUED: returns once the tasks' objects are staged (page-locked buffers of the plan, two chunks deep) and
 * every copy and kernel is in the context's stream order -- chunk i + 1 is uploaded on a copy stream while chunk i iterates, and
 * successive calls, on this plan or on other plans of the context, follow each other on the device without the host in
 * between.  The estimates go to DEVICE memory: dev_out [n_tasks][ny][nx] of out_dtype (RL_F32 / RL_F64; rl_device_alloc, or
 * any device pointer of this GPU), NULL: nowhere (the last chunk stays in the plan's buffers).  Nothing of the result may be
 * read, and the plan not destroyed, before rl_ctx_synchronize.  rl_batch_run is this call with a buffer of its own followed by
 * ONE download.  line_sted_figure_2.py:39-57 is a loop of such runs: one per (PSF set, test image).                       */
int rl_batch_submit(rl_deconv* h, const rl_task* tasks, int n_tasks, int k_iters, int rng_kind, void* dev_out, int out_dtype);

/* Device memory owned by the caller: the result buffer of a sweep (rl_batch_submit), the operands of rl_comm_gather_device.
 * rl_device_download: n elements of `dtype` -> host float64 (blocking; synchronises the context first).                   */
int rl_device_alloc(rl_ctx* ctx, size_t bytes, void** dev_out);
int rl_device_free(rl_ctx* ctx, void* dev);
int rl_device_download(rl_ctx* ctx, const void* dev, int dtype, size_t n_elements, double* host_out);
/* host float64 -> n elements of `dtype` at a device address of this GPU (a caller's buffer, or a plan buffer handed out by
 * rl_deconv_device_ptr -- the zero-copy way to replace a measurement: the plan re-measures its frames' levels before the next run). */
int rl_device_upload(rl_ctx* ctx, void* dev, int dtype, size_t n_elements, const double* host);

/* ---- multi-GPU: one process per GPU, frames sharded over ranks -----------------
 * The reference is single process; independent simulations shard with no
 * data-path collective and ONE gather of the results at the end (section 8e).  RCCL over xGMI,
 * loaded on first use.  rl_comm_unique_id on one rank produces RL_COMM_ID_BYTES bytes that every
 * rank passes to rl_comm_create (a collective call: all ranks of the world enter it together).   */
#define RL_COMM_ID_BYTES 128
int rl_comm_unique_id(void* id128);
int rl_comm_create(rl_ctx* ctx, int rank, int world, const void* id128, rl_comm** out);
int rl_comm_destroy(rl_comm* c);
int rl_comm_info(const rl_comm* c, int* rank, int* world);
/* device-synchronise this rank, then meet every other rank (the timing harness' barrier) */
int rl_comm_barrier(rl_comm* c);
/* *value = max over ranks of *value (every rank gets the result)                       */
int rl_comm_allreduce_max(rl_comm* c, double* value);
/* Gather the first counts[r] frames of every rank r's plan buffer `which` (as in
 * rl_deconv_device_ptr) on `root`, rank-major, straight from the device buffers: every rank sends
 * to the root, the root receives on all its links at once.  rl_gather delivers float64 on the
 * root's host (host_out: sum(counts) frames; ignored on other ranks); rl_gather_device leaves the
 * result in a device buffer owned by the communicator (root: *dev_out, valid until the next
 * rl_gather / rl_gather_device of this communicator -- rl_comm_gather_host stages through a buffer of its own and
 * leaves it alone; other ranks: NULL) in the plan's dtype.                                 */
/* The sweep's one gather: counts[r] elements of `dtype` from every rank r's DEVICE buffer dev_local, rank-major
 * into dev_out on the root (device memory of the root's GPU holding sum(counts) elements; ignored elsewhere) -- results of
 * different shapes travel unpadded, in the plan's own arithmetic type.  Synchronises the device on both sides of the transfer. */
int rl_comm_gather_device(rl_comm* c, const void* dev_local, const size_t* counts, int dtype, int root, void* dev_out);
/* n float64 values from `root`'s host buffer into every rank's (the PSF sets of a sweep are built once -- the reference builds
 * them once per figure, line_sted_figure_2.py:66-72 -- and sent to the other ranks).  Collective.                          */
int rl_comm_bcast_host(rl_comm* c, double* buf, size_t n, int root);
int rl_gather(rl_comm* c, rl_deconv* plan, int which, int root, const int* counts, double* host_out);
int rl_gather_device(rl_comm* c, rl_deconv* plan, int which, int root, const int* counts, void** dev_out,
                     size_t* n_elements, int* dtype);
/* The same for host arrays (results that no longer live in a plan, e.g. a sweep's stack of frames
 * of several shapes): counts[r] float64 values of rank r's `local`, rank-major into `out` on root. */
int rl_comm_gather_host(rl_comm* c, const double* local, const size_t* counts, int root, double* out);

/* ---- PSF generation: line_sted_tools.py:75-363, 653-668 ---------------------
 * get_width (:653-668): MINPACK-lmdif fit of A*exp(-(x-mu)^2/(2 sigma^2)) to
 * y[0..n-1] from [1, n/2, 1] with scipy.optimize.curve_fit's defaults; host
 * code, needs no GPU.  p3 = {A, mu, sigma}; *info = MINPACK info (may be NULL). */
int rl_gauss_fit(const double* y, int n, double* p3, int* info);

/* scipy.ndimage.gaussian_filter as the reference uses it (:185-213,260,280):
 * float64, mode 'reflect', radius int(truncate*sigma+0.5), axes with sigma <=
 * 1e-15 skipped.  in/out: host [nz][ny][nx].                                   */
int rl_gaussian_filter(rl_ctx* ctx, const double* in, double* out, int nz, int ny, int nx,
                       const double* sigma3, double truncate);

/* generate_psfs (:168-363) for shape (1, ny, nx).  psf_type 0 = 'point', 1 =
 * 'line'.  rescan_ratio > 0 forces the integer line rescan ratio, <= 0 derives
 * it from the fitted width of the central sted row (:252-256).
 * arrays_out: NULL or [5 (point) | 7 (line)][ny][nx] = excitation, depletion,
 *   excitation_fraction, depletion_fraction, sted, descan_sted, rescan_sted.
 * rows_out:   NULL or [3][nx] central rows of excitation, sted, rescan_sted.
 * scalars_out[10]: 0 ratio used, 1 ideal ratio, 2-4 area sums of excitation /
 *   depletion / sted, 5-7 central-row sums of the same, 8 = 1 when every
 *   central-row maximum equals its array maximum (asserts :105-106,120).      */
int rl_psf_generate(rl_ctx* ctx, int psf_type, int ny, int nx, double excitation_brightness,
                    double depletion_brightness, double blur_sigma, int rescan_ratio,
                    double* arrays_out, double* rows_out, double* scalars_out);

/* The two intermediate arrays of the 'line' branch that generate_psfs(output_dir=...) also writes (:339-341):
 * emission_psf_out [ny][nx] = gaussian_filter(centred delta, blur_sigma) (:258-260) and rescan_unscaled_out
 * [ny][rescan_ratio * nx] = rescanned_signal_cumu, the detector ring before it is rolled and binned (:266-298).
 * rescan_ratio: the integer ratio a previous rl_psf_generate of the same parameters reported (scalars_out[0]); it
 * sizes the second buffer.  Either pointer may be NULL.                                                      */
int rl_psf_generate_line_extras(rl_ctx* ctx, int ny, int nx, double excitation_brightness, double depletion_brightness,
                                double blur_sigma, int rescan_ratio, double* emission_psf_out, double* rescan_unscaled_out);

/* psf_report (:75-166).  report_out[8] = { resolution_improvement_descanned,
 * resolution_improvement_rescanned (NaN for 'point'), excitation_dose,
 * depletion_dose, expected_emission, num_steps n, line rescan ratio, invariant
 * flag as scalars_out[8] above }.  arrays_out as in rl_psf_generate with ny = nx
 * = n = 1 + 2*round(5*sigma) (NULL: scalars only).                             */
int rl_psf_report(rl_ctx* ctx, int psf_type, double excitation_brightness, double depletion_brightness,
                  double steps_per_excitation_psf_width, double pulses_per_position,
                  double* arrays_out, double* report_out);

/* psf_report for n_sets parameter sets in one call (the sweeps of line_sted_figure_1.py:33-48 and
 * line_sted_figure_a1.py:29,64,102,172): one launch per pipeline stage over all sets, the Gaussian fits of
 * all sets on the host in between.  params[n_sets][5] = { psf_type (0 / 1), excitation_brightness,
 * depletion_brightness, steps_per_excitation_psf_width, pulses_per_position }; report_out[n_sets][8] as
 * rl_psf_report, bit for bit; arrays_out: NULL, or n_sets pointers, each NULL or [5 | 7][n][n].        */
int rl_psf_report_batch(rl_ctx* ctx, int n_sets, const double* params, double* report_out, double* const* arrays_out);

/* rotate of line_sted_figure_2.py:264-272 for one [ny][nx] plane (general angles; the
 * caller keeps the script's exact 0 and 90 degree special cases): cubic B-spline
 * rotation about the centre as scipy.ndimage.rotate(order=3, reshape=False), then
 * clipped to [0, 1.1 * max(in)].  Host in / out, float64.                       */
int rl_rotate_psf(rl_ctx* ctx, const double* in, double* out, int ny, int nx, double degrees);
/* The same for a stack [nz][ny][nx] in one call: every plane rotated in its plane, all of them clipped to
 * [0, 1.1 * max(in)] with the maximum of the WHOLE array, as the script's np.clip does (fig2:271).      */
int rl_rotate_psf_stack(rl_ctx* ctx, const double* in, double* out, int nz, int ny, int nx, double degrees);

/* ---- reconstruction quality (line_sted_tools.py:539-547, line_sted_figure_2.py:353-390) ----
 * out[i] = f(|fftshift(fft2(x[i]))| * scale) for n_img real images [n_img][ny][nx] of any
 * size, f(m) = log(1 + m) when log1p != 0, else m.  record_iteration's FT-error history is
 * (x = estimate_k - truth, scale 1, log1p 1); fourier_error of the figure-2 harness
 * (:353-355) is (x = estimate - truth, scale 1/(ny*nx), log1p 0).                      */
int rl_fft2_magnitude(rl_ctx* ctx, const double* x, int n_img, int ny, int nx, double scale, int log1p,
                      double* out);

/* scipy.ndimage.map_coordinates(image, [ys, xs]) with its defaults (order 3, mode 'constant',
 * cval 0, prefilter) for a [ny][nx] image at n points (:381-384): out[n].               */
int rl_spline_sample(rl_ctx* ctx, const double* image, int ny, int nx, const double* ys, const double* xs, int n,
                     double* out);

/* ---- line_sted_figure_3.py: the scan-position-by-scan-position imaging simulator (:76-273) ----
 * rl_rotate_image: `rotate` (:382-391) for one [ny][nx] plane -- scipy.ndimage.rotate(order 3,
 * mode 'nearest', reshape=False) about the centre; clip != 0 clips to [0, 1.1 * max(in)].  Host in / out. */
int rl_rotate_image(rl_ctx* ctx, const double* in, double* out, int ny, int nx, double degrees, int clip);

/* One orientation's scan (:172-239), all scan positions in one call, batched on the device.
 * rot_obj, centered_exc: [ny][nx] (the padded, rotated object :169; the blurred excitation :139).
 * positions: [n_pos][2] = (shift_y, shift_x) (:112-137).  display: ascending indices of the positions
 * whose detector images are wanted (the frames the reference renders, :258-263).
 * pos_scalars [n_pos][4] = { glow.max(), inst_detector_sig.max(), cum_detector_sig.max(),
 *   inst_detector_sig.sum() } per position (:252-256; the sum is descan_point's reconstruction value :200).
 * pos_values: descan_line [n_pos][nx] = inst_detector_sig.sum(axis=1) (:192); nondescan_multipoint
 *   [n_pos][ceil(n_y/exc_sep) * ceil(n_x/exc_sep)] = the region sums (:213-220), y major; else unused.
 * display_out [n_display][2][ny][nx] = inst_detector_sig, cum_detector_sig of the display positions.
 * cum_final [ny][nx]: rescan_line's accumulated detector image (:234), else unused.              */
typedef struct rl_fig3_params {
    int imaging_type;    /* 0 descan_point, 1 nondescan_multipoint, 2 descan_line, 3 rescan_line */
    int ny, nx;          /* padded shape */
    int n_y, n_x, pad;   /* object shape and padding (:106-107) */
    int step, exc_sep;   /* scan step (:102); spot separation (:131, multipoint) */
    double psf_sigma;    /* detection blur (:100) */
    double rescan_scale; /* 1 / (R^2 + 1) (:229), rescan_line */
} rl_fig3_params;
int rl_fig3_scan(rl_ctx* ctx, const rl_fig3_params* p, const double* rot_obj, const double* centered_exc,
                 const int* positions, int n_pos, const int* display, int n_display, double* pos_scalars,
                 double* pos_values, double* display_out, double* cum_final);

/* Per-kernel device time: launches each kernel of the RL iteration `reps`
 * times back to back between two hipEvents on the plan's stream and returns
 * the average milliseconds per launch in avg_ms[0..5] = { column pass (H),
 * row pass RATIO, column pass (H_t), row pass UPDATE, row pass FWD, Poisson };
 * avg_ms[6] = frames covered by one launch of the four RL kernels (the RL
 * loop works through the batch in equal slices sized for the Infinity Cache;
 * FWD and Poisson run over the whole batch).  avg_ms must hold 7 doubles.
 * Destroys the current estimate (the next iterate restarts from 1).          */
int rl_deconv_time_kernels(rl_deconv* h, int reps, double* avg_ms);

/* In-situ kernel durations of ONE whole cycle (simulate + k iterations, as rl_deconv_bench_cycles
 * runs it): a hipEvent pair around every launch, recorded on the stream the launch goes to, with
 * the batch slices overlapping on their streams as in production -- the figure a rocprofv3 kernel
 * trace of the same run reports.  avg_ms[8] / launches[8] (may be NULL): average duration and
 * number of launches of { column pass (H), row pass RATIO, column pass (H_t), row pass UPDATE, row
 * pass FWD, row pass INV, Poisson (both kernels), unused (0) };
 * *frames_per_launch: frames one launch of the RL kernels covers.                        */
int rl_deconv_time_cycle(rl_deconv* h, int k, int rng_kind, uint64_t seed, double* avg_ms, double* launches,
                         double* frames_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* RLSTED_H */
