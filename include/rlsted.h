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

const char* rl_last_error(void);
int rl_version(void);
int rl_device_count(int* count);

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
/* Inject a measurement [batch][n_psf][ny][nx] instead (load_data_from_tif
 * :514-518, or noise drawn on the host with numpy for figure reproduction).  */
int rl_deconv_set_measurement(rl_deconv* h, const double* noisy);

/* iterate (:520-531) K times; the first call starts from estimate = 1.       */
int rl_deconv_iterate(rl_deconv* h, int k);
int rl_deconv_reset_estimate(rl_deconv* h);

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

/* Per-kernel device time: launches each kernel of the RL iteration `reps`
 * times back to back between two hipEvents on the plan's stream and returns
 * the average milliseconds per launch in avg_ms[6] = { column pass (H),
 * row pass RATIO, column pass (H_t), row pass UPDATE, row pass FWD, Poisson }.
 * Destroys the current estimate (the next iterate restarts from 1).          */
int rl_deconv_time_kernels(rl_deconv* h, int reps, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* RLSTED_H */
