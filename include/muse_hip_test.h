/*
 * muse_hip_test.h -- TEST AND MEASUREMENT HOOKS of libmuse_hip.so.  Not part of the drop-in boundary: nothing a
 * go-muse host binds lives here (INTEGRATION.md binds include/muse_hip.h only).  Used by tests/, tools/ and bench.py.
 */
#ifndef MUSE_HIP_TEST_H
#define MUSE_HIP_TEST_H

#include "muse_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Forces the kernel of the all-scores pass (muse_batch_score): 0 = automatic (default), 1 = generic radix-2 kernel
 * (any power-of-two n), 7 = the n = 4096 kernel that rescales both series before the shared transform (the hand-off
 * target of the default kernel), 10 = the default n = 4096 kernel, 11 = round-1 radix-16 Stockham / four-step kernels
 * (n = 512 .. 2048, 8192 .. 65536), 12 = the pair-packed half-round kernels (n = 512 .. 2048: what auto takes; 8192, 16384; also the
 * two-sided xCorr at n = 16384), 13 = the four-step long-series kernel (n = 16384 .. 65536), 14 = one REAL series per workgroup
 * (n = 8192 .. 65536: what auto takes at 8192, 16384 and 65536), 15 = the same at n = 32768 with each 16384-point transform as
 * 16 x 1024 (wave-local 1024-point transforms around one workgroup transpose: what auto takes at 32768).  The parity tests run every kernel on the same inputs. */
int muse_ctx_set_kernel(muse_ctx *ctx, int32_t variant);
/* Scales the error bound the filter-and-refine Run assumes for its fp32 estimates (1.0 = the derived bound): the
 * guard test shrinks it a million-fold to force the fp64 re-run. */
int muse_test_set_screen_bound_scale(muse_ctx *ctx, double scale);
/* The bound itself, in the pass's scaled units, for FFT length n and max |X| (host only; docs/screen_error_bound.md). */
int muse_test_screen_bound(int32_t n, double xmax, double *Es);
/* Runs the screening pass of the filter-and-refine Run alone (MaxLag = max_lag, TopN = 1, no other filter) over a
 * batch of series of length 257 .. 65536 and returns, per series, the fp32 estimate of the signed score, the pass's
 * flag word (bit 0 / 1: a possible argmax has |lag| <= / > max_lag; bit 2 / 3: a possible argmax value is > 0 / < 0;
 * bit 4: fp32 not trusted, must be re-evaluated; bit 5: the exact score is NaN; bit 31: the row was re-evaluated and
 * `estimate` holds its fp64 score) and the bound *E (score units) that the selection assumes on
 * |estimate - exact score|.  Any out pointer may be NULL. */
int muse_batch_screen_estimates(muse_batch *b, int32_t max_lag, double *estimate, uint32_t *flags, double *E);

/* Shader clock held under load (bench.py's roofline.co_bounds): starts a one-wave kernel on a stream of its own that, for
 * total_ms, samples delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md) in windows of window_ms while
 * the caller launches its kernels; _read waits for it and returns the clock of every window (MHz) in order.
 * _start returns once the probe is resident; _stop ends it within microseconds (a host flag), so that a device-wide
 * synchronisation behind the measured launches does not wait out the rest of total_ms. */
int muse_test_clock_probe_start(muse_ctx *ctx, double window_ms, double total_ms);
int muse_test_clock_probe_stop(muse_ctx *ctx);
int muse_test_clock_probe_read(muse_ctx *ctx, double *mhz, int32_t cap, int32_t *windows);

/* Device unit test of the n = 4096 kernels' argmax step (foldk_device.h, wave_argmax_store; maxAbsIndex of
 * xcorr.go:39-50): ccA / ccB are 4096 host doubles each, laid out in registers as the kernels hold a pair's correlations;
 * out24 receives, per wave w = 0..3, {max |cc|, signed value at the winning index (cc[0] of the wave's first lane when
 * nothing is above 0), winning index (2147483647 when nothing is above 0)} of series A, then of series B. */
int muse_test_wave_argmax(muse_ctx *ctx, const double *ccA, const double *ccB, double *out24);

/* muse_batch_run_rows lets the fused kernel read groups of up to 256 KB straight out of the pinned staging buffer (no copy
 * command); always_copy = 1 sends every group through the host -> HBM copy instead (A/B of the two, and the parity suite
 * runs both). */
int muse_test_rows_always_copy(muse_ctx *ctx, int32_t always_copy);

/* The context's allocation cache (muse_ctx_trim): bytes and blocks it holds idle, device and pinned host.  Any out
 * pointer may be NULL. */
int muse_test_pool_stats(muse_ctx *ctx, int64_t *dev_idle_bytes, int64_t *dev_idle_blocks, int64_t *host_idle_bytes,
                         int64_t *host_idle_blocks);
/* Measurement hook: muse_xcorr_groups launches its kernel `repeat` times back to back (same results) -- a sustained burst
 * for the clock probe and for HIP-event timing without the host's work between calls (tools/clock_trace_two_sided.py). */
int muse_test_xcorr_repeat(muse_ctx *ctx, int32_t repeat);

/* Measurement hook: the work buffer of one batch of the long-series pass (FFT lengths above 65 536; xcorr_huge.hip) in MB;
 * 0 = the built-in 128 MB (half the Infinity Cache).  tools/huge_bench.py, profiles/r06_long_series.txt. */
int muse_test_huge_batch_mb(muse_ctx *ctx, int32_t megabytes);

#ifdef __cplusplus
}
#endif
#endif
