/*
 * muse_oracle.h -- CPU restatement of go-muse's z-normalized cross-correlation
 * hot path (reference: /root/reference/xcorr.go, muse_batch.go, muse.go,
 * results.go, scores.go).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker
 * or as the timed CPU baseline.  The product path (libmuse_hip.so) never
 * links, loads or calls anything declared here.
 *
 * Parity pin: the restatement is checked against every known-answer table the
 * reference's own tests hold for this path (tests/golden/, transcribed from
 * xcorr_test.go, muse_batch_test.go, muse_test.go) -- 1e-8 on full cc vectors
 * at n=5, 1e-3 on scores and exact lags at n=8/16.  The Go reference itself
 * cannot be built here (no Go toolchain; gonum v0.7.0 not vendored), so at
 * N=4096 parity is "oracle vs kernel" plus the oracle's own exactness check
 * against a long-double direct correlation (oracle_xcorr_direct_ld).
 */
#ifndef MUSE_ORACLE_H
#define MUSE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* xcorr.go:19-24  nextPowOf2 */
int64_t oracle_next_pow2(double val);

/* xcorr.go:84-95  zNormalize (in place). returns 0 ok, 1 = errStdDevZero
 * (x is left mean-subtracted in that case, as in the reference). */
int oracle_znormalize(double *x, int64_t len);

/* xcorr.go:70-80  zeroPad: writes max(n,len) values to out, returns the
 * output length (len when n < len: input returned unchanged). */
int64_t oracle_zero_pad(const double *x, int64_t len, int64_t n, double *out);

/* xcorr.go:39-50  maxAbsIndex */
int64_t oracle_max_abs_index(const double *x, int64_t len);

/* gonum dsp/fourier FFT.Coefficients / FFT.Sequence semantics (v0.7.0):
 * unnormalized forward real DFT -> n/2+1 complex (interleaved re,im), and the
 * unnormalized inverse.  Any n >= 1 (power of two: FFT; otherwise O(n^2)). */
void oracle_rfft(const double *seq, int64_t n, double *coef /* 2*(n/2+1) */);
void oracle_irfft(const double *coef, int64_t n, double *seq /* n */);

/* xcorr.go:102-153  xCorr.  x and y are NOT modified (the reference mutates
 * them in place when normalize is set; parity is defined on fresh copies).
 * cc must hold max(n, lenx, leny) doubles.  Returns 0, or 1 when the
 * reference returns (nil,0,0) (sigma == 0); *n_out = FFT length used. */
int oracle_xcorr(const double *x, int64_t lenx, const double *y, int64_t leny,
                 int64_t n, int normalize, double *cc, int64_t *n_out,
                 int64_t *lag, double *mv);

/* muse_batch.go:33-47 / muse.go:27-39  reference-spectrum precompute:
 * X = rfft(zeroPad(zNormalize(ref)/(N-1), n)).  X holds 2*(n/2+1) doubles.
 * Returns 0, or 1 on sigma(ref)==0 ("Invalid input query"). */
int oracle_ref_spectrum(const double *ref, int64_t N, int64_t n, double *X);

/* xcorr.go:160-197  xCorrWithX.  y (length N) is NOT modified.  cc may be
 * NULL.  Returns 0, or 1 when the reference returns (nil,0,0).
 * gap (optional): (max|cc| - second largest |cc| at another index)/max|cc|,
 * used by the parity harness to flag rounding-decided ties. */
int oracle_xcorr_with_x(const double *X, const double *y, int64_t N, int64_t n,
                        double *cc, int64_t *lag, double *mv, double *gap);

/* Independent exactness check: direct O(N*n) circular correlation of the
 * z-normalized, leading-zero-padded series in long double.  cc holds n. */
int oracle_xcorr_direct_ld(const double *ref, const double *y, int64_t N,
                           int64_t n, double *cc);

/* The inner loop of Batch.scoreSingle (muse_batch.go:68-73) over M rows:
 * per-series (lag, signed mv) exactly as xCorrWithX returns them.  rows is
 * row-major with row_stride doubles between rows.  nthreads worker threads,
 * one FFT plan + scratch each (muse_batch.go:62-64).  gap may be NULL.
 * Returns 0, 1 on sigma(ref)==0, 2 on bad arguments. */
int oracle_batch_scores(const double *ref, const double *rows, int64_t M,
                        int64_t N, int64_t row_stride, int nthreads,
                        int32_t *lag, double *mv, double *gap);

/* Batch.Run + Results.Update + Results.Fetch (muse_batch.go:99-130,
 * results.go:46-87) or, with abs_scores=0, the Muse.Run post-processing
 * (muse.go:72-90), given per-series (lag, mv).  group_id[i] in [0,G) or NULL
 * (each series its own group).  Groups are updated in group-id order; inside
 * a group series are visited in index order.  The Results heap is Go's
 * container/heap on |score| (scores.go:25-27).  Outputs are in Fetch order
 * (descending |score|).  out_* hold top_n entries.  Returns count. */
int64_t oracle_results(const int32_t *lag, const double *mv, int64_t M,
                       const int32_t *group_id, int64_t G, int abs_scores,
                       int64_t max_lag, int64_t top_n, double threshold,
                       int sign_filter, int64_t *out_series, int32_t *out_lag,
                       double *out_score, double *out_mean_abs);

#ifdef __cplusplus
}
#endif
#endif
