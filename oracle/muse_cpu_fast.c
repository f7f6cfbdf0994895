/*
 * muse_cpu_fast.c -- the TIMED CPU baseline of bench.py (cpu_baseline.kind = "port").
 *
 * TEST / MEASUREMENT INFRASTRUCTURE, like everything under oracle/: nothing in go-muse_amd/ may import, link or
 * call it.  It is NOT the checker either: parity is decided by muse_oracle.c (radix-2, -O2, no contraction), and
 * this file is itself checked against that oracle (tests/test_oracle_golden.py::test_fast_cpu_port_matches_oracle).
 *
 * Why a second CPU implementation: the reference's FFT is gonum v0.7.0 dsp/fourier (FFTPACK rfftf / rfftb, mostly
 * radix-4 passes at n = 4096); timing the checker's scalar radix-2 transform beside the GPU would make the baseline
 * a strawman.  Here: the same algorithm as the reference's hot path
 *     zNormalize (xcorr.go:84-95) -> leading zero pad (xcorr.go:176-181) -> real FFT (xcorr.go:183)
 *     -> conj * X (xcorr.go:184-185) -> inverse real FFT (xcorr.go:186) -> 1/n (xcorr.go:187)
 *     -> maxAbsIndex + lag unwrap (xcorr.go:39-50, 189-194)
 * with the real transforms done as a half-length complex radix-4 Stockham autosort FFT (no bit reversal, unit-stride
 * inner loops the compiler vectorises) plus the usual real/complex untangling; built -O3 -march=native; one plan and
 * one set of scratch buffers per thread (the Concurrency = #cores analogue of muse_batch.go:111).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PIL 3.14159265358979323846264338327950288L

typedef struct {
    int64_t n, m;      /* real length n (power of two >= 4), complex length m = n / 2 */
    double *tw;        /* m entries: exp(-2 pi i k / m) */
    double *tn;        /* m + 1 entries: exp(-2 pi i k / n) (untangle) */
    double *a, *b;     /* ping-pong buffers, m complex each */
} fplan;

static fplan *fplan_new(int64_t n)
{
    fplan *p = (fplan *)calloc(1, sizeof(fplan));
    p->n = n;
    p->m = n / 2;
    p->tw = (double *)malloc((size_t)p->m * 2 * sizeof(double));
    p->tn = (double *)malloc((size_t)(p->m + 1) * 2 * sizeof(double));
    p->a = (double *)malloc((size_t)p->m * 2 * sizeof(double));
    p->b = (double *)malloc((size_t)p->m * 2 * sizeof(double));
    for (int64_t k = 0; k < p->m; k++) {
        long double a = -2.0L * PIL * (long double)k / (long double)p->m;
        p->tw[2 * k] = (double)cosl(a);
        p->tw[2 * k + 1] = (double)sinl(a);
    }
    for (int64_t k = 0; k <= p->m; k++) {
        long double a = -2.0L * PIL * (long double)k / (long double)n;
        p->tn[2 * k] = (double)cosl(a);
        p->tn[2 * k + 1] = (double)sinl(a);
    }
    return p;
}

static void fplan_free(fplan *p)
{
    if (!p)
        return;
    free(p->tw);
    free(p->tn);
    free(p->a);
    free(p->b);
    free(p);
}

/* Forward complex FFT of length m (power of two), Stockham autosort, radix 4 with one radix-2 pass when log2 m is
 * odd.  Input in x, scratch y; returns the buffer that holds the result.  Stage with sub-length len and stride s:
 * out[q + s (4 p + r)] = W_len^(r p) * sum_k in[q + s (p + k len/4)] (-i)^(r k). */
static double *cfft_stockham(const fplan *pl, double *x, double *y)
{
    const int64_t m = pl->m;
    int64_t len = m, s = 1;
    while (len >= 4) {
        const int64_t n1 = len / 4, tstep = m / len;
        for (int64_t p = 0; p < n1; p++) {
            const double w1r = pl->tw[2 * (p * tstep)], w1i = pl->tw[2 * (p * tstep) + 1];
            const double w2r = pl->tw[2 * (2 * p * tstep)], w2i = pl->tw[2 * (2 * p * tstep) + 1];
            const double w3r = pl->tw[2 * (3 * p * tstep)], w3i = pl->tw[2 * (3 * p * tstep) + 1];
            const double *xa = x + 2 * s * p, *xb = x + 2 * s * (p + n1), *xc = x + 2 * s * (p + 2 * n1),
                         *xd = x + 2 * s * (p + 3 * n1);
            double *y0 = y + 2 * s * (4 * p), *y1 = y + 2 * s * (4 * p + 1), *y2 = y + 2 * s * (4 * p + 2),
                   *y3 = y + 2 * s * (4 * p + 3);
            for (int64_t q = 0; q < s; q++) {
                const double ar = xa[2 * q], ai = xa[2 * q + 1], br = xb[2 * q], bi = xb[2 * q + 1];
                const double cr = xc[2 * q], ci = xc[2 * q + 1], dr = xd[2 * q], di = xd[2 * q + 1];
                const double apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
                const double bpdr = br + dr, bpdi = bi + di;
                /* -i (b - d) */
                const double jr = bi - di, ji = -(br - dr);
                y0[2 * q] = apcr + bpdr;
                y0[2 * q + 1] = apci + bpdi;
                const double t1r = amcr + jr, t1i = amci + ji;
                y1[2 * q] = t1r * w1r - t1i * w1i;
                y1[2 * q + 1] = t1r * w1i + t1i * w1r;
                const double t2r = apcr - bpdr, t2i = apci - bpdi;
                y2[2 * q] = t2r * w2r - t2i * w2i;
                y2[2 * q + 1] = t2r * w2i + t2i * w2r;
                const double t3r = amcr - jr, t3i = amci - ji;
                y3[2 * q] = t3r * w3r - t3i * w3i;
                y3[2 * q + 1] = t3r * w3i + t3i * w3r;
            }
        }
        double *t = x;
        x = y;
        y = t;
        len /= 4;
        s *= 4;
    }
    if (len == 2) { /* last radix-2 pass: no twiddles */
        const double *xa = x, *xb = x + 2 * s;
        double *y0 = y, *y1 = y + 2 * s;
        for (int64_t q = 0; q < s; q++) {
            const double ar = xa[2 * q], ai = xa[2 * q + 1], br = xb[2 * q], bi = xb[2 * q + 1];
            y0[2 * q] = ar + br;
            y0[2 * q + 1] = ai + bi;
            y1[2 * q] = ar - br;
            y1[2 * q + 1] = ai - bi;
        }
        return y;
    }
    return x;
}

/* unnormalised forward real DFT: n reals -> n/2 + 1 complex (gonum FFT.Coefficients, xcorr.go:183) */
static void frfft(fplan *p, const double *seq, double *coef)
{
    const int64_t m = p->m;
    memcpy(p->a, seq, (size_t)p->n * sizeof(double)); /* z[j] = seq[2j] + i seq[2j+1] */
    const double *z = cfft_stockham(p, p->a, p->b);
    for (int64_t k = 0; k <= m; k++) {
        const int64_t k1 = k % m, k2 = (m - k) % m;
        const double ar = z[2 * k1], ai = z[2 * k1 + 1], br = z[2 * k2], bi = -z[2 * k2 + 1];
        const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
        const double dr = ar - br, di = ai - bi;
        const double or_ = 0.5 * di, oi = -0.5 * dr;
        const double wr = p->tn[2 * k], wi = p->tn[2 * k + 1];
        coef[2 * k] = er + (or_ * wr - oi * wi);
        coef[2 * k + 1] = ei + (or_ * wi + oi * wr);
    }
}

/* unnormalised inverse real DFT (gonum FFT.Sequence, xcorr.go:186); imaginary parts of DC / Nyquist ignored */
static void firfft(fplan *p, const double *coef, double *seq)
{
    const int64_t m = p->m;
    double *z = p->a;
    for (int64_t k = 0; k < m; k++) {
        double ar = coef[2 * k], ai = coef[2 * k + 1];
        double br = coef[2 * (m - k)], bi = -coef[2 * (m - k) + 1];
        if (k == 0) {
            ai = 0.0;
            bi = 0.0;
        }
        const double sr = ar + br, si = ai + bi, dr = ar - br, di = ai - bi;
        const double wr = p->tn[2 * k], wi = -p->tn[2 * k + 1];
        const double tr = dr * wr - di * wi, ti = dr * wi + di * wr;
        z[2 * k] = sr - ti;
        z[2 * k + 1] = -(si + tr); /* conj on the way in: inverse = conj(FFT(conj(.))) */
    }
    const double *r = cfft_stockham(p, p->a, p->b);
    for (int64_t i = 0; i < m; i++) {
        seq[2 * i] = r[2 * i];
        seq[2 * i + 1] = -r[2 * i + 1];
    }
}

/* zNormalize (xcorr.go:84-95): mean = Sum/N; x -= mean; sigma = stat.StdDev (gonum: corrected two-pass sample variance,
 * mean recomputed); sigma == 0 -> error; x *= 1/sigma.  Returns 1 when sigma == 0. */
static int fznorm(double *x, int64_t N)
{
    double s = 0.0;
    for (int64_t i = 0; i < N; i++)
        s += x[i];
    const double mean = s / (double)N;
    for (int64_t i = 0; i < N; i++)
        x[i] -= mean;
    double s2 = 0.0;
    for (int64_t i = 0; i < N; i++)
        s2 += x[i];
    const double mu = s2 / (double)N;
    double ss = 0.0, comp = 0.0;
    for (int64_t i = 0; i < N; i++) {
        const double d = x[i] - mu;
        ss += d * d;
        comp += d;
    }
    const double var = (ss - comp * comp / (double)N) / (double)(N - 1);
    const double sd = sqrt(var);
    if (sd == 0.0)
        return 1;
    const double inv = 1.0 / sd;
    for (int64_t i = 0; i < N; i++)
        x[i] *= inv;
    return 0;
}

typedef struct {
    const double *X, *rows;
    int64_t N, n, stride, lo, hi;
    int32_t *lag;
    double *mv;
} fwork;

static void *fworker(void *arg)
{
    fwork *w = (fwork *)arg;
    const int64_t N = w->N, n = w->n;
    fplan *p = fplan_new(n);
    double *y = (double *)malloc((size_t)N * sizeof(double));
    double *seq = (double *)malloc((size_t)n * sizeof(double));
    double *coef = (double *)malloc((size_t)(n / 2 + 1) * 2 * sizeof(double));
    for (int64_t r = w->lo; r < w->hi; r++) {
        memcpy(y, w->rows + r * w->stride, (size_t)N * sizeof(double));
        if (fznorm(y, N)) { /* xcorr.go:164-172: (nil, 0, 0) */
            w->lag[r] = 0;
            w->mv[r] = 0.0;
            continue;
        }
        memset(seq, 0, (size_t)(n - N) * sizeof(double)); /* xcorr.go:176-181 */
        memcpy(seq + (n - N), y, (size_t)N * sizeof(double));
        frfft(p, seq, coef);
        for (int64_t k = 0; k <= n / 2; k++) { /* conj (xcorr.go:63-67) then mult by X (xcorr.go:53-60) */
            const double ar = coef[2 * k], ai = -coef[2 * k + 1], br = w->X[2 * k], bi = w->X[2 * k + 1];
            coef[2 * k] = ar * br - ai * bi;
            coef[2 * k + 1] = ar * bi + ai * br;
        }
        firfft(p, coef, seq);
        const double sc = 1.0 / (double)n; /* xcorr.go:187 */
        int64_t mi = 0;
        double mx = 0.0;
        for (int64_t i = 0; i < n; i++) { /* xcorr.go:39-50 */
            seq[i] *= sc;
            const double a = fabs(seq[i]);
            if (a > mx) {
                mx = a;
                mi = i;
            }
        }
        w->mv[r] = seq[mi];
        w->lag[r] = (int32_t)(mi > n / 2 ? mi - n : mi); /* xcorr.go:192-194 */
    }
    free(y);
    free(seq);
    free(coef);
    fplan_free(p);
    return NULL;
}

/* 0 ok, 1 sigma(ref) == 0, 2 bad arguments (N must pad to a power of two n >= 4) */
int fast_batch_scores(const double *ref, const double *rows, int64_t M, int64_t N, int64_t row_stride, int nthreads,
                      int32_t *lag, double *mv)
{
    if (N < 2 || M < 0 || row_stride < N)
        return 2;
    int64_t n = 4;
    while (n < N)
        n <<= 1;
    /* x = FFT(zeroPad(zNormalize(ref) / (N-1), n))  (muse_batch.go:37-47) */
    double *x = (double *)malloc((size_t)N * sizeof(double));
    memcpy(x, ref, (size_t)N * sizeof(double));
    if (fznorm(x, N)) {
        free(x);
        return 1;
    }
    double *seq = (double *)calloc((size_t)n, sizeof(double));
    for (int64_t i = 0; i < N; i++)
        seq[n - N + i] = x[i] / (double)(N - 1);
    double *X = (double *)malloc((size_t)(n / 2 + 1) * 2 * sizeof(double));
    fplan *p = fplan_new(n);
    frfft(p, seq, X);
    fplan_free(p);
    free(seq);
    free(x);
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > 256)
        nthreads = 256;
    pthread_t th[256];
    fwork wk[256];
    const int64_t per = (M + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        int64_t lo = t * per, hi = lo + per;
        if (lo > M)
            lo = M;
        if (hi > M)
            hi = M;
        wk[t] = (fwork){X, rows, N, n, row_stride, lo, hi, lag, mv};
        if (nthreads == 1)
            fworker(&wk[t]);
        else
            pthread_create(&th[t], NULL, fworker, &wk[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++)
            pthread_join(th[t], NULL);
    free(X);
    return 0;
}
