/*
 * muse_oracle.c -- CPU restatement (plain C, double precision) of go-muse's
 * XCorr / XCorrWithX / Batch.Run / Results path.  See muse_oracle.h for the
 * role of this file: TEST INFRASTRUCTURE + timed CPU baseline only.
 *
 * Every function cites the reference lines it follows (paths are relative to
 * /root/reference).  Third-party arithmetic that is not vendored there --
 * gonum.org/v1/gonum v0.7.0 (go.mod:8): dsp/fourier.FFT.{Coefficients,
 * Sequence}, floats.{Sum,AddConst,Scale}, stat.StdDev -- is restated from its
 * published contract: unnormalized forward real DFT with e^{-2 pi i jk/n}
 * returning n/2+1 coefficients; unnormalized inverse (imaginary parts of the
 * DC and Nyquist terms ignored); sample standard deviation by the corrected
 * two-pass algorithm (Chan/Golub/LeVeque eq. 1.7) around the recomputed mean.
 * FFTPACK's rounding is not reproduced (any fp64 FFT agrees to ~1e-15).
 */
#include "muse_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_PIL 3.14159265358979323846264338327950288L

/* ------------------------------------------------------------------ a-1 */
/* xcorr.go:19-24 */
int64_t oracle_next_pow2(double val)
{
    if (val <= 0)
        return 0;
    return (int64_t)pow(2.0, ceil(log(val) / log(2.0)));
}

/* ------------------------------------------------------------------ a-2 */
/* xcorr.go:84-95.  floats.Sum / AddConst (xcorr.go:86), stat.StdDev
 * (xcorr.go:88), floats.Scale (xcorr.go:93). */
int oracle_znormalize(double *x, int64_t len)
{
    double n = (double)len;
    double sum = 0.0;
    for (int64_t i = 0; i < len; i++)
        sum += x[i];
    double c = -sum / n;
    for (int64_t i = 0; i < len; i++)
        x[i] += c;

    /* stat.StdDev(x, nil) = sqrt(Variance): mean recomputed, corrected
     * two-pass, divisor len-1. */
    double msum = 0.0;
    for (int64_t i = 0; i < len; i++)
        msum += x[i];
    double mean = msum / n;
    double ss = 0.0, comp = 0.0;
    for (int64_t i = 0; i < len; i++) {
        double d = x[i] - mean;
        ss += d * d;
        comp += d;
    }
    double variance = (ss - comp * comp / n) / (double)(len - 1);
    double std = sqrt(variance);
    if (std == 0)
        return 1;
    double s = 1 / std;
    for (int64_t i = 0; i < len; i++)
        x[i] *= s;
    return 0;
}

/* ------------------------------------------------------------------ a-3 */
/* xcorr.go:70-80 */
int64_t oracle_zero_pad(const double *x, int64_t len, int64_t n, double *out)
{
    if (n < len) {
        memmove(out, x, (size_t)len * sizeof(double));
        return len;
    }
    for (int64_t i = 0; i < n - len; i++)
        out[i] = 0.0;
    for (int64_t i = 0; i < len; i++)
        out[n - len + i] = x[i];
    return n;
}

/* ------------------------------------------------------------------ a-9 */
/* xcorr.go:39-50: first index with strictly greatest |v|, start (0, 0.0). */
int64_t oracle_max_abs_index(const double *x, int64_t len)
{
    int64_t max_index = 0;
    double max_val = 0.0;
    for (int64_t i = 0; i < len; i++) {
        if (fabs(x[i]) > fabs(max_val)) {
            max_val = x[i];
            max_index = i;
        }
    }
    return max_index;
}

/* ------------------------------------------------------- a-5 / a-7: DFT */
typedef struct {
    int64_t n;      /* real length */
    int pow2;       /* n is a power of two and >= 4 */
    int64_t m;      /* n/2: complex FFT size */
    int logm;
    int64_t *rev;   /* bit reversal of size m */
    double *tw;     /* m/2 twiddles exp(-2 pi i k/m), interleaved */
    double *tn;     /* n/4+1.. : exp(-2 pi i k/n), k<=m/2, interleaved */
    double *zr;     /* scratch m complex */
} plan_t;

static int is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

static plan_t *plan_new(int64_t n)
{
    plan_t *p = (plan_t *)calloc(1, sizeof(plan_t));
    p->n = n;
    p->pow2 = is_pow2(n) && n >= 4;
    if (!p->pow2)
        return p;
    int64_t m = n / 2;
    p->m = m;
    int lg = 0;
    while (((int64_t)1 << lg) < m)
        lg++;
    p->logm = lg;
    p->rev = (int64_t *)malloc((size_t)m * sizeof(int64_t));
    for (int64_t i = 0; i < m; i++) {
        int64_t r = 0;
        for (int b = 0; b < lg; b++)
            if (i & ((int64_t)1 << b))
                r |= (int64_t)1 << (lg - 1 - b);
        p->rev[i] = r;
    }
    p->tw = (double *)malloc((size_t)(m / 2 + 1) * 2 * sizeof(double));
    for (int64_t k = 0; k < m / 2 + 1; k++) {
        long double a = -2.0L * ORACLE_PIL * (long double)k / (long double)m;
        p->tw[2 * k] = (double)cosl(a);
        p->tw[2 * k + 1] = (double)sinl(a);
    }
    p->tn = (double *)malloc((size_t)(m + 1) * 2 * sizeof(double));
    for (int64_t k = 0; k <= m; k++) {
        long double a = -2.0L * ORACLE_PIL * (long double)k / (long double)n;
        p->tn[2 * k] = (double)cosl(a);
        p->tn[2 * k + 1] = (double)sinl(a);
    }
    p->zr = (double *)malloc((size_t)m * 2 * sizeof(double));
    return p;
}

static void plan_free(plan_t *p)
{
    if (!p)
        return;
    free(p->rev);
    free(p->tw);
    free(p->tn);
    free(p->zr);
    free(p);
}

/* in-place forward complex FFT of size m on bit-reversed-loaded data z */
static void cfft_inplace(const plan_t *p, double *z)
{
    int64_t m = p->m;
    for (int64_t h = 1; h < m; h <<= 1) {
        int64_t step = m / (2 * h);
        for (int64_t base = 0; base < m; base += 2 * h) {
            for (int64_t j = 0; j < h; j++) {
                double wr = p->tw[2 * (j * step)], wi = p->tw[2 * (j * step) + 1];
                double *a = z + 2 * (base + j), *b = z + 2 * (base + j + h);
                double tr = b[0] * wr - b[1] * wi;
                double ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr;
                b[1] = a[1] - ti;
                a[0] += tr;
                a[1] += ti;
            }
        }
    }
}

static void naive_rfft(const double *seq, int64_t n, double *coef)
{
    for (int64_t k = 0; k <= n / 2; k++) {
        long double re = 0, im = 0;
        for (int64_t j = 0; j < n; j++) {
            int64_t r = (j * k) % n;
            long double a = -2.0L * ORACLE_PIL * (long double)r / (long double)n;
            re += (long double)seq[j] * cosl(a);
            im += (long double)seq[j] * sinl(a);
        }
        coef[2 * k] = (double)re;
        coef[2 * k + 1] = (double)im;
    }
}

static void naive_irfft(const double *coef, int64_t n, double *seq)
{
    /* unnormalized inverse of a Hermitian spectrum given by k = 0..n/2 */
    for (int64_t j = 0; j < n; j++) {
        long double acc = (long double)coef[0];
        for (int64_t k = 1; k <= n / 2; k++) {
            int64_t r = (j * k) % n;
            long double a = 2.0L * ORACLE_PIL * (long double)r / (long double)n;
            long double cr = coef[2 * k], ci = coef[2 * k + 1];
            if (n % 2 == 0 && k == n / 2) {
                acc += cr * cosl(a); /* Nyquist: imaginary part ignored */
            } else {
                acc += 2.0L * (cr * cosl(a) - ci * sinl(a));
            }
        }
        seq[j] = (double)acc;
    }
}

static void plan_rfft(const plan_t *p, const double *seq, double *coef)
{
    if (!p->pow2) {
        naive_rfft(seq, p->n, coef);
        return;
    }
    int64_t m = p->m, n = p->n;
    double *z = p->zr;
    for (int64_t i = 0; i < m; i++) {
        int64_t r = p->rev[i];
        z[2 * r] = seq[2 * i];
        z[2 * r + 1] = seq[2 * i + 1];
    }
    cfft_inplace(p, z);
    /* untangle: X[k] = E[k] + W_n^k O[k] */
    for (int64_t k = 0; k <= m; k++) {
        int64_t k1 = k % m, k2 = (m - k) % m;
        double ar = z[2 * k1], ai = z[2 * k1 + 1];
        double br = z[2 * k2], bi = -z[2 * k2 + 1]; /* conj(Z[m-k]) */
        double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
        /* O = (a - b)/(2i) = (-i/2)(a-b) */
        double dr = ar - br, di = ai - bi;
        double or_ = 0.5 * di, oi = -0.5 * dr;
        double wr = p->tn[2 * k], wi = p->tn[2 * k + 1];
        coef[2 * k] = er + (or_ * wr - oi * wi);
        coef[2 * k + 1] = ei + (or_ * wi + oi * wr);
    }
    (void)n;
}

static void plan_irfft(const plan_t *p, const double *coef, double *seq)
{
    if (!p->pow2) {
        naive_irfft(coef, p->n, seq);
        return;
    }
    int64_t m = p->m;
    double *z = p->zr;
    /* Z'[k] = (X[k] + conj(X[m-k])) + i W_n^{-k} (X[k] - conj(X[m-k]));
     * the inverse complex FFT is run as conj(FFT(conj(Z'))). */
    for (int64_t k = 0; k < m; k++) {
        double ar = coef[2 * k], ai = coef[2 * k + 1];
        double br = coef[2 * (m - k)], bi = -coef[2 * (m - k) + 1];
        if (k == 0) {
            ai = 0.0; /* imag of DC ignored   */
            bi = 0.0; /* imag of Nyquist ignored */
        }
        double sr = ar + br, si = ai + bi;
        double dr = ar - br, di = ai - bi;
        /* i * conj(W_n^k) * d,  conj(W) = (wr, -wi) */
        double wr = p->tn[2 * k], wi = -p->tn[2 * k + 1];
        double tr = dr * wr - di * wi, ti = dr * wi + di * wr;
        double zr = sr - ti, zi = si + tr;
        int64_t r = p->rev[k];
        z[2 * r] = zr;
        z[2 * r + 1] = -zi; /* conj on the way in */
    }
    cfft_inplace(p, z);
    for (int64_t i = 0; i < m; i++) {
        seq[2 * i] = z[2 * i];
        seq[2 * i + 1] = -z[2 * i + 1]; /* conj on the way out */
    }
}

void oracle_rfft(const double *seq, int64_t n, double *coef)
{
    plan_t *p = plan_new(n);
    plan_rfft(p, seq, coef);
    plan_free(p);
}

void oracle_irfft(const double *coef, int64_t n, double *seq)
{
    plan_t *p = plan_new(n);
    plan_irfft(p, coef, seq);
    plan_free(p);
}

/* ---------------------------------------------------------- a-6 helpers */
/* xcorr.go:63-67 conj + xcorr.go:53-60 mult: dst = conj(dst) * src */
static void conj_mult(double *dst, const double *src, int64_t cnt)
{
    for (int64_t k = 0; k < cnt; k++) {
        double ar = dst[2 * k], ai = -dst[2 * k + 1];
        double br = src[2 * k], bi = src[2 * k + 1];
        dst[2 * k] = ar * br - ai * bi;
        dst[2 * k + 1] = ar * bi + ai * br;
    }
}

static double second_gap(const double *cc, int64_t n, int64_t mi)
{
    double mx = fabs(cc[mi]), second = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (i == mi)
            continue;
        double a = fabs(cc[i]);
        if (a > second)
            second = a;
    }
    if (!(mx > 0))
        return 0.0;
    return (mx - second) / mx;
}

/* ----------------------------------------------------------------- a-11 */
/* xcorr.go:102-153 */
int oracle_xcorr(const double *x_in, int64_t lenx, const double *y_in,
                 int64_t leny, int64_t n, int normalize, double *cc,
                 int64_t *n_out, int64_t *lag, double *mv)
{
    int64_t minn = lenx > leny ? lenx : leny; /* xcorr.go:104-106 */
    if (n < minn)
        n = minn;
    if (n_out)
        *n_out = n;
    *lag = 0;
    *mv = 0;
    double *x = (double *)malloc((size_t)(lenx > 0 ? lenx : 1) * sizeof(double));
    double *y = (double *)malloc((size_t)(leny > 0 ? leny : 1) * sizeof(double));
    memcpy(x, x_in, (size_t)lenx * sizeof(double));
    memcpy(y, y_in, (size_t)leny * sizeof(double));
    if (normalize) { /* xcorr.go:108-128 */
        if (oracle_znormalize(x, lenx) || oracle_znormalize(y, leny)) {
            free(x);
            free(y);
            return 1;
        }
    }
    double *xp = (double *)malloc((size_t)n * sizeof(double));
    double *yp = (double *)malloc((size_t)n * sizeof(double));
    oracle_zero_pad(x, lenx, n, xp); /* xcorr.go:129-130 */
    oracle_zero_pad(y, leny, n, yp);
    plan_t *p = plan_new(n); /* xcorr.go:132 */
    int64_t nc = n / 2 + 1;
    double *X = (double *)malloc((size_t)nc * 2 * sizeof(double));
    double *Y = (double *)malloc((size_t)nc * 2 * sizeof(double));
    plan_rfft(p, xp, X); /* xcorr.go:134-135 */
    plan_rfft(p, yp, Y);
    /* conj(Y); mult(X, Y): X = X * conj(Y)   xcorr.go:136-137 */
    for (int64_t k = 0; k < nc; k++) {
        double ar = X[2 * k], ai = X[2 * k + 1];
        double br = Y[2 * k], bi = -Y[2 * k + 1];
        X[2 * k] = ar * br - ai * bi;
        X[2 * k + 1] = ar * bi + ai * br;
    }
    plan_irfft(p, X, cc); /* xcorr.go:138 */
    double s = normalize ? 1.0 / (double)(n * (n - 1)) : 1.0 / (double)n; /* :139-143 */
    for (int64_t i = 0; i < n; i++)
        cc[i] *= s;
    int64_t mi = oracle_max_abs_index(cc, n); /* :145-146 */
    *mv = cc[mi];
    if (mi > n / 2) /* :148-150 */
        mi -= n;
    *lag = mi;
    plan_free(p);
    free(x); free(y); free(xp); free(yp); free(X); free(Y);
    return 0;
}

/* ----------------------------------------------------------------- a-12 */
/* muse_batch.go:35-47; muse.go:27-39 */
static int ref_spectrum_plan(const plan_t *p, const double *ref, int64_t N,
                             int64_t n, double *X)
{
    double *x = (double *)malloc((size_t)N * sizeof(double));
    memcpy(x, ref, (size_t)N * sizeof(double));
    if (oracle_znormalize(x, N)) { /* muse_batch.go:38-41 */
        free(x);
        return 1;
    }
    double s = 1 / (double)(N - 1); /* muse_batch.go:42 */
    for (int64_t i = 0; i < N; i++)
        x[i] *= s;
    double *xp = (double *)malloc((size_t)(n > N ? n : N) * sizeof(double));
    oracle_zero_pad(x, N, n, xp); /* muse_batch.go:43 */
    plan_rfft(p, xp, X);          /* muse_batch.go:47 */
    free(x);
    free(xp);
    return 0;
}

int oracle_ref_spectrum(const double *ref, int64_t N, int64_t n, double *X)
{
    plan_t *p = plan_new(n);
    int rc = ref_spectrum_plan(p, ref, N, n, X);
    plan_free(p);
    return rc;
}

/* ----------------------------------------------------------------- a-10 */
/* xcorr.go:160-197, with caller-provided scratch (coef: 2*(n/2+1), seq: n,
 * yz: N) the way scoreSingle provides it (muse_batch.go:62-64). */
static int xcorr_with_x_plan(const plan_t *p, const double *X, const double *y,
                             int64_t N, int64_t n, double *yz, double *coef,
                             double *seq, int64_t *lag, double *mv, double *gap)
{
    *lag = 0;
    *mv = 0;
    if (gap)
        *gap = 0;
    memcpy(yz, y, (size_t)N * sizeof(double));
    if (oracle_znormalize(yz, N)) /* xcorr.go:164-172 */
        return 1;
    for (int64_t i = 0; i < n - N; i++) /* xcorr.go:176-178 */
        seq[i] = 0;
    for (int64_t i = 0; i < N; i++) /* xcorr.go:179-181 */
        seq[n - N + i] = yz[i];
    plan_rfft(p, seq, coef);        /* xcorr.go:183 */
    conj_mult(coef, X, n / 2 + 1);  /* xcorr.go:184-185 */
    plan_irfft(p, coef, seq);       /* xcorr.go:186 */
    double s = 1.0 / (double)n;     /* xcorr.go:187 */
    for (int64_t i = 0; i < n; i++)
        seq[i] *= s;
    int64_t mi = oracle_max_abs_index(seq, n); /* xcorr.go:189 */
    *mv = seq[mi];                             /* xcorr.go:190 */
    if (gap)
        *gap = second_gap(seq, n, mi);
    if (mi > n / 2) /* xcorr.go:192-194 */
        mi -= n;
    *lag = mi;
    return 0;
}

int oracle_xcorr_with_x(const double *X, const double *y, int64_t N, int64_t n,
                        double *cc, int64_t *lag, double *mv, double *gap)
{
    plan_t *p = plan_new(n);
    double *yz = (double *)malloc((size_t)N * sizeof(double));
    double *coef = (double *)malloc((size_t)(n / 2 + 1) * 2 * sizeof(double));
    double *seq = (double *)malloc((size_t)n * sizeof(double));
    int rc = xcorr_with_x_plan(p, X, y, N, n, yz, coef, seq, lag, mv, gap);
    if (cc && rc == 0)
        memcpy(cc, seq, (size_t)n * sizeof(double));
    free(yz); free(coef); free(seq);
    plan_free(p);
    return rc;
}

/* Independent check: cc[k] = sum_j yz_pad[j] * xs_pad[(j+k) mod n] in long
 * double, where xs = zNormalize(ref)/(N-1). */
int oracle_xcorr_direct_ld(const double *ref, const double *y, int64_t N,
                           int64_t n, double *cc)
{
    double *x = (double *)malloc((size_t)N * sizeof(double));
    double *yz = (double *)malloc((size_t)N * sizeof(double));
    memcpy(x, ref, (size_t)N * sizeof(double));
    memcpy(yz, y, (size_t)N * sizeof(double));
    if (oracle_znormalize(x, N) || oracle_znormalize(yz, N)) {
        free(x); free(yz);
        return 1;
    }
    double s = 1 / (double)(N - 1);
    for (int64_t i = 0; i < N; i++)
        x[i] *= s;
    int64_t off = n - N;
    for (int64_t k = 0; k < n; k++) {
        long double acc = 0;
        for (int64_t j = 0; j < N; j++) {
            int64_t q = (off + j + k) % n; /* padded index of the x sample */
            if (q >= off)
                acc += (long double)yz[j] * (long double)x[q - off];
        }
        cc[k] = (double)acc;
    }
    free(x); free(yz);
    return 0;
}

/* ------------------------------------------------ a-13/14: batch scoring */
typedef struct {
    const double *X, *rows;
    int64_t N, n, row_stride, lo, hi;
    int32_t *lag;
    double *mv, *gap;
} work_t;

static void *score_worker(void *arg)
{
    work_t *w = (work_t *)arg;
    /* per-goroutine FFT + scratch: muse_batch.go:62-64 */
    plan_t *p = plan_new(w->n);
    double *yz = (double *)malloc((size_t)w->N * sizeof(double));
    double *coef = (double *)malloc((size_t)(w->n / 2 + 1) * 2 * sizeof(double));
    double *seq = (double *)malloc((size_t)w->n * sizeof(double));
    for (int64_t i = w->lo; i < w->hi; i++) { /* muse_batch.go:68-73 */
        int64_t lg;
        double mv, gp;
        xcorr_with_x_plan(p, w->X, w->rows + i * w->row_stride, w->N, w->n, yz,
                          coef, seq, &lg, &mv, w->gap ? &gp : NULL);
        w->lag[i] = (int32_t)lg;
        w->mv[i] = mv;
        if (w->gap)
            w->gap[i] = gp;
    }
    free(yz); free(coef); free(seq);
    plan_free(p);
    return NULL;
}

int oracle_batch_scores(const double *ref, const double *rows, int64_t M,
                        int64_t N, int64_t row_stride, int nthreads,
                        int32_t *lag, double *mv, double *gap)
{
    if (N < 1 || M < 0 || row_stride < N)
        return 2;
    int64_t n = oracle_next_pow2((double)N); /* muse_batch.go:35 */
    double *X = (double *)malloc((size_t)(n / 2 + 1) * 2 * sizeof(double));
    if (oracle_ref_spectrum(ref, N, n, X)) {
        free(X);
        return 1;
    }
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > 256)
        nthreads = 256;
    pthread_t th[256];
    work_t wk[256];
    int64_t per = (M + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        int64_t lo = t * per, hi = lo + per;
        if (lo > M) lo = M;
        if (hi > M) hi = M;
        wk[t] = (work_t){X, rows, N, n, row_stride, lo, hi, lag, mv, gap};
        if (nthreads == 1)
            score_worker(&wk[t]);
        else
            pthread_create(&th[t], NULL, score_worker, &wk[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++)
            pthread_join(th[t], NULL);
    free(X);
    return 0;
}

/* -------------------------------------- a-13/15/16: group max + Results */
typedef struct {
    int64_t series;
    int32_t lag;
    double score;
} score_t;

/* scores.go:25-27 Less on |PercentScore|; container/heap up/down */
static int sc_less(const score_t *h, int64_t i, int64_t j)
{
    return fabs(h[i].score) < fabs(h[j].score);
}
static void sc_swap(score_t *h, int64_t i, int64_t j)
{
    score_t t = h[i]; h[i] = h[j]; h[j] = t;
}
static void heap_up(score_t *h, int64_t j)
{
    for (;;) {
        int64_t i = (j - 1) / 2; /* Go: truncating division, (-1)/2 == 0 */
        if (i == j || !sc_less(h, j, i))
            break;
        sc_swap(h, i, j);
        j = i;
    }
}
static void heap_down(score_t *h, int64_t i0, int64_t n)
{
    int64_t i = i0;
    for (;;) {
        int64_t j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0)
            break;
        int64_t j = j1, j2 = j1 + 1;
        if (j2 < n && sc_less(h, j2, j1))
            j = j2;
        if (!sc_less(h, j, i))
            break;
        sc_swap(h, i, j);
        i = j;
    }
}

/* results.go:46-52 */
static int passed(const score_t *s, int64_t max_lag, double threshold, int sign_filter)
{
    return fabs((double)s->lag) <= (double)max_lag &&
           fabs(s->score) >= threshold &&
           (sign_filter == 0 || (s->score > 0 && sign_filter == 1) ||
            (s->score < 0 && sign_filter == -1));
}

int64_t oracle_results(const int32_t *lag, const double *mv, int64_t M,
                       const int32_t *group_id, int64_t G, int abs_scores,
                       int64_t max_lag, int64_t top_n, double threshold,
                       int sign_filter, int64_t *out_series, int32_t *out_lag,
                       double *out_score, double *out_mean_abs)
{
    if (!group_id)
        G = M;
    if (out_mean_abs)
        *out_mean_abs = NAN; /* results.go:86: 0/0 when empty */
    if (G <= 0 || top_n <= 0)
        return 0;
    /* group maxima: muse_batch.go:68-90 (abs) / muse.go:64-89 (signed) */
    score_t *best = (score_t *)malloc((size_t)G * sizeof(score_t));
    for (int64_t g = 0; g < G; g++)
        best[g].series = -1; /* maxScore.Labels == nil */
    for (int64_t i = 0; i < M; i++) {
        int64_t g = group_id ? group_id[i] : i;
        if (g < 0 || g >= G)
            continue;
        double v = mv[i];
        if (abs_scores) { /* muse_batch.go:74-77 */
            v = fabs(v);
            if (v > 1.0)
                v = 1.0;
        } else { /* muse.go:72-76 */
            if (v > 1.0)
                v = 1.0;
            else if (v < -1.0)
                v = -1.0;
        }
        score_t s = {i, lag[i], v};
        int take;
        if (best[g].series < 0)
            take = 1;
        else if (abs_scores)
            take = s.score > best[g].score; /* muse_batch.go:87 */
        else
            take = fabs(s.score) > fabs(best[g].score); /* muse.go:86 */
        if (take)
            best[g] = s;
    }
    /* Results.Update in group order: muse_batch.go:124-128, results.go:55-72 */
    score_t *h = (score_t *)malloc((size_t)top_n * sizeof(score_t));
    int64_t hn = 0;
    for (int64_t g = 0; g < G; g++) {
        if (best[g].series < 0)
            continue; /* results.go:56-59 */
        if (!passed(&best[g], max_lag, threshold, sign_filter))
            continue;
        if (hn == top_n) {
            if (fabs(best[g].score) > fabs(h[0].score)) { /* results.go:63 */
                /* heap.Pop */
                sc_swap(h, 0, hn - 1);
                heap_down(h, 0, hn - 1);
                hn--;
                /* heap.Push */
                h[hn++] = best[g];
                heap_up(h, hn - 1);
            }
        } else {
            h[hn++] = best[g];
            heap_up(h, hn - 1);
        }
    }
    /* Results.Fetch: results.go:75-87 */
    int64_t num = hn;
    double sum = 0;
    for (int64_t i = num - 1; i >= 0; i--) {
        sc_swap(h, 0, hn - 1);
        heap_down(h, 0, hn - 1);
        hn--;
        score_t s = h[hn];
        sum += fabs(s.score);
        out_series[i] = s.series;
        out_lag[i] = s.lag;
        out_score[i] = s.score;
    }
    if (out_mean_abs)
        *out_mean_abs = sum / (double)num;
    free(best);
    free(h);
    return num;
}
