// capi_huge.hip -- host side of the kernels for series longer than 65 536 samples (xcorr_huge.hip; FFT lengths 2^17 ... 2^20):
// the context's tables and work buffers, the reference spectrum of a batch (NewBatch, muse_batch.go:35-47), the all-scores
// pass (Batch.scoreSingle / Muse.Run, muse_batch.go:68-73, muse.go:64-71) and the pairwise form the two-sided xCorr and the
// single-pair entry points use (xcorr.go:102-153).
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere.
#include "capi_internal.h"
#include "xcorr_huge.h"

using namespace muse;

namespace {

// tables of the FFT length and work buffers for `pairs` transforms in flight (and as many per-pair tables when `tables`):
// grown on demand, kept by the context; every launch that uses them is enqueued on the context's stream under huge_mu
int huge_ensure(muse_ctx *ctx, int logn, int64_t pairs, bool tables)
{
    HugeWork &w = ctx->huge;
    const int li = logn - HUGE_MIN_LOGN;
    const int64_t n = (int64_t)1 << logn;
    if (!w.thi[li]) {
        std::vector<double2> hi((size_t)(n / 1024)), lo(1024);
        for (int64_t j = 0; j < n / 1024; j++)
            fill_twiddle(hi, (size_t)j, 1024 * j, n);
        for (int j = 0; j < 1024; j++)
            fill_twiddle(lo, (size_t)j, j, n);
        double2 *dhi = nullptr, *dlo = nullptr;
        hipError_t e = hipMalloc(&dhi, hi.size() * sizeof(double2));
        if (e == hipSuccess)
            e = hipMalloc(&dlo, lo.size() * sizeof(double2));
        if (e == hipSuccess)
            e = hipMemcpy(dhi, hi.data(), hi.size() * sizeof(double2), hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(dlo, lo.data(), lo.size() * sizeof(double2), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(dhi);
            (void)hipFree(dlo);
            return fail(MUSE_ERR_NOMEM, "long-series tables: %s", hipGetErrorString(e));
        }
        w.thi[li] = dhi;
        w.tlo[li] = dlo;
    }
    const auto grow = [&](void **ptr, size_t &have, size_t need) -> hipError_t {
        if (need <= have)
            return hipSuccess;
        hipError_t e = hipStreamSynchronize(ctx->stream); // nothing may still be using the old buffer
        if (e != hipSuccess)
            return e;
        (void)hipFree(*ptr);
        *ptr = nullptr;
        have = 0;
        e = hipMalloc(ptr, need);
        if (e == hipSuccess)
            have = need;
        return e;
    };
    const int64_t R1 = n / 4096;
    hipError_t e = grow((void **)&w.Y, w.Y_bytes, (size_t)pairs * (size_t)n * sizeof(double2));
    if (e == hipSuccess && tables)
        e = grow((void **)&w.T, w.T_bytes, (size_t)pairs * (size_t)n * sizeof(double2));
    if (e == hipSuccess)
        e = grow((void **)&w.part, w.part_bytes, (size_t)(2 * pairs) * (size_t)R1 * 2 * sizeof(double));
    if (e == hipSuccess)
        e = grow((void **)&w.snorm, w.snorm_bytes, (size_t)(2 * pairs) * 4 * sizeof(double));
    if (e == hipSuccess)
        e = grow((void **)&w.sfin, w.sfin_bytes, (size_t)(2 * pairs) * sizeof(double));
    if (e == hipSuccess)
        e = grow((void **)&w.sfin_x, w.sfin_x_bytes, (size_t)(2 * pairs) * sizeof(double));
    if (e == hipSuccess)
        e = grow((void **)&w.amax, w.amax_bytes, (size_t)pairs * (size_t)R1 * 8 * sizeof(double));
    if (e != hipSuccess)
        return fail(MUSE_ERR_NOMEM, "long-series work buffers: %s", hipGetErrorString(e));
    return MUSE_OK;
}

HugeParams huge_base(muse_ctx *ctx, int logn)
{
    HugeParams p{};
    p.logn = logn;
    p.n = 1 << logn;
    p.R1 = p.n / 4096;
    p.thi = ctx->huge.thi[logn - HUGE_MIN_LOGN];
    p.tlo = ctx->huge.tlo[logn - HUGE_MIN_LOGN];
    p.g2 = ctx->g2;
    p.g3a = ctx->g3a;
    p.g3b = ctx->g3b;
    p.Y = ctx->huge.Y;
    p.part = ctx->huge.part;
    p.snorm = ctx->huge.snorm;
    p.sfin = ctx->huge.sfin;
    p.amax = ctx->huge.amax;
    p.pre_scale = 1.0;
#ifdef MUSE_HUGE_ABL
    p.abl = getenv("MUSE_HUGE_ABL") ? atoi(getenv("MUSE_HUGE_ABL")) : 0;
#endif
    return p;
}

int64_t pairs_per_batch(const muse_ctx *ctx, int64_t n)
{
    // (measurement hook muse_test_huge_batch_mb, tools/huge_bench.py: the work buffer of one batch; 0 = the built-in 128 MB)
    const size_t bytes = ctx->huge_batch_mb != 0 ? (size_t)std::abs(ctx->huge_batch_mb) << 20 : HUGE_BATCH_BYTES;
    return std::max<int64_t>(1, (int64_t)(bytes / ((size_t)n * sizeof(double2))));
}

} // namespace

void huge_free(muse_ctx *ctx)
{
    HugeWork &w = ctx->huge;
    for (int i = 0; i < 4; i++) {
        (void)hipFree(w.thi[i]);
        (void)hipFree(w.tlo[i]);
    }
    (void)hipFree(w.Y);
    (void)hipFree(w.T);
    (void)hipFree(w.part);
    (void)hipFree(w.snorm);
    (void)hipFree(w.sfin);
    (void)hipFree(w.sfin_x);
    (void)hipFree(w.amax);
    (void)hipFree(w.Y2);
    (void)hipFree(w.amax2);
    if (w.stream2) {
        (void)hipStreamSynchronize(w.stream2);
        (void)hipStreamDestroy(w.stream2);
    }
    if (w.fork)
        (void)hipEventDestroy(w.fork);
    if (w.join)
        (void)hipEventDestroy(w.join);
    w = HugeWork{};
}

// x = zNormalize(ref) / (N - 1), zeroPad, FFT (muse_batch.go:38-47): X[0 .. n / 2] and the lane-ordered multiplier rows
// table[k] = conj(X[k]) / n of the batch; *zero_std = 1 when sigma(ref) is 0 (or not a number)
int huge_reference(muse_ctx *ctx, const double *ref_host, int N, int n, double2 *X, double2 *table, int *zero_std)
{
    const int logn = ilog2(n);
    std::lock_guard<std::mutex> lock(ctx->huge_mu);
    int rc = huge_ensure(ctx, logn, 1, false);
    if (rc)
        return rc;
    double *dref = nullptr;
    HIP_TRY(dmalloc(ctx, &dref, (size_t)N * sizeof(double)));
    hipError_t e = hipMemcpyAsync(dref, ref_host, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    HugeParams p = huge_base(ctx, logn);
    p.rows = dref;
    p.stride = N;
    p.first = 0;
    p.count = 1;
    p.N = N;
    p.solo = 1;
    p.normalize = 1;
    p.pre_scale = 1.0 / (double)(N - 1);
    p.table_out = table;
    p.table_scale = 1.0 / (double)n;
    p.X_out = X;
    if (e == hipSuccess)
        e = launch_huge(p, HUGE_STAGE_STATS | HUGE_STAGE_SWEEP1 | HUGE_STAGE_ROWS_FORWARD, ctx->stream);
    double flag = 0.0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(&flag, p.sfin, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    dfree(ctx, dref);
    HIP_TRY(e);
    *zero_std = flag != 0.0 ? 1 : 0;
    return MUSE_OK;
}

// the all-scores pass of a batch whose FFT length is above 65 536: (lag, signed max value) of every series of the group.
// The series' statistics are the GROUP's (muse_group::hstats): the rows are immutable, so the first pass over them computes first
// sample, mean, 1 / sigma and flag per row and every later Run -- and every other reference -- skips that read of the rows.
int huge_score(muse_batch *b)
{
    muse_ctx *ctx = b->ctx;
    muse_group *g = b->g;
    const int64_t M = g->M, n = b->n;
    const int64_t ppb = pairs_per_batch(ctx, n);
    std::lock_guard<std::mutex> lock(ctx->huge_mu);
    int rc = huge_ensure(ctx, b->logn, ppb, false);
    if (rc)
        return rc;
    if (g->hstats_cap < M) { // (grown with the group; what was computed is recomputed: rare)
        dfree(ctx, g->hstats);
        g->hstats = nullptr;
        g->hstats_cap = g->hstats_rows = 0;
        const int64_t cap = std::max<int64_t>(M, g->cap);
        HIP_TRY(dmalloc(ctx, &g->hstats, (size_t)cap * 4 * sizeof(double)));
        g->hstats_cap = cap;
    }
    // two streams: the batches alternate between the batch's stream and a second one, each with its own work buffer (half the
    // batch size each: the same footprint in the Infinity Cache), so the ragged end of one batch's kernels runs under the next
    // batch's; the second stream forks behind the pass's start and joins before its end (the timer's bracket covers both)
    const bool dual = ctx->huge_batch_mb >= 0 && M > 2 * ppb;
    const int64_t spb = dual ? std::max<int64_t>(2, ppb) : 2 * ppb; // series per batch (dual: ppb / 2 pairs per stream)
    HugeWork &w = ctx->huge;
    if (dual) {
        const int64_t R1 = n / 4096, pairs2 = (spb + 1) / 2;
        const auto grow2 = [&](void **ptr, size_t &have, size_t need) -> hipError_t {
            if (need <= have)
                return hipSuccess;
            if (w.stream2)
                (void)hipStreamSynchronize(w.stream2);
            (void)hipFree(*ptr);
            *ptr = nullptr;
            have = 0;
            const hipError_t e = hipMalloc(ptr, need);
            if (e == hipSuccess)
                have = need;
            return e;
        };
        HIP_TRY(grow2((void **)&w.Y2, w.Y2_bytes, (size_t)pairs2 * (size_t)n * sizeof(double2)));
        HIP_TRY(grow2((void **)&w.amax2, w.amax2_bytes, (size_t)pairs2 * (size_t)R1 * 8 * sizeof(double)));
        if (!w.stream2)
            HIP_TRY(hipStreamCreateWithFlags(&w.stream2, hipStreamNonBlocking));
        if (!w.fork)
            HIP_TRY(hipEventCreateWithFlags(&w.fork, hipEventDisableTiming));
        if (!w.join)
            HIP_TRY(hipEventCreateWithFlags(&w.join, hipEventDisableTiming));
    }
    LaunchTimer timer(ctx, false, b->stream()); // (one bracket around the pass: its kernels are one unit of work per batch of pairs)
    HIP_TRY(timer.begin());
    for (int64_t first = g->hstats_rows; first < M; first += 2 * ppb) { // rows without statistics yet
        HugeParams p = huge_base(ctx, b->logn);
        p.rows = g->rows;
        p.stride = g->stride;
        p.first = first;
        p.count = (int)std::min<int64_t>(2 * ppb, M - first);
        p.N = b->N;
        p.normalize = 1;
        p.snorm = g->hstats + first * 4;
        HIP_TRY(launch_huge(p, HUGE_STAGE_STATS_ONLY, b->stream()));
    }
    g->hstats_rows = M;
    if (dual) {
        HIP_TRY(hipEventRecord(w.fork, b->stream()));
        HIP_TRY(hipStreamWaitEvent(w.stream2, w.fork, 0));
    }
    int which = 0;
    for (int64_t first = 0; first < M; first += spb, which ^= 1) {
        HugeParams p = huge_base(ctx, b->logn);
        p.rows = g->rows;
        p.stride = g->stride;
        p.first = first;
        p.count = (int)std::min<int64_t>(spb, M - first);
        p.N = b->N;
        p.solo = 0;
        p.normalize = 1;
        p.snorm = g->hstats + first * 4;
        p.table = b->xcp;
        p.table_stride = 0;
        p.mv = b->mv;
        p.lag = b->lag;
        const bool second = dual && which;
        if (second) {
            p.Y = w.Y2;
            p.amax = w.amax2;
        }
        HIP_TRY(launch_huge(p, HUGE_STAGE_SWEEP1 | HUGE_STAGE_ROWS | HUGE_STAGE_SWEEP2 | HUGE_STAGE_FINAL, second ? w.stream2 : b->stream()));
    }
    if (dual) {
        HIP_TRY(hipEventRecord(w.join, w.stream2));
        HIP_TRY(hipStreamWaitEvent(b->stream(), w.join, 0));
    }
    HIP_TRY(timer.end());
    return MUSE_OK;
}

// M independent pairs (x_i, y_i), one series per transform: every x its own multiplier table (xCorr, xcorr.go:102-153; with
// normalize_x, x_scale = 1 / (N - 1) and cc_scale = 1 / n also xCorrWithX with its cc slice, xcorr.go:160-197).
// Device pointers throughout; cc (optional) M x n.
int huge_pairs(muse_ctx *ctx, const double *xrows, int64_t xstride, int Nx, int normalize_x, double x_scale, const double *yrows,
               int64_t ystride, int Ny, int normalize_y, int64_t M, int n, double cc_scale, double *mv, int *lag, int *nil, double *cc)
{
    const int logn = ilog2(n);
    const int64_t ppb = pairs_per_batch(ctx, n);
    std::lock_guard<std::mutex> lock(ctx->huge_mu);
    int rc = huge_ensure(ctx, logn, std::min<int64_t>(ppb, M), true);
    if (rc)
        return rc;
    for (int64_t first = 0; first < M; first += ppb) {
        const int count = (int)std::min<int64_t>(ppb, M - first);
        HugeParams px = huge_base(ctx, logn);
        px.rows = xrows;
        px.stride = xstride;
        px.first = first;
        px.count = count;
        px.N = Nx;
        px.solo = 1;
        px.normalize = normalize_x;
        px.pre_scale = x_scale;
        px.sfin = ctx->huge.sfin_x;
        px.table_out = ctx->huge.T;
        px.table_scale = cc_scale;
        HIP_TRY(launch_huge(px, HUGE_STAGE_STATS | HUGE_STAGE_SWEEP1 | HUGE_STAGE_ROWS_FORWARD, ctx->stream));
        HugeParams py = huge_base(ctx, logn);
        py.rows = yrows;
        py.stride = ystride;
        py.first = first;
        py.count = count;
        py.N = Ny;
        py.solo = 1;
        py.normalize = normalize_y;
        py.table = ctx->huge.T;
        py.table_stride = n;
        py.sfin_x = normalize_x ? ctx->huge.sfin_x : nullptr;
        py.mv = mv;
        py.lag = lag;
        py.nil = nil;
        py.cc_out = cc;
        HIP_TRY(launch_huge(py, HUGE_STAGE_STATS | HUGE_STAGE_SWEEP1 | HUGE_STAGE_ROWS | HUGE_STAGE_SWEEP2 | HUGE_STAGE_FINAL, ctx->stream));
    }
    return MUSE_OK;
}

extern "C" int muse_test_huge_batch_mb(muse_ctx *ctx, int32_t megabytes)
{
    if (!ctx || megabytes < -4096 || megabytes > 4096)
        return fail(MUSE_ERR_INVALID, "batch size 0 (built-in) ... 4096 MB; negative: that size with every batch on one stream");
    ctx->huge_batch_mb = megabytes;
    return MUSE_OK;
}
