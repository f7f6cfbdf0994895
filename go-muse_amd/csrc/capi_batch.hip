// capi_batch.hip -- NewBatch (muse_batch.go:23-52): reference spectrum and tables; the all-scores pass and its kernel selection
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
#include "capi_internal.h"
#include "xcorr_huge.h"

using namespace muse;


// ------------------------------------------------------------------- batch
int ilog2(int64_t n)
{
    int l = 0;
    while (((int64_t)1 << l) < n)
        l++;
    return l;
}

// device reference spectrum for (ref, N) at FFT length n: fills X, xc
int build_spectrum(muse_ctx *ctx, const double *ref_host, int N, int n, int normalize, double x_scale,
                          double xc_scale, double2 *X, double2 *xc, float2 *xcf, double *xs, int *zero_std)
{
    double *dref = nullptr;
    int *dstat = nullptr;
    double2 *dscr = nullptr; // n > 8192: global work buffer for the radix-2 passes
    HIP_TRY(dmalloc(ctx, &dref, (size_t)N * sizeof(double)));
    hipError_t e = dmalloc(ctx, &dstat, sizeof(int));
    if (e == hipSuccess && n > GENERIC_LDS_MAX_N)
        e = dmalloc(ctx, &dscr, (size_t)n * sizeof(double2));
    if (e == hipSuccess)
        e = hipMemcpyAsync(dref, ref_host, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = launch_ref_spectrum(dref, N, n, ilog2(n), normalize, x_scale, xc_scale, ctx->twm, X, xc, xcf, xs, dscr,
                                dstat, ctx->stream);
    int st = 0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(&st, dstat, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    dfree(ctx, dref);
    dfree(ctx, dstat);
    dfree(ctx, dscr);
    HIP_TRY(e);
    *zero_std = st;
    return MUSE_OK;
}

hipError_t ensure_gscratch(muse_ctx *ctx, int64_t n, int slices_per_cu)
{
    if (n < GENERIC_LDS_MAX_N) // generic kernel above 8192: one slice per workgroup; Stockham from 8192: up to two
        return hipSuccess;
    const size_t need = (size_t)ctx->num_cus * (size_t)slices_per_cu * (size_t)n;
    std::lock_guard<std::mutex> lock(ctx->stage_mu); // launches that use the buffer hold the same lock (muse_batch_score)
    if (need <= ctx->gscratch_elems)
        return hipSuccess;
    hipError_t e = hipStreamSynchronize(ctx->stream); // nothing may still be using the old buffer
    if (e != hipSuccess)
        return e;
    (void)hipFree(ctx->gscratch);
    ctx->gscratch = nullptr;
    ctx->gscratch_elems = 0;
    e = hipMalloc(&ctx->gscratch, need * sizeof(double2));
    if (e == hipSuccess)
        ctx->gscratch_elems = need;
    return e;
}

// the base table of the long-series sweeps' twiddles, [4096] W_n^(m2) (n = 16384, 32768, 65536): built on first use per length
// (rare, so always under the lock -- no unlocked read of the pointer another thread may be storing)
hipError_t ensure_twl(muse_ctx *ctx, int64_t n)
{
    const int li = ilog2(n) - 14;
    if (li < 0 || li > 2)
        return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(ctx->stage_mu);
    if (ctx->twl[li])
        return hipSuccess;
    std::vector<double2> tl(4096);
    for (int m2 = 0; m2 < 4096; m2++)
        fill_twiddle(tl, (size_t)m2, (long long)m2, n);
    double2 *d = nullptr;
    hipError_t e = hipMalloc(&d, tl.size() * sizeof(double2));
    if (e == hipSuccess)
        e = hipMemcpy(d, tl.data(), tl.size() * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        ctx->twl[li] = d;
    else
        (void)hipFree(d);
    return e;
}

void adopt_spectrum(muse_batch *b)
{
    b->X = b->sp->X;
    b->xc = b->sp->xc;
    b->xcp = b->sp->xcp;
    b->xcw = b->sp->xcw;
    b->xcf = b->sp->xcf;
    b->xs = b->sp->xs;
    b->c1 = b->sp->c1;
}

extern "C" int muse_batch_create(muse_ctx *ctx, muse_group *g, const double *ref, int32_t N, muse_batch **out)
{
    if (!out)
        return fail(MUSE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (N < 1) // muse.go:24-26
        return fail(MUSE_ERR_EMPTY, "Reference series length must be greater than zero");
    if (!g || !ref || g->ctx != ctx)
        return fail(MUSE_ERR_INVALID, "bad batch arguments");
    if (g->N != N) // muse_batch.go:24-28
        return fail(MUSE_ERR_LENGTH, "comparison group series does not have the same length as the reference (%d vs %d)",
                    g->N, N);
    if (N < 2)
        return fail(MUSE_ERR_INVALID, "series length 1 has no sample standard deviation");
    const int64_t n = muse_next_pow2((double)N); // muse_batch.go:35
    if (n > HUGE_MAX_N)
        return fail(MUSE_ERR_UNSUPPORTED, "FFT length %lld > %d is not built", (long long)n, HUGE_MAX_N);
    const bool huge = n > GENERIC_MAX_N; // series longer than 65 536 samples: xcorr_huge.hip
    muse_batch *b = new (std::nothrow) muse_batch();
    if (!b)
        return fail(MUSE_ERR_NOMEM, "host allocation failed");
    b->ctx = ctx;
    b->g = g;
    g->refs.fetch_add(1);
    ctx->refs.fetch_add(1);
    b->N = N;
    b->n = (int32_t)n;
    b->logn = ilog2(n);
    hipError_t e = dmalloc(ctx, &b->ovf_count, 2 * sizeof(int)); // [0] overflow-pair count, [1] dynamic work counter
    if (e == hipSuccess && !huge)
        e = ensure_gscratch(ctx, n);
    muse_spectrum *sp = new (std::nothrow) muse_spectrum();
    if (!sp)
        e = hipErrorOutOfMemory;
    b->sp = sp;
    if (e == hipSuccess)
        e = dmalloc(ctx, &sp->X, (size_t)(n / 2 + 1) * sizeof(double2));
    if (huge) {
        // the spectrum in natural order (muse_batch_spectrum) and as the lane-ordered multiplier rows of huge_rows: the only
        // two tables these lengths use (no fp32 screening copy, no time-domain copy, no indicator correlation)
        if (e == hipSuccess)
            e = dmalloc(ctx, &sp->xcp, (size_t)n * sizeof(double2));
        if (e != hipSuccess) {
            muse_batch_free(b);
            return fail(MUSE_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
        }
        adopt_spectrum(b);
        int zero = 0;
        rc = huge_reference(ctx, ref, N, (int)n, sp->X, sp->xcp, &zero);
        if (rc) {
            muse_batch_free(b);
            return rc;
        }
        if (zero) { // muse_batch.go:39-41
            muse_batch_free(b);
            return fail(MUSE_ERR_ZERO_STD, "Invalid input query, Standard deviation of zero");
        }
        *out = b;
        return MUSE_OK;
    }
    if (e == hipSuccess)
        e = dmalloc(ctx, &sp->xc, (size_t)n * sizeof(double2));
    if (e == hipSuccess)
        e = dmalloc(ctx, &sp->xcf, (size_t)n * sizeof(float2));
    if (e == hipSuccess)
        e = dmalloc(ctx, &sp->xs, (size_t)n * sizeof(double));
    const bool long_n = n == 16384 || n == 32768 || n == 65536; // xcorr_long.hip: spectrum rows in lane order, sweep twiddles
    if (e == hipSuccess && (n == 4096 || long_n))
        e = dmalloc(ctx, &sp->xcp, (size_t)n * sizeof(double2));
    if (e == hipSuccess && n == 8192) // xcorr_fused_real8k: the spectrum at its threads' bins, lane-ordered
        e = dmalloc(ctx, &sp->xcp, (size_t)4096 * sizeof(double2));
    if (e == hipSuccess && n == 32768) // xcorr_real.hip, the 16 x 1024 split: xc at the threads' lower eight bins and their mirror bins
        e = dmalloc(ctx, &sp->xcw, (size_t)16385 * sizeof(double2)); // ([16384]: the reference's bin n / 4)
    if (e == hipSuccess && (n == 4096 || long_n) && N < n)
        e = dmalloc(ctx, &sp->c1, (size_t)n * sizeof(double));
    if (e != hipSuccess) {
        muse_batch_free(b);
        return fail(MUSE_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    adopt_spectrum(b);
    int zero = 0;
    // x = zNormalize(ref) / (N-1), zeroPad, FFT   (muse_batch.go:38-47)
    rc = build_spectrum(ctx, ref, N, (int)n, 1, 1.0 / (double)(N - 1), 1.0 / (double)n, b->X, b->xc, b->xcf, b->xs,
                        &zero);
    if (rc) {
        muse_batch_free(b);
        return rc;
    }
    if (n == 4096) {
        e = launch_lane_order(b->xc, b->xcp, ctx->stream);
        if (e == hipSuccess && b->c1)
            e = launch_indicator_corr(b->xs, 4096, 4096 - N, b->c1, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            muse_batch_free(b);
            return fail(MUSE_ERR_HIP, "lane-order table: %s", hipGetErrorString(e));
        }
    }
    if (n == 8192) {
        e = launch_real8k_tables(b->xc, b->xcp, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            muse_batch_free(b);
            return fail(MUSE_ERR_HIP, "n = 8192 lane-order table: %s", hipGetErrorString(e));
        }
    }
    if (long_n) {
        const int R1 = (int)(n / 4096);
        e = ensure_twl(ctx, n);
        if (e == hipSuccess)
            e = launch_lane_order_rows(b->xc, b->xcp, R1, ctx->stream);
        if (e == hipSuccess && b->xcw)
            e = launch_real_split_tables(b->xc, b->xcw, ctx->stream);
        if (e == hipSuccess && b->c1)
            e = launch_indicator_corr(b->xs, (int)n, (int)(n - N), b->c1, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            muse_batch_free(b);
            return fail(MUSE_ERR_HIP, "long-series tables: %s", hipGetErrorString(e));
        }
    }
    if (zero) { // muse_batch.go:39-41
        muse_batch_free(b);
        return fail(MUSE_ERR_ZERO_STD, "Invalid input query, Standard deviation of zero");
    }
    *out = b;
    return MUSE_OK;
}

extern "C" int muse_batch_create_like(muse_batch *src, muse_group *g, muse_batch **out)
{
    if (!out)
        return fail(MUSE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!src || !g || g->ctx != src->ctx)
        return fail(MUSE_ERR_INVALID, "bad batch arguments");
    if (g->N != src->N) // muse_batch.go:24-28 / muse.go:68-70
        return fail(MUSE_ERR_LENGTH, "comparison group series does not have the same length as the reference (%d vs %d)",
                    g->N, src->N);
    muse_ctx *ctx = src->ctx;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    muse_batch *b = new (std::nothrow) muse_batch();
    if (!b)
        return fail(MUSE_ERR_NOMEM, "host allocation failed");
    b->ctx = ctx;
    b->g = g;
    g->refs.fetch_add(1);
    ctx->refs.fetch_add(1);
    b->N = src->N;
    b->n = src->n;
    b->logn = src->logn;
    b->sp = src->sp;
    b->sp->refs.fetch_add(1);
    adopt_spectrum(b);
    hipError_t e = dmalloc(ctx, &b->ovf_count, 2 * sizeof(int));
    if (e != hipSuccess) {
        muse_batch_free(b);
        return fail(MUSE_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    *out = b;
    return MUSE_OK;
}

extern "C" int muse_batch_fft_len(muse_batch *b, int32_t *n)
{
    if (!b || !n)
        return fail(MUSE_ERR_INVALID, "NULL argument");
    *n = b->n;
    return MUSE_OK;
}

extern "C" int muse_batch_spectrum(muse_batch *b, double *out)
{
    if (!b || !out)
        return fail(MUSE_ERR_INVALID, "NULL argument");
    int rc = use_device(b->ctx);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(out, b->X, (size_t)(b->n / 2 + 1) * sizeof(double2), hipMemcpyDeviceToHost));
    return MUSE_OK;
}

int ensure_scores(muse_batch *b)
{
    const int64_t M = b->g->M;
    if (M <= b->score_cap)
        return MUSE_OK;
    dfree(b->ctx, b->mv);
    dfree(b->ctx, b->lag);
    b->mv = nullptr;
    b->lag = nullptr;
    b->score_cap = 0;
    HIP_TRY(dmalloc(b->ctx, &b->mv, (size_t)M * sizeof(double)));
    HIP_TRY(dmalloc(b->ctx, &b->lag, (size_t)M * sizeof(int)));
    b->score_cap = M;
    return MUSE_OK;
}

// the launch parameters every fused kernel shares for batch b (group flushed, M > 0, scores allocated)
FusedParams base_params(muse_batch *b)
{
    muse_ctx *ctx = b->ctx;
    const int64_t M = b->g->M;
    FusedParams p{};
    p.rows = b->g->f32 ? nullptr : b->g->rows;
    p.rows32 = b->g->f32 ? b->g->rows32 : nullptr;
    p.M = M;
    p.stride = b->g->stride;
    p.npairs = (M + 1) / 2;
    p.N = b->N;
    p.n = b->n;
    p.logn = b->logn;
    p.normalize_y = 1;
    p.xc = b->xc;
    p.tw1 = ctx->tw1;
    p.tw2 = ctx->tw2;
    p.twm = ctx->twm;
    p.gscratch = ctx->gscratch;
    p.gscratch_slices = b->n > 0 ? (long long)(ctx->gscratch_elems / (size_t)b->n) : 0;
    p.mv = b->mv;
    p.lag = b->lag;
    p.cc_out = nullptr;
    p.nil_out = nullptr;
    p.g2 = ctx->g2;
    p.g3a = ctx->g3a;
    p.g3b = ctx->g3b;
    p.gsmall = (b->logn >= 9 && b->logn <= 11) ? ctx->gsmall[b->logn - 9] : (b->logn == 13 || b->logn == 14) ? ctx->gsmall[b->logn - 10] : nullptr;
    p.xcp = b->xcp;
    p.xcw = b->xcw;
    p.gsmall_b = ctx->gsmall[1];
    p.wsplit = ctx->wsplit;
    p.c1 = b->c1;
    p.twl = (b->logn >= 14 && b->logn <= 16) ? ctx->twl[b->logn - 14] : nullptr;
    p.tw1f = ctx->tw1f;
    p.twmf = ctx->twmf;
    p.tw2f = ctx->tw2f;
    p.xcf = b->xcf;
    p.xs = b->xs;
    p.screen_delta = ctx->screen_delta;
#ifdef MUSE_REAL64_STAMPS
    // diagnostic build (tools/ablate/ab_real64_stamps.sh): phase stamps of xcorr_fused_real64k, dumped when the context goes
    if (!ctx->dbg_stamps && hipMalloc(&ctx->dbg_stamps, (size_t)ctx->num_cus * 4 * 16 * 16 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(ctx->dbg_stamps, 0, (size_t)ctx->num_cus * 4 * 16 * 16 * sizeof(unsigned long long));
    p.dbg = ctx->dbg_stamps;
#endif
    return p;
}

// The kernel the all-scores pass of batch b takes (ctx->variant 0 = automatic selection; the others are test hooks,
// muse_hip_test.h): decided HERE, for muse_batch_score's launch and for the name muse_batch_kernel_name reports.
struct KernelChoice {
    int variant = KERNEL_GENERIC;
    const double2 *gsmall = nullptr; // the pass table the kernel wants instead of base_params' (nullptr: keep)
    const char *error = nullptr;
};
static KernelChoice choose_kernel(muse_batch *b, long long npairs)
{
    muse_ctx *ctx = b->ctx;
    const int64_t M = b->g->M;
    const bool has_twl = b->logn >= 14 && b->logn <= 16 && ctx->twl[b->logn - 14];
    KernelChoice kc;
    if (b->n == 4096) {
        switch (ctx->variant) {
        case 0: case 10: kc.variant = KERNEL_R16_FOLD; break; // fastest measured (profiles/)
        case 7: kc.variant = KERNEL_R16_OCC3; break;          // rescales both series before the shared transform
        default: kc.variant = KERNEL_GENERIC; break;
        }
        if (b->g->f32 && kc.variant == KERNEL_GENERIC) {
            kc.error = "the generic kernel does not read float32-storage groups";
            return kc;
        }
        if (kc.variant == KERNEL_R16_FOLD && b->N != 4096 && !b->c1) // (N < n needs the batch's correction table)
            kc.variant = KERNEL_R16_OCC3;
        // a group of mixed-unit series (sigmas far apart inside most pairs) makes the default kernel hand most
        // pairs to kernel 7 anyway: once a pass over these rows has shown that, go there directly
        if (kc.variant == KERNEL_R16_FOLD && ctx->variant == 0 && b->handoff_host && b->handoff_M == M &&
            b->handoff_rewrites == b->g->rewrites && (long long)*(volatile int *)b->handoff_host * 8 > npairs)
            kc.variant = KERNEL_R16_OCC3;
    } else if (b->g->f32 && !(((b->n >= 512 && b->n <= 2048) || b->n == 8192 || b->n == 16384) && (ctx->variant == 0 || ctx->variant == 12))) {
        kc.error = "float32-storage groups run on the default kernels only (FFT lengths 512 ... 16384)";
    } else if (b->n == 8192 && (ctx->variant == 0 || ctx->variant == 14) && !b->g->f32 && b->xcp) {
        kc.variant = KERNEL_REAL; // one real series per 256-thread workgroup on the n = 4096 kernel's transforms (xcorr_real.hip)
    } else if (b->n == 16384 && (ctx->variant == 0 || ctx->variant == 14) && !b->g->f32 && ctx->gsmall[3]) {
        kc.variant = KERNEL_REAL; // one real series per 512-thread workgroup on the 8192-point complex transform, two workgroups per CU
        kc.gsmall = ctx->gsmall[3];
    } else if (b->n == 32768 && (ctx->variant == 0 || ctx->variant == 15) && !b->g->f32 && ctx->gsmall[4] && b->xcw && ctx->wsplit) {
        // one real series per 1024-thread workgroup with each 16384-point transform as 16 x 1024 (wave-local 1024-point transforms around one
        // workgroup transpose): + 1 ... 4 % over test hook 14's three-transpose form on every box measured (profiles/r05_real_transform.txt)
        kc.variant = KERNEL_REAL_SPLIT;
        kc.gsmall = ctx->gsmall[4];
    } else if ((b->n == 32768 || (b->n == 65536 && (b->N == b->n || b->c1))) && (ctx->variant == 0 || ctx->variant == 14) && !b->g->f32 && ctx->gsmall[4]) {
        // one real series per 1024-thread workgroup on the 16384-point complex transform (xcorr_real.hip): n = 32768 never leaves the CU,
        // n = 65536 in two passes with 1 MB parked per series (3 x the algorithmic bytes; the four-step kernel: 5 x)
        kc.variant = KERNEL_REAL;
        kc.gsmall = ctx->gsmall[4];
    } else if (b->xcp && has_twl && (b->N == b->n || b->c1) && ((b->n >= 32768 && ctx->variant == 0) || (b->n >= 16384 && ctx->variant == 13))) {
        kc.variant = KERNEL_LONG; // four-step, 4096-point rows on the n = 4096 kernel's transforms (xcorr_long.hip)
    } else if (((b->n >= 512 && b->n <= 2048) || b->n == 8192 || b->n == 16384) && (ctx->variant == 0 || ctx->variant == 12)) {
        kc.variant = KERNEL_SMALL; // half-round transposes at 16 waves per CU (xcorr_small.hip)
    } else if (((b->n >= 512 && b->n <= 2048) || b->n >= 8192) && (ctx->variant == 0 || ctx->variant == 11)) {
        kc.variant = KERNEL_STOCKHAM; // radix-16 Stockham through LDS / global scratch (xcorr_stockham.hip)
    }
    return kc;
}

extern "C" int muse_batch_score(muse_batch *b)
{
    if (!b)
        return fail(MUSE_ERR_INVALID, "NULL batch");
    muse_ctx *ctx = b->ctx;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    rc = group_ready(b->g, b->stream()); // rows still in the staging buffer are uploaded (copy stream) ahead of the kernel
    if (rc)
        return rc;
    const int64_t M = b->g->M;
    if (M == 0)
        return MUSE_OK;
    const hipStream_t st = b->stream();
    rc = ensure_scores(b);
    if (rc)
        return rc;
    if (b->n > GENERIC_MAX_N) { // series longer than 65 536 samples: a sequence of chip-wide kernels per batch of pairs (xcorr_huge.hip)
        if (b->g->f32)
            return fail(MUSE_ERR_UNSUPPORTED, "float32-storage groups run on the default kernels only (FFT lengths 512 ... 16384)");
        b->scores_exact = true;
        return huge_score(b);
    }
    // long series work in the context's scratch buffer: its pointer must not be swapped (a concurrent
    // muse_batch_create growing it) between reading it and enqueueing the launch
    std::unique_lock<std::mutex> scratch_lock(ctx->stage_mu, std::defer_lock);
    if (b->n >= GENERIC_LDS_MAX_N)
        scratch_lock.lock();
    FusedParams p = base_params(b);
    b->scores_exact = true;
    // kernel selection: ctx->variant 0 = auto; the others are test hooks (muse_hip_test.h).  One helper decides, for the launch
    // here and for the name muse_batch_kernel_name reports (bench.py keys its counters by it)
    const KernelChoice kc = choose_kernel(b, p.npairs);
    if (kc.error)
        return fail(MUSE_ERR_UNSUPPORTED, "%s", kc.error);
    const int variant = kc.variant;
    if (kc.gsmall)
        p.gsmall = kc.gsmall;
    if (variant == KERNEL_GENERIC && b->n <= GENERIC_LDS_MAX_N)
        p.gscratch = nullptr; // the generic kernel takes a non-NULL scratch pointer as "work in global memory"
    LaunchTimer timer(ctx, false, st); // (brackets the fused launch alone: not the counter reset in front of it, not the redo launch behind it)
    LaunchTimer redo_timer(ctx, true, st); // the launch that redoes the listed pairs: its own sum (muse_ctx_redo_time)
    if (variant == KERNEL_R16_FOLD) {
        // pairs with a NaN/Inf series or with sigmas too far apart for one shared transform are listed by the kernel
        // (once per such series: 2 entries per pair) and redone by the rescaling kernel right behind it (no host round
        // trip: the count stays on the device and bounds the second launch's loop)
        if (2 * p.npairs > b->ovf_cap) {
            dfree(ctx, b->ovf_list);
            b->ovf_list = nullptr;
            b->ovf_cap = 0;
            HIP_TRY(dmalloc(ctx, &b->ovf_list, (size_t)(2 * p.npairs) * sizeof(long long)));
            b->ovf_cap = 2 * p.npairs;
        }
        p.ovf_count = b->ovf_count;
        p.work_counter = b->ovf_count + 1;
        p.ovf_list = b->ovf_list;
        HIP_TRY(hipMemsetAsync(b->ovf_count, 0, 2 * sizeof(int), st));
        HIP_TRY(timer.begin());
        HIP_TRY(launch_fused(p, variant, ctx->num_cus, st));
        HIP_TRY(timer.end());
        FusedParams q = p;
        q.pair_list = b->ovf_list;
        q.pair_count = b->ovf_count;
        // a dense list (the same threshold as the hand-off rule above) makes the redo kernel redo EVERY pair: the results of a
        // mixed-unit group then come from kernel 7 in this pass exactly as in the later ones that go there directly
        q.dense_total = ctx->variant == 0 ? p.npairs : 0;
        // grid size only (the loop bound is *pair_count): one resident set, so a group with MANY listed pairs
        // (mixed-unit metrics: sigmas far apart) is redone at full width; an empty list costs a few microseconds
        q.npairs = std::min<long long>(p.npairs, (long long)ctx->num_cus * 3);
        HIP_TRY(redo_timer.begin());
        HIP_TRY(launch_fused(q, KERNEL_R16_OCC3, ctx->num_cus, st));
        HIP_TRY(redo_timer.end());
        // (small groups: a pinned allocation costs more than it can save; the same rows again: the count is known, no second copy --
        // measured: the 4-byte copy was not visible in a Run over 10 000 x 4096 either way)
        if (p.npairs >= 1024 && !(b->handoff_host && b->handoff_M == M && b->handoff_rewrites == b->g->rewrites)) {
            if (!b->handoff_host)
                HIP_TRY(hmalloc(ctx, &b->handoff_host, sizeof(int)));
            *b->handoff_host = 0;
            b->handoff_M = M;
            b->handoff_rewrites = b->g->rewrites;
            HIP_TRY(hipMemcpyAsync(b->handoff_host, b->ovf_count, sizeof(int), hipMemcpyDeviceToHost, st));
        }
    } else if (variant == KERNEL_LONG) {
        // as above: NaN / Inf and sigma-spread pairs are listed (one entry per pair) and redone by the four-step kernel that
        // isolates and rescales the series first
        if (2 * p.npairs > b->ovf_cap) {
            dfree(ctx, b->ovf_list);
            b->ovf_list = nullptr;
            b->ovf_cap = 0;
            HIP_TRY(dmalloc(ctx, &b->ovf_list, (size_t)(2 * p.npairs) * sizeof(long long)));
            b->ovf_cap = 2 * p.npairs;
        }
        p.ovf_count = b->ovf_count;
        p.ovf_list = b->ovf_list;
        HIP_TRY(hipMemsetAsync(b->ovf_count, 0, 2 * sizeof(int), st));
        HIP_TRY(timer.begin());
        HIP_TRY(launch_fused(p, variant, ctx->num_cus, st));
        HIP_TRY(timer.end());
        FusedParams q = p;
        q.pair_list = b->ovf_list;
        q.pair_count = b->ovf_count;
        q.npairs = std::min<long long>(p.npairs, (long long)ctx->num_cus * STOCKHAM_GLOBAL_WGS_PER_CU);
        HIP_TRY(redo_timer.begin());
        HIP_TRY(launch_fused(q, KERNEL_STOCKHAM, ctx->num_cus, st));
        HIP_TRY(redo_timer.end());
    } else {
        HIP_TRY(timer.begin());
        HIP_TRY(launch_fused(p, variant, ctx->num_cus, st));
        HIP_TRY(timer.end());
    }
    return MUSE_OK;
}

extern "C" int muse_batch_scores(muse_batch *b, int32_t *lag, double *mv)
{
    int rc = muse_batch_score(b);
    if (rc)
        return rc;
    const int64_t M = b->g->M;
    if (M == 0)
        return MUSE_OK;
    if (!lag || !mv)
        return fail(MUSE_ERR_INVALID, "NULL output");
    HIP_TRY(hipMemcpyAsync(lag, b->lag, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, b->stream()));
    HIP_TRY(hipMemcpyAsync(mv, b->mv, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, b->stream()));
    HIP_TRY(hipStreamSynchronize(b->stream()));
    return MUSE_OK;
}

// the kernel automatic selection takes for this batch's all-scores pass (bench.py names it in its roofline object)
// the kernel the all-scores pass takes for this batch (bench.py names it in its roofline object), from the same choice the launch
// makes -- test hooks, missing tables and the learned hand-off included
extern "C" int muse_batch_kernel_name(muse_batch *b, char *name, int32_t cap)
{
    if (!b || !name || cap < 1)
        return fail(MUSE_ERR_INVALID, "NULL argument");
    // (the names rocprofv3 prints for the instantiations: profiles/r*_counters.json is keyed by them)
    char k[96] = "xcorr_fused_generic";
    const char *padded = b->N < b->n ? "true" : "false", *f32 = b->g->f32 ? "true" : "false";
    if (b->n > GENERIC_MAX_N) { // (the pass is a sequence of kernels: the one that moves the most bytes)
        snprintf(k, sizeof(k), "huge_rows<false>");
    } else {
        const KernelChoice kc = choose_kernel(b, (b->g->M + 1) / 2);
        switch (kc.error ? -1 : kc.variant) {
        case KERNEL_R16_FOLD: snprintf(k, sizeof(k), "xcorr_fused_n4096_fold<false, %s, %s>", padded, f32); break;
        case KERNEL_R16_OCC3: snprintf(k, sizeof(k), "xcorr_fused_n4096_occ4"); break;
        case KERNEL_SMALL: snprintf(k, sizeof(k), "xcorr_fused_small<%d, %s, false, %s>", b->logn, padded, f32); break;
        case KERNEL_LONG: snprintf(k, sizeof(k), "xcorr_fused_long<%d, %s, false>", b->logn, padded); break;
        case KERNEL_STOCKHAM: snprintf(k, sizeof(k), b->n >= 8192 ? "xcorr_fused_stk_4step" : "xcorr_fused_stockham"); break;
        case KERNEL_REAL_SPLIT: snprintf(k, sizeof(k), "xcorr_fused_real32k_split<%s>", padded); break;
        case KERNEL_REAL: snprintf(k, sizeof(k), "xcorr_fused_real%dk<%s>", b->n / 1024, padded); break;
        default: break;
        }
    }
    snprintf(name, (size_t)cap, "%s", k);
    return MUSE_OK;
}

extern "C" int muse_batch_free(muse_batch *b)
{
    if (!b)
        return MUSE_OK;
    muse_ctx *const c = b->ctx;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(b->stream());
    if (b->sp && b->sp->refs.fetch_sub(1) == 1) {
        dfree(c, b->sp->X);
        dfree(c, b->sp->xc);
        dfree(c, b->sp->xcp);
        dfree(c, b->sp->xcw);
        dfree(c, b->sp->xcf);
        dfree(c, b->sp->xs);
        dfree(c, b->sp->c1);
        delete b->sp;
    }
    if (b->small_out) { // kept by the context for its next batch (coherent pinned memory is slow to allocate)
        std::lock_guard<std::mutex> lock(c->small_mu);
        c->small_free.emplace_back(b->small_out, b->small_cap);
    }
    dfree(c, b->ovf_count);
    if (b->handoff_host)
        hfree(c, b->handoff_host);
    dfree(c, b->ovf_list);
    dfree(c, b->mv);
    dfree(c, b->lag);
    dfree(c, b->gid_dev);
    dfree(c, b->gw.key);
    dfree(c, b->gw.first);
    dfree(c, b->gw.win);
    dfree(c, b->rec);
    dfree(c, b->selkey);
    dfree(c, b->cand);
    if (b->cand_host)
        hfree(c, b->cand_host);
    if (b->cnt_host)
        hfree(c, b->cnt_host);
    if (b->rec_host)
        hfree(c, b->rec_host);
    if (b->key_host)
        hfree(c, b->key_host);
    dfree(c, b->cnt);
    dfree(c, b->scr_flags);
    dfree(c, b->scr_var);
    dfree(c, b->include);
    dfree(c, b->scr_keys);
    dfree(c, b->scr_gmay);
    dfree(c, b->scr_gkplus);
    dfree(c, b->scr_gcert);
    if (b->refine_host)
        hfree(c, b->refine_host);
    if (b->err_host)
        hfree(c, b->err_host);
    dfree(c, b->err_dev);
    dfree(c, b->est_save);
    muse_group *g = b->g;
    muse_ctx *ctx = b->ctx;
    delete b;
    group_release(g);
    ctx_release(ctx);
    return MUSE_OK;
}
