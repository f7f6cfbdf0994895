// capi_xcorr.hip -- single-pair entry points (xcorr_test.go-style access) and the batched two-sided xCorr (xcorr.go:102-153)
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
#include "capi_internal.h"
#include "xcorr_huge.h"

using namespace muse;


// ------------------------------------------------- single-pair entry points
static bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

// shared tail: x (len lenx) vs y (len leny) at FFT length n
static int single_pair(muse_ctx *ctx, const double *x, int lenx, const double *y, int leny, int n, int normalize_x,
                       int normalize_y, double x_scale, double cc_scale, double *cc, int32_t *lag, double *mv,
                       int32_t *is_nil)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!x || !y || lenx < 1 || leny < 1 || n < lenx || n < leny || !lag || !mv)
        return fail(MUSE_ERR_INVALID, "bad single-pair arguments");
    if ((normalize_x && lenx < 2) || (normalize_y && leny < 2))
        return fail(MUSE_ERR_INVALID, "series length 1 has no sample standard deviation");
    if (n > HUGE_MAX_N || (!is_pow2(n) && n > HUGE_MAX_N / 2))
        return fail(MUSE_ERR_UNSUPPORTED, "FFT length %d is not built (powers of two up to %d, any other n up to %d)", n,
                    HUGE_MAX_N, HUGE_MAX_N / 2);
    if (!is_pow2(n) && n > 8192) {
        // Any n (xCorr takes the n it is given, xcorr.go:104-106; gonum transforms any length): the circular correlation of the two
        // sequences padded to n is folded out of the one at a power of two L >= 2 n, where nothing wraps around --
        //   cc_L[k] = r[k] (0 <= k < n),  cc_L[L - m] = r[-m] (0 < m < n)   (r = the linear correlation),   cc_n[k] = r[k] + r[k - n]
        // -- on the batched kernels of that length; zeroPad(x, L) = zeroPad(zeroPad(x, n), L), the statistics are the series' own,
        // and the scale keeps its n: cc_scale is the reference's factor BEHIND an unnormalised inverse of length n, so at length L
        // it becomes cc_scale n / L.  The fold and maxAbsIndex (xcorr.go:39-50) run on the host: a single-pair entry point.
        int64_t L = 1;
        while (L < 2 * (int64_t)n)
            L <<= 1;
        std::vector<double> ccL((size_t)L);
        int32_t lagL = 0, nilL = 0;
        double mvL = 0.0;
        rc = single_pair(ctx, x, lenx, y, leny, (int)L, normalize_x, normalize_y, x_scale, cc_scale * (double)n / (double)L, ccL.data(),
                         &lagL, &mvL, &nilL);
        if (rc)
            return rc;
        if (is_nil)
            *is_nil = nilL;
        if (nilL) { // (nil, 0, 0)
            *lag = 0;
            *mv = 0.0;
            return MUSE_OK;
        }
        int64_t mi = 0;
        double best = 0.0, first = 0.0;
        for (int64_t k = 0; k < n; k++) {
            const double v = k == 0 ? ccL[0] : ccL[(size_t)k] + ccL[(size_t)(L - n + k)];
            if (cc)
                cc[k] = v;
            if (k == 0)
                first = v;
            if (std::fabs(v) > std::fabs(best)) { // strictly greater: the first index keeps ties; NaN never wins
                best = v;
                mi = k;
            }
        }
        *mv = mi == 0 ? first : best; // (nothing above 0, or every cc NaN: index 0, mv = cc[0])
        *lag = (int32_t)(mi > n / 2 ? mi - n : mi);
        return MUSE_OK;
    }
    double *dx = nullptr, *dy = nullptr, *dcc = nullptr, *dmv = nullptr;
    int *dlag = nullptr, *dstat = nullptr;
    double2 *dX = nullptr, *dxc = nullptr, *dscr = nullptr;
    int nil = 0, lg = 0;
    double val = 0.0;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() {
        (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dcc); (void)hipFree(dmv);
        (void)hipFree(dlag); (void)hipFree(dstat); (void)hipFree(dX); (void)hipFree(dxc); (void)hipFree(dscr);
    };
#define SP_TRY(expr)                                                                                        \
    do {                                                                                                    \
        e = (expr);                                                                                         \
        if (e != hipSuccess) {                                                                              \
            cleanup();                                                                                      \
            return fail(MUSE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e));                        \
        }                                                                                                   \
    } while (0)
    SP_TRY(hipMalloc(&dy, (size_t)leny * sizeof(double)));
    SP_TRY(hipMalloc(&dcc, (size_t)n * sizeof(double)));
    SP_TRY(hipMalloc(&dmv, sizeof(double)));
    SP_TRY(hipMalloc(&dlag, sizeof(int)));
    SP_TRY(hipMalloc(&dstat, sizeof(int)));
    SP_TRY(hipMemcpyAsync(dy, y, (size_t)leny * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (n > GENERIC_MAX_N) { // series longer than 65 536 samples: the pairwise form of xcorr_huge.hip, one pair
        SP_TRY(hipMalloc(&dx, (size_t)lenx * sizeof(double)));
        SP_TRY(hipMemcpyAsync(dx, x, (size_t)lenx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        rc = huge_pairs(ctx, dx, lenx, lenx, normalize_x, x_scale, dy, leny, leny, normalize_y, 1, n, cc_scale, dmv, dlag, dstat, dcc);
        if (rc) {
            (void)hipStreamSynchronize(ctx->stream);
            cleanup();
            return rc;
        }
        SP_TRY(hipMemcpyAsync(&lg, dlag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipMemcpyAsync(&val, dmv, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipMemcpyAsync(&nil, dstat, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipStreamSynchronize(ctx->stream));
    } else if (is_pow2(n) && n >= 2) {
        // FFT path: the same device code the batch uses (generic kernel)
        SP_TRY(hipMalloc(&dX, (size_t)(n / 2 + 1) * sizeof(double2)));
        SP_TRY(hipMalloc(&dxc, (size_t)n * sizeof(double2)));
        int zero = 0;
        rc = build_spectrum(ctx, x, lenx, n, normalize_x, x_scale, cc_scale, dX, dxc, nullptr, nullptr, &zero);
        if (rc) {
            cleanup();
            return rc;
        }
        FusedParams p{};
        p.rows = dy;
        p.M = 1;
        p.stride = leny;
        p.npairs = 1;
        p.N = leny;
        p.n = n;
        p.logn = ilog2(n);
        p.normalize_y = normalize_y;
        p.xc = dxc;
        p.tw1 = ctx->tw1;
        p.tw2 = ctx->tw2;
        p.twm = ctx->twm;
        p.mv = dmv;
        p.lag = dlag;
        p.cc_out = dcc;
        p.nil_out = dstat;
        if (n > GENERIC_LDS_MAX_N) {
            SP_TRY(hipMalloc(&dscr, (size_t)n * sizeof(double2)));
            p.gscratch = dscr;
            p.gscratch_slices = 1;
        }
        SP_TRY(launch_fused(p, KERNEL_GENERIC, ctx->num_cus, ctx->stream));
        SP_TRY(hipMemcpyAsync(&lg, dlag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipMemcpyAsync(&val, dmv, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipMemcpyAsync(&nil, dstat, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipStreamSynchronize(ctx->stream));
        nil = (nil || zero) ? 1 : 0; // sigma(y) == 0 or sigma(x) == 0 -> (nil, 0, 0)
    } else {
        SP_TRY(hipMalloc(&dx, (size_t)lenx * sizeof(double)));
        SP_TRY(hipMemcpyAsync(dx, x, (size_t)lenx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SP_TRY(launch_direct(dx, lenx, dy, leny, n, normalize_x, normalize_y, x_scale, cc_scale * (double)n, dcc,
                             dlag, dmv, dstat, ctx->stream));
        SP_TRY(hipMemcpyAsync(&lg, dlag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipMemcpyAsync(&val, dmv, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipMemcpyAsync(&nil, dstat, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (cc && !nil)
        SP_TRY(hipMemcpy(cc, dcc, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (cc && !nil && n > GENERIC_MAX_N && val != val) // (a NaN / Inf series enters the long-series transform as zeros: every cc is NaN in the reference)
        std::fill(cc, cc + n, std::numeric_limits<double>::quiet_NaN());
#undef SP_TRY
    cleanup();
    *lag = nil ? 0 : lg;
    *mv = nil ? 0.0 : val;
    if (is_nil)
        *is_nil = nil;
    return MUSE_OK;
}

extern "C" int muse_xcorr_with_x(muse_ctx *ctx, const double *ref, const double *y, int32_t N, int32_t n, double *cc,
                                 int32_t *lag, double *mv, int32_t *is_nil)
{
    if (N < 2)
        return fail(MUSE_ERR_INVALID, "N must be >= 2");
    // reference side: zNormalize(ref)/(N-1) (xcorr_test.go:259-266 == muse_batch.go:38-47);
    // sigma(ref) == 0 is the caller's "Invalid input query" error.
    int32_t nil = 0;
    // probe sigma(ref) through the same path: build with normalize and check flag
    int rc = single_pair(ctx, ref, N, y, N, n, 1, 1, 1.0 / (double)(N - 1), 1.0 / (double)n, cc, lag, mv, &nil);
    if (rc)
        return rc;
    if (is_nil)
        *is_nil = nil;
    return MUSE_OK;
}

extern "C" int muse_xcorr(muse_ctx *ctx, const double *x, int32_t lenx, const double *y, int32_t leny, int32_t n,
                          int32_t normalize, double *cc, int32_t *lag, double *mv, int32_t *is_nil)
{
    const int32_t minn = std::max(lenx, leny); // xcorr.go:104-106
    if (n < minn)
        n = minn;
    // xcorr.go:139-143: 1/(n(n-1)) when normalized, else 1/n
    const double cc_scale = normalize ? 1.0 / ((double)n * (double)(n - 1)) : 1.0 / (double)n;
    return single_pair(ctx, x, lenx, y, leny, n, normalize, normalize, 1.0, cc_scale, cc, lag, mv, is_nil);
}

// ---- batched two-sided xCorr (xcorr.go:102-153)
static int xcorr_groups_impl(muse_group *gx, muse_group *gy, int32_t n, int32_t normalize, int32_t *lag, double *mv,
                             int32_t *is_nil, double *cc, bool rescue);

// Pairs that came back NaN although the reference may still return numbers: FINITE samples whose squares leave the float64
// range (|x| >~ 1e154; the batched kernels take sum d^2 for their power-of-two scales).  A device kernel looks at each such
// pair again (xcorr_kernels.hip, two_sided_rescue_kernel): NaN stands (a NaN / Inf sample, or the reference's own arithmetic
// overflows), every cc is zero (normalized with sigma = +Inf; raw with an all-zero series), or the pair is recomputed by the same
// batched kernels on copies scaled by exact powers of two that leave the result unchanged.
static int rescue_overflowed_pairs(muse_group *gx, muse_group *gy, int32_t n, int32_t normalize, int32_t *lag, double *mv,
                                   const std::vector<int> &nil, double *cc)
{
    muse_ctx *ctx = gx->ctx;
    const int64_t M = gx->M;
    std::vector<long long> list;
    for (int64_t i = 0; i < M; i++)
        if (!nil[(size_t)i] && mv[i] != mv[i])
            list.push_back(i);
    if (list.empty())
        return MUSE_OK;
    const int K = (int)std::min<size_t>(list.size(), 0x7fffffff);
    long long *dlist = nullptr;
    int *dcode = nullptr;
    double2 *dscale = nullptr;
    muse_group *sx = nullptr, *sy = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(dlist); (void)hipFree(dcode); (void)hipFree(dscale);
        muse_group_free(sx);
        muse_group_free(sy);
    };
    std::vector<int> code((size_t)K);
    hipError_t e = hipMalloc(&dlist, (size_t)K * sizeof(long long));
    if (e == hipSuccess)
        e = hipMalloc(&dcode, (size_t)K * sizeof(int));
    if (e == hipSuccess)
        e = hipMalloc(&dscale, (size_t)K * sizeof(double2));
    if (e == hipSuccess)
        e = hipMemcpyAsync(dlist, list.data(), (size_t)K * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = launch_two_sided_rescue(gx->rows, gx->stride, gx->N, gy->rows, gy->stride, gy->N, n, normalize, dlist, K, dcode, dscale,
                                    ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(code.data(), dcode, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        return fail(MUSE_ERR_HIP, "two-sided xCorr (overflowed statistics): %s", hipGetErrorString(e));
    }
    std::vector<long long> again; // positions in `list` of the pairs to recompute
    for (int k = 0; k < K; k++) {
        const int64_t i = list[(size_t)k];
        if (code[(size_t)k] == 1) { // every cc is zero: maxAbsIndex keeps index 0
            mv[i] = 0.0;
            lag[i] = 0;
            if (cc)
                std::fill(cc + (size_t)i * (size_t)n, cc + (size_t)(i + 1) * (size_t)n, 0.0);
        } else if (code[(size_t)k] == 2) {
            again.push_back(k);
        }
    }
    if (again.empty()) {
        cleanup();
        return MUSE_OK;
    }
    // the pairs to recompute, compacted: list entries and scales in the order of `again`
    const int K2 = (int)again.size();
    std::vector<long long> list2((size_t)K2);
    std::vector<double2> scale((size_t)K), scale2((size_t)K2);
    e = hipMemcpy(scale.data(), dscale, (size_t)K * sizeof(double2), hipMemcpyDeviceToHost);
    for (int k = 0; k < K2; k++) {
        list2[(size_t)k] = list[(size_t)again[(size_t)k]];
        scale2[(size_t)k] = scale[(size_t)again[(size_t)k]];
    }
    int rc = MUSE_OK;
    if (e == hipSuccess)
        e = hipMemcpyAsync(dlist, list2.data(), (size_t)K2 * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(dscale, scale2.data(), (size_t)K2 * sizeof(double2), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        rc = muse_group_create(ctx, K2, gx->N, &sx);
    if (e == hipSuccess && !rc)
        rc = muse_group_create(ctx, K2, gy->N, &sy);
    if (e == hipSuccess && !rc)
        e = launch_scale_listed_rows(gx->rows, gx->stride, gx->N, dlist, dscale, 0, K2, sx->rows, ctx->stream);
    if (e == hipSuccess && !rc)
        e = launch_scale_listed_rows(gy->rows, gy->stride, gy->N, dlist, dscale, 1, K2, sy->rows, ctx->stream);
    if (e != hipSuccess || rc) {
        cleanup();
        return rc ? rc : fail(MUSE_ERR_HIP, "two-sided xCorr (overflowed statistics): %s", hipGetErrorString(e));
    }
    sx->M = K2;
    sy->M = K2;
    std::vector<int32_t> lag2((size_t)K2), nil2((size_t)K2);
    std::vector<double> mv2((size_t)K2), cc2(cc ? (size_t)K2 * (size_t)n : 0);
    rc = xcorr_groups_impl(sx, sy, n, normalize, lag2.data(), mv2.data(), nil2.data(), cc ? cc2.data() : nullptr, false);
    if (!rc)
        for (int k = 0; k < K2; k++) {
            const int64_t i = list2[(size_t)k];
            // raw samples: the pair was recomputed at magnitude ~1; the result goes back by the exact power of two the two scales
            // took out (normalised: the scales cancel in x / sigma).  A value that leaves the float64 range on the way back is
            // where the reference's own products overflow: NaN stands for the whole pair.
            const int back = normalize ? 0 : -(std::ilogb(scale2[(size_t)k].x) + std::ilogb(scale2[(size_t)k].y));
            double v = std::ldexp(mv2[(size_t)k], back);
            bool over = std::isinf(v) && !std::isinf(mv2[(size_t)k]);
            double *dst = cc ? cc + (size_t)i * (size_t)n : nullptr;
            if (cc) {
                const double *src = cc2.data() + (size_t)k * (size_t)n;
                for (int32_t q = 0; q < n; q++) {
                    dst[q] = std::ldexp(src[q], back);
                    over = over || (std::isinf(dst[q]) && !std::isinf(src[q]));
                }
            }
            if (over) {
                lag[i] = 0;
                mv[i] = std::numeric_limits<double>::quiet_NaN();
                if (cc)
                    std::fill(dst, dst + n, std::numeric_limits<double>::quiet_NaN());
                continue;
            }
            lag[i] = lag2[(size_t)k];
            mv[i] = v;
        }
    cleanup();
    return rc;
}

extern "C" int muse_xcorr_groups(muse_group *gx, muse_group *gy, int32_t n, int32_t normalize, int32_t *lag, double *mv,
                                 int32_t *is_nil, double *cc)
{
    return xcorr_groups_impl(gx, gy, n, normalize, lag, mv, is_nil, cc, true);
}

static int xcorr_groups_impl(muse_group *gx, muse_group *gy, int32_t n, int32_t normalize, int32_t *lag, double *mv,
                             int32_t *is_nil, double *cc, bool rescue)
{
    if (!gx || !gy || gx->ctx != gy->ctx)
        return fail(MUSE_ERR_INVALID, "the two groups must share a context");
    if (gx->f32 || gy->f32)
        return fail(MUSE_ERR_UNSUPPORTED, "xCorr is not built for float32-storage groups");
    muse_ctx *ctx = gx->ctx;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (gx->M != gy->M)
        return fail(MUSE_ERR_LENGTH, "xCorr pairs row i of x with row i of y: %lld vs %lld rows", (long long)gx->M, (long long)gy->M);
    const int64_t M = gx->M;
    if (M == 0)
        return MUSE_OK;
    if (!lag || !mv)
        return fail(MUSE_ERR_INVALID, "NULL output");
    const int32_t Nx = gx->N, Ny = gy->N;
    n = std::max(n, std::max(Nx, Ny)); // xcorr.go:104-106
    if (normalize && (Nx < 2 || Ny < 2))
        return fail(MUSE_ERR_INVALID, "series length 1 has no sample standard deviation");
    rc = group_ready(gx);
    if (!rc)
        rc = group_ready(gy);
    if (rc)
        return rc;
    if (n > HUGE_MAX_N)
        return fail(MUSE_ERR_UNSUPPORTED, "FFT length %d > %d is not built", n, HUGE_MAX_N);
    const bool huge = is_pow2(n) && n > GENERIC_MAX_N; // series longer than 65 536 samples (xcorr_huge.hip)
    if (!is_pow2(n) || n < 512) {
        // FFT lengths without a batched kernel (the reference's n = 5 tables, short series): pair by pair through the
        // single-pair path (generic radix-2 kernel, or the direct kernel for n that is not a power of two)
        std::vector<double> x((size_t)Nx), y((size_t)Ny);
        HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i < M; i++) {
            HIP_TRY(hipMemcpy(x.data(), gx->rows + i * gx->stride, (size_t)Nx * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(y.data(), gy->rows + i * gy->stride, (size_t)Ny * sizeof(double), hipMemcpyDeviceToHost));
            int32_t nil = 0;
            rc = muse_xcorr(ctx, x.data(), Nx, y.data(), Ny, n, normalize, cc ? cc + (size_t)i * (size_t)n : nullptr, lag + i, mv + i, &nil);
            if (rc)
                return rc;
            if (cc && nil)
                std::fill(cc + (size_t)i * (size_t)n, cc + (size_t)(i + 1) * (size_t)n, 0.0);
            if (is_nil)
                is_nil[i] = nil;
        }
        return MUSE_OK;
    }
    hipError_t e = huge ? hipSuccess : ensure_gscratch(ctx, n);
    if (e == hipSuccess && n >= 32768 && !huge)
        e = ensure_twl(ctx, n);
    if (e != hipSuccess)
        return fail(MUSE_ERR_NOMEM, "scratch: %s", hipGetErrorString(e));
    double *dmv = nullptr, *dcc = nullptr;
    int *dlag = nullptr, *dnil = nullptr, *dcount = nullptr;
    long long *dlist = nullptr; // n >= 32768, no padding: the pairs the single-read launch hands to the launch that scales first
    auto cleanup = [&]() { (void)hipFree(dmv); (void)hipFree(dcc); (void)hipFree(dlag); (void)hipFree(dnil); (void)hipFree(dlist); (void)hipFree(dcount); };
    const bool listing = n == 65536 && Nx == n && Ny == n;
    e = hipMalloc(&dmv, (size_t)M * sizeof(double));
    if (e == hipSuccess && listing)
        e = hipMalloc(&dlist, (size_t)M * sizeof(long long));
    if (e == hipSuccess && listing)
        e = hipMalloc(&dcount, 2 * sizeof(int));
    if (e == hipSuccess && listing)
        e = hipMemsetAsync(dcount, 0, 2 * sizeof(int), ctx->stream);
    if (e == hipSuccess)
        e = hipMalloc(&dlag, (size_t)M * sizeof(int));
    if (e == hipSuccess)
        e = hipMalloc(&dnil, (size_t)M * sizeof(int));
    if (e == hipSuccess && cc)
        e = hipMalloc(&dcc, (size_t)M * (size_t)n * sizeof(double));
    if (e == hipSuccess && cc)
        e = hipMemsetAsync(dcc, 0, (size_t)M * (size_t)n * sizeof(double), ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        return fail(MUSE_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    if (huge) {
        rc = huge_pairs(ctx, gx->rows, gx->stride, Nx, normalize ? 1 : 0, 1.0, gy->rows, gy->stride, Ny, normalize ? 1 : 0, M, n,
                        normalize ? 1.0 / ((double)n * (double)(n - 1)) : 1.0 / (double)n, dmv, dlag, dnil, dcc);
        if (rc) {
            (void)hipStreamSynchronize(ctx->stream);
            cleanup();
            return rc;
        }
    } else {
        // long series work in the context's scratch buffer: its pointer must not be swapped between reading it and the launch
        std::unique_lock<std::mutex> scratch_lock(ctx->stage_mu, std::defer_lock);
        if (n >= GENERIC_LDS_MAX_N)
            scratch_lock.lock();
        FusedParams p{};
        p.rows = gy->rows;
        p.stride = gy->stride;
        p.N = Ny;
        p.xrows = gx->rows;
        p.xstride = gx->stride;
        p.Nx = Nx;
        p.M = M;
        p.npairs = M;
        p.n = n;
        p.logn = ilog2(n);
        p.normalize_y = normalize ? 1 : 0;
        p.twm = ctx->twm;
        p.g2 = ctx->g2;
        p.g3a = ctx->g3a;
        p.g3b = ctx->g3b;
        p.gsmall = (p.logn >= 9 && p.logn <= 11) ? ctx->gsmall[p.logn - 9] : (p.logn == 13 || p.logn == 14) ? ctx->gsmall[p.logn - 10]
                   : p.logn == 15 ? ctx->gsmall[4] /* n = 32768 runs on the 16384-point transform's tables (xcorr_real.hip) */ : nullptr;
        p.gscratch = ctx->gscratch;
        p.gscratch_slices = (long long)(ctx->gscratch_elems / (size_t)n);
        p.twl = (p.logn >= 14 && p.logn <= 16) ? ctx->twl[p.logn - 14] : nullptr;
        p.mv = dmv;
        p.lag = dlag;
        p.nil_out = dnil;
        p.cc_out = dcc;
        p.ovf_list = dlist;
        p.ovf_count = dcount;
        // n = 8192, 16384: each series a real transform on the 4096- / 8192-point machinery (xcorr_real.hip); test hook 12 keeps the
        // pair-packed kernels (xcorr_two_sided_small<13 / 14>)
        const bool real_form = ctx->variant != 12 && (p.logn == 13 || (p.logn == 14 && ctx->gsmall[3]));
        if (real_form && p.logn == 14)
            p.gsmall = ctx->gsmall[3]; // (n = 8192 runs on the n = 4096 kernel's tables: p.g2, p.g3a, p.g3b)
        // (the measurement hook repeats the launch back to back: same results; not where the launch appends to a pair list)
        for (int rep = 0; rep < (listing ? 1 : std::max(1, ctx->xcorr_repeat)) && e == hipSuccess; rep++) {
            LaunchTimer timer(ctx);
            e = timer.begin();
            if (e == hipSuccess)
                e = real_form ? launch_two_sided_real(p, ctx->num_cus, ctx->stream) : launch_two_sided(p, ctx->num_cus, ctx->stream);
            if (e == hipSuccess)
                e = timer.end();
        }
    }
    std::vector<int> nil((size_t)M);
    if (e == hipSuccess)
        e = hipMemcpyAsync(lag, dlag, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(mv, dmv, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(nil.data(), dnil, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && cc)
        e = hipMemcpyAsync(cc, dcc, (size_t)M * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess)
        return fail(MUSE_ERR_HIP, "two-sided xCorr: %s", hipGetErrorString(e));
    for (int64_t i = 0; i < M; i++) {
        if (is_nil)
            is_nil[i] = nil[(size_t)i];
        if (cc && !nil[(size_t)i] && mv[i] != mv[i]) // NaN / Inf statistics: the reference's cc is NaN throughout
            std::fill(cc + (size_t)i * (size_t)n, cc + (size_t)(i + 1) * (size_t)n, std::numeric_limits<double>::quiet_NaN());
    }
    // ... unless the samples are finite and only their squares overflowed
    return rescue ? rescue_overflowed_pairs(gx, gy, n, normalize, lag, mv, nil, cc) : MUSE_OK;
}

extern "C" int muse_xcorr_batch(muse_ctx *ctx, const double *x_rows, const double *y_rows, int64_t M, int32_t lenx, int32_t leny,
                                int32_t n, int32_t normalize, int32_t *lag, double *mv, int32_t *is_nil, double *cc)
{
    if (!ctx || M < 0 || lenx < 1 || leny < 1 || (M > 0 && (!x_rows || !y_rows)))
        return fail(MUSE_ERR_INVALID, "bad xCorr batch arguments");
    if (M == 0)
        return MUSE_OK;
    muse_group *gx = nullptr, *gy = nullptr;
    int rc = muse_group_upload(ctx, x_rows, M, lenx, lenx, &gx);
    if (!rc)
        rc = muse_group_upload(ctx, y_rows, M, leny, leny, &gy);
    if (!rc)
        rc = muse_xcorr_groups(gx, gy, n, normalize, lag, mv, is_nil, cc);
    const std::string msg = rc ? g_last_error : std::string();
    muse_group_free(gx);
    muse_group_free(gy);
    if (rc)
        g_last_error = msg;
    return rc;
}
