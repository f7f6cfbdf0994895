// capi_screen.hip -- the opt-in filter-and-refine Run (fp32 screening pass + fp64 re-evaluation): frozen
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
#include "capi_internal.h"

using namespace muse;


// ---- filter-and-refine Run (docs/HISTORY.md 4.6): ungrouped N = n = 4096 Runs under automatic kernel selection
muse_batch::RunKey run_key(const muse_batch *b, const int32_t *group_id, int64_t G, int32_t max_lag, int32_t top_n,
                                  double threshold, int32_t sign_filter, int32_t abs_scores)
{
    muse_batch::RunKey k;
    k.M = b->g->M;
    k.G = group_id ? G : 0;
    k.grouped = group_id ? 1 : 0;
    k.max_lag = max_lag;
    k.top_n = top_n;
    k.threshold = threshold;
    k.sign_filter = sign_filter;
    k.abs_scores = abs_scores ? 1 : 0;
    return k;
}

// which path a Run with these filters takes (MUSE_RUN_PATH_*); label groups are handled too (per-group bounds:
// reduce_kernels.hip, screen_g1..g4)
int32_t screen_path(const muse_batch *b, const muse_batch::RunKey &key, bool already_scored)
{
    const muse_ctx *ctx = b->ctx;
    const int64_t M = b->g->M;
    const bool length_ok = b->n >= 512 && b->n <= 65536; // every FFT length with a tuned kernel (N > n/2 by construction)
    const bool eligible = !already_scored && ctx->screening && ctx->variant == 0 && length_ok && b->xcf && !b->g->f32 && key.top_n >= 1 &&
                          key.top_n <= TOPN_DEVICE_MAX && M / 2 < 0x7fffffffLL &&
                          (ctx->screen_min_rows > 0 ? M >= ctx->screen_min_rows : M * (int64_t)b->n >= (int64_t)32768 * 4096);
    if (!eligible)
        return MUSE_RUN_PATH_FP64;
    if (b->guard_off)
        return MUSE_RUN_PATH_FP64_GUARD;
    if (b->costly_key == key)
        return MUSE_RUN_PATH_FP64_COSTLY;
    return MUSE_RUN_PATH_SCREENED;
}

// Error bound of the screening pass's estimates, in its SCALED units (docs/screen_error_bound.md derives every number;
// tests/test_abi_cpu.py re-sums the per-stage constants and compares).  The pass scales each centred series by
// scl = 2^-(e >> 1), e = exponent of its variance, so scl * sigma lies in [1, 2) and
//     score = estimate / (scl * sigma),   |score error| <= |estimate error|            (scl * sigma >= 1),
//     ||z||_2 <= sqrt(2) * 2 * sqrt(N - 1) < 2 sqrt(2 n)      (z = A + iB: 181 at n = 4096).
// Both transforms run on z and the product spectrum is bounded by max|X| * ||Z||_2 / n, so every error term of the
// standard fp32 FFT analysis (Higham, Accuracy and Stability of Numerical Algorithms, Thm 24.2: per radix-2 stage
// eta = mu + gamma_4 (sqrt 2 + mu)) scales with u * max|X| * ||z||_2.  First-order constants: 6.66 u per radix-2 stage
// (rounded butterfly constants), 15.3 u per scaling by a twiddle that is a product of <= 4 rounded factors, 3.83 u per
// scaling by a single rounded table entry (pass-2 twiddles, the spectrum table, the four-step twiddle):
//     n = 4096:  2 (12 * 6.66 + 15.3 + 3.83) + 3.83 = 202   -> 256 used
//     n = 8192:  2 (13 * 6.66 + 3 * 15.3)    + 3.83 = 269   -> 320 used
//     n = 65536: 2 (16 * 6.66 + 15.3 + 2 * 3.83) + 3.83 = 263 -> 384 used
// The second term is the rounding of the fp32 input copy (|mean d| <= 8 sigma is enforced by the kernel):
// ||delta c||_2 <= 2u (2 + 16) sqrt(N) and |delta cc| <= ||delta c||_2 ||xs||_2, ||xs||_2 = 1 / sqrt(N-1): 2.2e-6, plus
// (N < n) 1e-6 for the rounded mean acting through the indicator correlation: 3e-6 used.
static double screen_error_scaled(double xmax, int n)
{
    const double u = 5.9604644775390625e-08; // 2^-24
    const double C = n > 8192 ? 384.0 : n > 4096 ? 320.0 : 256.0;
    return C * u * (2.0 * std::sqrt(2.0 * (double)n)) * xmax + 3e-6;
}

// test hook (muse_hip_test.h): the bound for an FFT length and max|X|
extern "C" int muse_test_screen_bound(int32_t n, double xmax, double *Es)
{
    if (!Es || n < 2)
        return fail(MUSE_ERR_INVALID, "bad arguments");
    *Es = screen_error_scaled(xmax, n);
    return MUSE_OK;
}

// The filter-and-refine scoring in three steps, so that the screening pass can be one launch per batch or one launch
// for several batches (muse_batch_run_many): screen_prepare (workspace, bound, cleared flags), the pass, screen_finish
// (keys, cut, compaction, fp64 re-evaluation of the listed pairs, guard).
struct ScreenPlan {
    double Es = 0.0; // the bound in the pass's scaled units
};

static int screen_prepare(muse_batch *b, int32_t top_n, const int *gid_dev, int64_t G, ScreenPlan &plan)
{
    muse_ctx *ctx = b->ctx;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    rc = group_ready(b->g);
    if (rc)
        return rc;
    const int64_t M = b->g->M;
    rc = ensure_scores(b);
    if (rc)
        return rc;
    const int64_t npairs = (M + 1) / 2;
    if (M > b->scr_cap) {
        dfree(b->ctx, b->scr_flags);
        dfree(b->ctx, b->scr_var);
        dfree(b->ctx, b->include);
        b->scr_flags = nullptr;
        b->scr_var = nullptr;
        b->include = nullptr;
        b->scr_cap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->scr_flags, (size_t)M * sizeof(unsigned)));
        HIP_TRY(dmalloc(b->ctx, &b->scr_var, (size_t)M * sizeof(double)));
        HIP_TRY(dmalloc(b->ctx, &b->include, (size_t)M));
        b->scr_cap = M;
    }
    const int64_t nkeys = screen_select_scratch(gid_dev ? G : M, top_n);
    if (gid_dev && G > b->scr_gcap) {
        dfree(b->ctx, b->scr_gmay);
        dfree(b->ctx, b->scr_gkplus);
        dfree(b->ctx, b->scr_gcert);
        b->scr_gmay = b->scr_gkplus = nullptr;
        b->scr_gcert = nullptr;
        b->scr_gcap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->scr_gmay, (size_t)G * sizeof(unsigned long long)));
        HIP_TRY(dmalloc(b->ctx, &b->scr_gkplus, (size_t)G * sizeof(unsigned long long)));
        HIP_TRY(dmalloc(b->ctx, &b->scr_gcert, (size_t)G * sizeof(int)));
        b->scr_gcap = G;
    }
    if (nkeys > b->scr_keys_cap) {
        dfree(b->ctx, b->scr_keys);
        b->scr_keys = nullptr;
        b->scr_keys_cap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->scr_keys, (size_t)nkeys * sizeof(unsigned long long)));
        b->scr_keys_cap = nkeys;
    }
    if (!b->refine_host)
        HIP_TRY(hmalloc(b->ctx, &b->refine_host, sizeof(int)));
    if (!b->err_host)
        HIP_TRY(hmalloc(b->ctx, &b->err_host, sizeof(unsigned long long)));
    if (!b->err_dev)
        HIP_TRY(dmalloc(b->ctx, &b->err_dev, sizeof(unsigned long long)));
    if (4 * npairs > b->est_cap) { // two estimates per listed pair; the list holds the selection's pairs plus the guard sample
        dfree(b->ctx, b->est_save);
        b->est_save = nullptr;
        b->est_cap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->est_save, (size_t)(4 * npairs) * sizeof(double)));
        b->est_cap = 4 * npairs;
    }
    // the list takes the selection's pairs (at most npairs) plus the guard sample (about npairs / 1024, not de-duplicated
    // against the selection): 2 npairs entries, the same capacity the fp64 pass's hand-off list has
    if (2 * npairs > b->ovf_cap) {
        dfree(b->ctx, b->ovf_list);
        b->ovf_list = nullptr;
        b->ovf_cap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->ovf_list, (size_t)(2 * npairs) * sizeof(long long)));
        b->ovf_cap = 2 * npairs;
    }
    if (b->sp->xmax < 0.0) { // once per reference: max |X[f]| (X holds the non-redundant half of a real signal's spectrum)
        std::vector<double2> X((size_t)(b->n / 2 + 1));
        HIP_TRY(hipMemcpyAsync(X.data(), b->X, X.size() * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        double m = 0.0;
        for (const double2 &x : X)
            m = std::max(m, std::hypot(x.x, x.y));
        b->sp->xmax = m;
    }
    plan.Es = screen_error_scaled(b->sp->xmax, b->n) * ctx->screen_e_scale;
    HIP_TRY(hipMemsetAsync(b->err_dev, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(hipMemsetAsync(b->scr_flags, 0, (size_t)M * sizeof(unsigned), ctx->stream));
    HIP_TRY(hipMemsetAsync(b->include, 0, (size_t)M, ctx->stream));
    HIP_TRY(hipMemsetAsync(b->ovf_count, 0, 2 * sizeof(int), ctx->stream));
    return MUSE_OK;
}

static FusedParams screen_pass_params(muse_batch *b, int32_t max_lag, const ScreenPlan &plan, bool need_sign = true)
{
    FusedParams p = base_params(b);
    p.scr_need_sign = need_sign ? 1 : 0;
    p.scr_flags = b->scr_flags;
    p.scr_var = b->scr_var;
    p.scr_max_lag = max_lag;
    p.screen_delta = 2.0 * plan.Es; // every lag whose fp32 |cc| is within 2 E of the fp32 maximum may be the exact argmax
    return p;
}

static int screen_finish(muse_batch *b, int32_t top_n, double threshold, int32_t sign_filter, int32_t abs_scores,
                         const int *gid_dev, int64_t G, const ScreenPlan &plan)
{
    muse_ctx *ctx = b->ctx;
    const int64_t M = b->g->M;
    const int64_t npairs = (M + 1) / 2;
    const double Es = plan.Es;
    ScreenSelect q{};
    q.mv = b->mv;
    q.var = b->scr_var;
    q.flags = b->scr_flags;
    q.M = M;
    q.threshold = threshold;
    q.sign_filter = sign_filter;
    q.abs_scores = abs_scores ? 1 : 0;
    q.E = Es; // score = estimate / (scl sigma) with scl sigma in [1, 2): the score's error is at most the estimate's
    q.group_id = gid_dev;
    q.G = (int)G;
    // (the group scratch borrows the final reduction's arrays: that reduction re-initialises them afterwards)
    const ScreenGroupWork sgw{b->gw.first, b->gw.key, b->scr_gmay, b->scr_gkplus, b->scr_gcert};
    HIP_TRY(launch_screen_select(q, top_n, b->selkey, b->scr_keys, sgw, b->ovf_list, b->ovf_count, b->include,
                                 ctx->stream));
    // guard sample (one pair in 1024, a different set every Run): re-evaluated like the listed pairs, so the check of the
    // bound below is not confined to rows the selection wanted anyway
    HIP_TRY(launch_screen_sample(npairs, M, 0x6d757365ull + 0x9E3779B97F4A7C15ull * (unsigned long long)(++b->guard_salt),
                                 b->ovf_list, b->ovf_count, b->include, ctx->stream));
    // the fp64 kernel re-evaluates the listed pairs (count stays on the device and bounds its loop)
    FusedParams r = base_params(b);
    r.pair_list = b->ovf_list;
    r.pair_count = b->ovf_count;
    r.npairs = std::min<long long>(npairs, (long long)ctx->num_cus * 3);
    HIP_TRY(launch_screen_save(q, b->ovf_list, b->ovf_count, b->est_save, ctx->stream));
    HIP_TRY(launch_fused(r, b->n == 4096 ? KERNEL_R16_OCC3 : (b->n <= 2048 || b->n == 8192 || b->n == 16384) ? KERNEL_SMALL : KERNEL_STOCKHAM, ctx->num_cus, ctx->stream));
    // guard: the re-evaluated rows have an estimate and an fp64 score; the largest difference must respect the bound
    HIP_TRY(launch_screen_check(b->mv, M, b->ovf_list, b->ovf_count, b->est_save, b->err_dev, ctx->stream));
    *b->refine_host = 0;
    *b->err_host = 0ull;
    b->last_E = q.E;
    HIP_TRY(hipMemcpyAsync(b->refine_host, b->ovf_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(b->err_host, b->err_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    b->scores_exact = false;
    return MUSE_OK;
}

int score_screened(muse_batch *b, int32_t max_lag, int32_t top_n, double threshold, int32_t sign_filter,
                          int32_t abs_scores, const int *gid_dev, int64_t G)
{
    muse_ctx *ctx = b->ctx;
    ScreenPlan plan;
    int rc = screen_prepare(b, top_n, gid_dev, G, plan);
    if (rc)
        return rc;
    // long series work in the context's scratch buffer: its pointer must not be swapped (a concurrent
    // muse_batch_create growing it) between reading it and enqueueing the launches (as in muse_batch_score)
    std::unique_lock<std::mutex> scratch_lock(ctx->stage_mu, std::defer_lock);
    if (b->n >= GENERIC_LDS_MAX_N)
        scratch_lock.lock();
    // (Batch.Run filters the sign of |score|, Muse.Run that of the signed score: only the latter needs the pass's sign flags)
    const FusedParams p = screen_pass_params(b, max_lag, plan, sign_filter != 0 && !abs_scores);
    LaunchTimer timer(ctx);
    HIP_TRY(timer.begin());
    HIP_TRY(b->n == 4096 ? launch_screen_pass(p, ctx->num_cus, ctx->stream) : launch_screen_pass_stk(p, ctx->num_cus, ctx->stream));
    HIP_TRY(timer.end());
    return screen_finish(b, top_n, threshold, sign_filter, abs_scores, gid_dev, G, plan);
}

extern "C" int muse_batch_last_run_info(muse_batch *b, int32_t *screened, int64_t *refined_pairs)
{
    if (!b)
        return fail(MUSE_ERR_INVALID, "NULL batch");
    if (screened)
        *screened = b->last_screened ? 1 : 0;
    if (refined_pairs)
        *refined_pairs = (b->last_screened && b->refine_host) ? (int64_t)*b->refine_host : 0;
    return MUSE_OK;
}

extern "C" int muse_batch_last_run_path(muse_batch *b, int32_t *path)
{
    if (!b || !path)
        return fail(MUSE_ERR_INVALID, "NULL argument");
    *path = b->last_path;
    return MUSE_OK;
}

// test hook (muse_hip_test.h): scales the error bound the filter-and-refine Run assumes, to exercise its guard
extern "C" int muse_test_set_screen_bound_scale(muse_ctx *ctx, double scale)
{
    if (!ctx || !(scale > 0.0))
        return fail(MUSE_ERR_INVALID, "bad bound scale");
    ctx->screen_e_scale = scale;
    return MUSE_OK;
}

// test / measurement hook: the screening pass alone (estimates, SCR_* flags and the bound E in score units)
extern "C" int muse_batch_screen_estimates(muse_batch *b, int32_t max_lag, double *estimate, uint32_t *flags, double *E)
{
    if (!b)
        return fail(MUSE_ERR_INVALID, "NULL batch");
    if (b->n < 512 || b->n > 65536 || !b->xcf || b->g->f32)
        return fail(MUSE_ERR_UNSUPPORTED, "the screening pass is built for float64 groups of series of length 257 .. 65536");
    muse_ctx *ctx = b->ctx;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    const int64_t M = b->g->M;
    if (M == 0)
        return MUSE_OK;
    rc = ensure_select_ws(b, M, M, false, 1, false);
    if (rc)
        return rc;
    rc = score_screened(b, max_lag, 1, 0.0, 0, 1);
    if (rc)
        return rc;
    // (score_screened also ran the selection and the fp64 pass over the rows it picked: fetch the estimates of
    // the rows it did NOT re-evaluate, and mark the others)
    std::vector<unsigned char> inc((size_t)M);
    HIP_TRY(hipMemcpyAsync(inc.data(), b->include, (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
    if (estimate)
        HIP_TRY(hipMemcpyAsync(estimate, b->mv, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (flags)
        HIP_TRY(hipMemcpyAsync(flags, b->scr_flags, (size_t)M * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (estimate) { // rows the fp64 kernel did not touch hold the scaled fp32 value: divide by sigma
        std::vector<double> var((size_t)M);
        std::vector<unsigned> fl((size_t)M);
        HIP_TRY(hipMemcpy(var.data(), b->scr_var, (size_t)M * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(fl.data(), b->scr_flags, (size_t)M * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < M; i++)
            if (!inc[(size_t)i] && !(fl[(size_t)i] & (SCR_NAN | SCR_REFINE)))
                estimate[i] = var[(size_t)i] > 0.0 ? estimate[i] * (1.0 / std::sqrt(var[(size_t)i])) : 0.0;
    }
    if (flags)
        for (int64_t i = 0; i < M; i++)
            if (inc[(size_t)i])
                flags[i] |= 0x80000000u; // re-evaluated: `estimate` holds the fp64 result for this row
    if (E)
        *E = screen_error_scaled(b->sp->xmax, b->n);
    return MUSE_OK;
}

// after the synchronisation of a screened Run: did any re-evaluated row's estimate miss its fp64 score by more than the
// bound the selection assumed?  (Never observed -- the bound is ~3 600x the measured error -- but if it happens the bound
// cannot be trusted for the rows that were NOT re-evaluated either: the batch leaves the filter-and-refine path.)
bool screen_guard_tripped(muse_batch *b)
{
    double err;
    static_assert(sizeof(err) == sizeof(*b->err_host), "bit copy");
    memcpy(&err, b->err_host, sizeof(err));
    if (!(err > b->last_E))
        return false;
    b->guard_off = true;
    b->guard_trips++;
    return true;
}

// The filter-and-refine Run for R references over one group (muse_batch_run_many): ONE screening pass reads, reduces and
// forward-transforms every pair of series once and reports into each reference's arrays; keys, cut, compaction, fp64
// re-evaluation and guard then run per reference.  Sets `done` when the batches have been screened (otherwise nothing
// was touched and the caller takes the fp64 one-pass kernel).
int screen_many(muse_batch *const *bs, int32_t R, const int32_t *group_id, int32_t G_in, int32_t max_lag,
                       int32_t top_n, double threshold, int32_t sign_filter, int32_t abs_scores, bool &done)
{
    done = false;
    if (!bs || R < 2 || !bs[0])
        return MUSE_OK;
    muse_batch *b0 = bs[0];
    muse_ctx *ctx = b0->ctx;
    const int64_t M = b0->g->M;
    if (sign_filter < -1 || sign_filter > 1 || (group_id && G_in < 0))
        return MUSE_OK; // (run_select reports the error)
    for (int r = 0; r < R; r++) {
        if (!bs[r] || bs[r]->ctx != ctx || bs[r]->g != b0->g)
            return MUSE_OK; // (muse_batch_score_many reports the error)
        for (int q = 0; q < r; q++)
            if (bs[q] == bs[r])
                return MUSE_OK;
        if (bs[r]->N != 4096 ||
            screen_path(bs[r], run_key(bs[r], group_id, group_id ? (int64_t)G_in : 0, max_lag, top_n, threshold, sign_filter, abs_scores),
                        false) != MUSE_RUN_PATH_SCREENED)
            return MUSE_OK;
    }
    const int64_t G = group_id ? (int64_t)G_in : M;
    if (M == 0 || G == 0 || G > 0x7fffffffLL)
        return MUSE_OK;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    const bool on_device = select_on_device(top_n, G);
    const int K = on_device ? top_n : 1;
    std::vector<ScreenPlan> plan((size_t)R);
    double Es_max = 0.0;
    for (int r = 0; r < R; r++) {
        rc = ensure_select_ws(bs[r], M, G, group_id != nullptr, K, on_device);
        if (rc)
            return rc;
        rc = upload_group_ids(bs[r], group_id, M);
        if (rc)
            return rc;
        rc = screen_prepare(bs[r], top_n, group_id ? bs[r]->gid_dev : nullptr, G, plan[(size_t)r]);
        if (rc)
            return rc;
        Es_max = std::max(Es_max, plan[(size_t)r].Es);
    }
    if (!ctx->zscratch) {
        const int slots = ctx->num_cus * 4; // one 64 KB slice per resident workgroup of the fp64 one-pass kernel
        HIP_TRY(hipMalloc(&ctx->zscratch, (size_t)slots * 4096 * sizeof(double2)));
        ctx->zslots = slots;
    }
    if (R > ctx->many_cap) {
        (void)hipFree(ctx->many_tab);
        ctx->many_tab = nullptr;
        ctx->many_cap = 0;
        HIP_TRY(hipMalloc(&ctx->many_tab, (size_t)R * 5 * sizeof(void *)));
        ctx->many_cap = R;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // the previous pass may still be reading the host image
    std::vector<void *> &tab = ctx->many_host;
    tab.assign((size_t)R * 5, nullptr);
    for (int r = 0; r < R; r++) {
        tab[(size_t)r] = bs[r]->xcf;
        tab[(size_t)R + r] = bs[r]->mv;
        tab[(size_t)2 * R + r] = bs[r]->lag;
        tab[(size_t)3 * R + r] = bs[r]->scr_flags;
        tab[(size_t)4 * R + r] = bs[r]->scr_var;
    }
    HIP_TRY(hipMemcpyAsync(ctx->many_tab, tab.data(), tab.size() * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    ScreenPlan widest;
    widest.Es = Es_max; // one window for the pass: the widest of the references' (a wider window only flags more lags)
    FusedParams p = screen_pass_params(b0, max_lag, widest);
    p.R = R;
    p.xcf_many = (const float2 *const *)ctx->many_tab;
    p.mv_many = (double *const *)((void **)ctx->many_tab + R);
    p.lag_many = (int *const *)((void **)ctx->many_tab + 2 * R);
    p.flags_many = (unsigned *const *)((void **)ctx->many_tab + 3 * R);
    p.var_many = (double *const *)((void **)ctx->many_tab + 4 * R);
    p.zscratch = ctx->zscratch;
    p.zslots = ctx->zslots;
    LaunchTimer timer(ctx);
    HIP_TRY(timer.begin());
    HIP_TRY(launch_screen_pass_many(p, ctx->num_cus, ctx->stream));
    HIP_TRY(timer.end());
    for (int r = 0; r < R; r++) {
        rc = screen_finish(bs[r], top_n, threshold, sign_filter, abs_scores, group_id ? bs[r]->gid_dev : nullptr, G,
                           plan[(size_t)r]);
        if (rc)
            return rc;
    }
    done = true;
    return MUSE_OK;
}
