// capi_many.hip -- many references against one resident Group in one pass
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
#include "capi_internal.h"

using namespace muse;


// -------------------------------------------------------- many references
extern "C" int muse_batch_read_scores(muse_batch *b, int32_t *lag, double *mv)
{
    if (!b)
        return fail(MUSE_ERR_INVALID, "NULL batch");
    int rc = use_device(b->ctx);
    if (rc)
        return rc;
    const int64_t M = b->g->M;
    if (M == 0)
        return MUSE_OK;
    if (!lag || !mv)
        return fail(MUSE_ERR_INVALID, "NULL output");
    if (M > b->score_cap)
        return fail(MUSE_ERR_INVALID, "the batch has not been scored since the group grew");
    if (!b->scores_exact) { // the last Run screened in fp32 and re-evaluated only the rows it needed
        rc = muse_batch_score(b);
        if (rc)
            return rc;
    }
    HIP_TRY(hipMemcpyAsync(lag, b->lag, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipMemcpyAsync(mv, b->mv, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return MUSE_OK;
}

extern "C" int muse_batch_score_many(muse_batch *const *bs, int32_t R)
{
    if (!bs || R < 1)
        return fail(MUSE_ERR_INVALID, "bad batch list");
    for (int r = 0; r < R; r++) {
        if (!bs[r])
            return fail(MUSE_ERR_INVALID, "NULL batch in list");
        if (bs[r]->ctx != bs[0]->ctx || bs[r]->g != bs[0]->g)
            return fail(MUSE_ERR_INVALID, "batches of one pass must share the context and the comparison group");
        for (int q = 0; q < r; q++)
            if (bs[q] == bs[r])
                return fail(MUSE_ERR_INVALID, "the same batch appears twice in the list");
    }
    muse_batch *b0 = bs[0];
    muse_ctx *ctx = b0->ctx;
    // the one-pass kernel is built for N == n == 4096 (and is only taken under automatic kernel
    // selection); everything else scores the batches one after the other
    const bool small_n = (b0->n >= 512 && b0->n <= 2048) || b0->n == 8192 || b0->n == 16384; // xcorr_small.hip's lengths
    // (float32-storage groups: the n = 4096 one-pass kernel reads them; the other lengths' one-pass builds do not)
    // long series (xcorr_long.hip, MULTI): 3 + 3 R slice crossings per pair against 4 R -- from three references on
    // n = 32768 (round 6): ONE real series per workgroup, first-transformed once, every reference from the parked spectrum
    // (xcorr_fused_real32k_multi: 1 x the row bytes where the four-step kernel below moves 5 x); from three references on (two: as
    // fast as two passes -- profiles/r06_many_refs.txt)
    bool real_n = b0->n == 32768 && R >= 3 && !b0->g->f32 && ctx->variant == 0 && ctx->gsmall[4] && ctx->wsplit;
    for (int r = 0; r < R && real_n; r++)
        real_n = bs[r]->N == b0->N && bs[r]->xcw != nullptr;
    const bool long_n = !real_n && (b0->n == 32768 || b0->n == 65536) && R >= 3 && !b0->g->f32 && ctx->variant == 0 && b0->logn >= 14 && ctx->twl[b0->logn - 14];
    bool one_pass = R > 1 &&
                    ((b0->n == 4096 && (ctx->variant == 0 || ctx->variant == 10)) ||
                     (small_n && !b0->g->f32 && (ctx->variant == 0 || ctx->variant == 12)) || long_n || real_n);
    for (int r = 0; r < R && one_pass; r++)
        one_pass = bs[r]->N == b0->N && (small_n || real_n || b0->N == b0->n || bs[r]->c1 != nullptr) && (!long_n || bs[r]->xcp != nullptr);
    if (!one_pass) {
        for (int r = 0; r < R; r++) {
            int rc = muse_batch_score(bs[r]);
            if (rc)
                return rc;
        }
        return MUSE_OK;
    }
    int rc = use_device(ctx);
    if (rc)
        return rc;
    rc = group_ready(b0->g);
    if (rc)
        return rc;
    const int64_t M = b0->g->M;
    if (M == 0)
        return MUSE_OK;
    for (int r = 0; r < R; r++) {
        rc = ensure_scores(bs[r]);
        if (rc)
            return rc;
    }
    if (long_n) // two n-element slices per resident workgroup
        HIP_TRY(ensure_gscratch(ctx, b0->n, 2 * LONG_WGS_PER_CU));
    if (real_n) // (half an n-element slice per workgroup: the batches' own creation sized the buffer for far more)
        HIP_TRY(ensure_gscratch(ctx, b0->n));
    if (b0->n == 16384 && !ctx->zscratch) { // (every other length keeps the spectra in registers: no scratch)
        const int slots = ctx->num_cus * 4; // one 64 KB slice per resident workgroup
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
    tab.assign((size_t)R * 4, nullptr);
    for (int r = 0; r < R; r++) {
        tab[(size_t)r] = small_n ? bs[r]->xc : real_n ? bs[r]->xcw : bs[r]->xcp;
        tab[(size_t)R + r] = bs[r]->mv;
        tab[(size_t)2 * R + r] = bs[r]->lag;
        tab[(size_t)3 * R + r] = bs[r]->c1;
    }
    HIP_TRY(hipMemcpyAsync(ctx->many_tab, tab.data(), tab.size() * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    FusedParams p = base_params(b0);
    p.R = R;
    p.xcp_many = (const double2 *const *)ctx->many_tab;
    p.mv_many = (double *const *)((void **)ctx->many_tab + R);
    p.lag_many = (int *const *)((void **)ctx->many_tab + 2 * R);
    p.c1_many = (const double *const *)((void **)ctx->many_tab + 3 * R);
    p.zscratch = ctx->zscratch;
    p.zslots = ctx->zslots;
    if (2 * p.npairs > b0->ovf_cap) {
        dfree(b0->ctx, b0->ovf_list);
        b0->ovf_list = nullptr;
        b0->ovf_cap = 0;
        HIP_TRY(dmalloc(b0->ctx, &b0->ovf_list, (size_t)(2 * p.npairs) * sizeof(long long)));
        b0->ovf_cap = 2 * p.npairs;
    }
    p.ovf_count = b0->ovf_count;
    p.work_counter = b0->ovf_count + 1;
    p.ovf_list = b0->ovf_list;
    // (the long-series kernel works in the context's scratch buffer: its pointer must not be swapped between reading it and the launch)
    std::unique_lock<std::mutex> scratch_lock(ctx->stage_mu, std::defer_lock);
    if (long_n || real_n) {
        scratch_lock.lock();
        p.gscratch = ctx->gscratch;
        p.gscratch_slices = (long long)(ctx->gscratch_elems / (size_t)b0->n);
    }
    LaunchTimer timer(ctx);
    HIP_TRY(hipMemsetAsync(b0->ovf_count, 0, 2 * sizeof(int), ctx->stream));
    HIP_TRY(timer.begin());
    if (small_n) { // (this kernel isolates dead series itself: nothing is handed on)
        HIP_TRY(launch_fused_small(p, ctx->num_cus, ctx->stream));
        for (int r = 0; r < R; r++)
            bs[r]->scores_exact = true;
    } else if (real_n) { // (one series per transform: nothing to isolate, no redo list)
        p.gsmall = ctx->gsmall[4];
        HIP_TRY(launch_fused_real_split(p, ctx->num_cus, ctx->stream));
        for (int r = 0; r < R; r++)
            bs[r]->scores_exact = true;
    } else if (long_n)
        HIP_TRY(launch_fused_long(p, ctx->num_cus, ctx->stream));
    else
        HIP_TRY(launch_fused_multi(p, ctx->num_cus, ctx->stream));
    HIP_TRY(timer.end());
    // pairs holding a NaN/Inf series (listed once, by reference 0): redone per reference by the
    // kernel that isolates the dead series before the shared transform
    LaunchTimer redo_timer(ctx, true); // (one bracket around the R redo launches)
    if (!small_n && !real_n)
        HIP_TRY(redo_timer.begin());
    for (int r = 0; r < R && !small_n && !real_n; r++) {
        FusedParams q = base_params(bs[r]);
        q.pair_list = b0->ovf_list;
        q.pair_count = b0->ovf_count;
        if (long_n) { // (the four-step kernel that isolates and rescales first, as behind a single long-series pass)
            q.npairs = std::min<long long>(q.npairs, (long long)ctx->num_cus * STOCKHAM_GLOBAL_WGS_PER_CU);
            HIP_TRY(launch_fused(q, KERNEL_STOCKHAM, ctx->num_cus, ctx->stream));
        } else {
            q.npairs = std::min<long long>(q.npairs, (long long)ctx->num_cus * 3);
            HIP_TRY(launch_fused(q, KERNEL_R16_OCC3, ctx->num_cus, ctx->stream));
        }
        bs[r]->scores_exact = true; // mv / lag of every batch now hold fp64 results for every row
    }
    HIP_TRY(redo_timer.end());
    return MUSE_OK;
}

extern "C" int muse_batch_run_many(muse_batch *const *bs, int32_t R, const int32_t *group_id, int32_t G,
                                   int32_t max_lag, int32_t top_n, double threshold, int32_t sign_filter,
                                   int32_t abs_scores, int64_t *out_series, int32_t *out_lag, double *out_score,
                                   int32_t *out_count, double *out_mean_abs)
{
    bool prescreened = false;
    int rc = screen_many(bs, R, group_id, G, max_lag, top_n, threshold, sign_filter, abs_scores, prescreened);
    if (rc)
        return rc;
    if (!prescreened) {
        rc = muse_batch_score_many(bs, R);
        if (rc)
            return rc;
    }
    const size_t cap = (size_t)std::max(top_n, 0);
    for (int r = 0; r < R; r++) {
        std::vector<muse_record> sel;
        rc = run_select(bs[r], group_id, G, 0, max_lag, top_n, threshold, sign_filter, abs_scores, sel, true, prescreened);
        if (rc)
            return rc;
        emit(sel, out_series ? out_series + cap * r : nullptr, out_lag ? out_lag + cap * r : nullptr,
             out_score ? out_score + cap * r : nullptr, out_count ? out_count + r : nullptr,
             out_mean_abs ? out_mean_abs + r : nullptr);
    }
    return MUSE_OK;
}
