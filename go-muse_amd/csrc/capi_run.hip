// capi_run.hip -- Batch.Run / Results (muse_batch.go:99-130, results.go:46-87): group max, filter, top-N, the merges of sharded Runs
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
#include "capi_internal.h"

using namespace muse;


// ---- Results: Go container/heap on |score| (scores.go:25-27, results.go)
namespace {
struct GoHeap {
    std::vector<muse_record> h;
    static bool less(const muse_record &a, const muse_record &b) { return std::fabs(a.score) < std::fabs(b.score); }
    void up(size_t j)
    {
        for (;;) {
            if (j == 0)
                break;
            size_t i = (j - 1) / 2;
            if (!less(h[j], h[i]))
                break;
            std::swap(h[i], h[j]);
            j = i;
        }
    }
    void down(size_t i0, size_t n)
    {
        size_t i = i0;
        for (;;) {
            size_t j1 = 2 * i + 1;
            if (j1 >= n)
                break;
            size_t j = j1, j2 = j1 + 1;
            if (j2 < n && less(h[j2], h[j1]))
                j = j2;
            if (!less(h[j], h[i]))
                break;
            std::swap(h[i], h[j]);
            i = j;
        }
    }
    void push(const muse_record &r)
    {
        h.push_back(r);
        up(h.size() - 1);
    }
    muse_record pop()
    {
        size_t n = h.size() - 1;
        std::swap(h[0], h[n]);
        down(0, n);
        muse_record r = h.back();
        h.pop_back();
        return r;
    }
};

// Results.Update over `cands` (already filtered by passed()) in group order,
// then Results.Fetch: descending |score|.
std::vector<muse_record> heap_select(std::vector<muse_record> cands, int64_t top_n)
{
    std::stable_sort(cands.begin(), cands.end(), [](const muse_record &a, const muse_record &b) {
        if (a.group != b.group)
            return a.group < b.group;
        return a.series < b.series;
    });
    GoHeap hp;
    if (top_n > 0) {
        for (const auto &r : cands) {
            if ((int64_t)hp.h.size() == top_n) { // results.go:62-66
                if (std::fabs(r.score) > std::fabs(hp.h[0].score)) {
                    hp.pop();
                    hp.push(r);
                }
            } else {
                hp.push(r);
            }
        }
    }
    std::vector<muse_record> out(hp.h.size());
    for (size_t i = out.size(); i-- > 0;) // results.go:81-85
        out[i] = hp.pop();
    return out;
}
} // namespace

int ensure_select_ws(muse_batch *b, int64_t M, int64_t G, bool with_gid, int K, bool on_device)
{
    if (with_gid && M > b->gid_cap) {
        dfree(b->ctx, b->gid_dev);
        b->gid_dev = nullptr;
        b->gid_cap = 0;
        b->gid_valid = false;
        HIP_TRY(dmalloc(b->ctx, &b->gid_dev, (size_t)M * sizeof(int)));
        b->gid_cap = M;
    }
    if (G > b->grp_cap) {
        dfree(b->ctx, b->gw.key);
        dfree(b->ctx, b->gw.first);
        dfree(b->ctx, b->gw.win);
        dfree(b->ctx, b->rec);
        dfree(b->ctx, b->selkey);
        b->gw = GroupWork{nullptr, nullptr, nullptr};
        b->rec = nullptr;
        b->selkey = nullptr;
        b->grp_cap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->gw.key, (size_t)G * sizeof(unsigned long long)));
        HIP_TRY(dmalloc(b->ctx, &b->gw.first, (size_t)G * sizeof(long long)));
        HIP_TRY(dmalloc(b->ctx, &b->gw.win, (size_t)G * sizeof(long long)));
        HIP_TRY(dmalloc(b->ctx, &b->rec, (size_t)G * sizeof(muse_record)));
        HIP_TRY(dmalloc(b->ctx, &b->selkey, (size_t)G * sizeof(unsigned long long)));
        b->grp_cap = G;
    }
    const int64_t nb = (G + TOPN_CHUNK - 1) / TOPN_CHUNK;
    if (nb > b->cnt_cap) {
        dfree(b->ctx, b->cnt);
        b->cnt = nullptr;
        b->cnt_cap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->cnt, (size_t)nb * sizeof(int)));
        b->cnt_cap = nb;
    }
    if (nb * K > b->cand_cap) {
        dfree(b->ctx, b->cand);
        b->cand = nullptr;
        b->cand_cap = 0;
        HIP_TRY(dmalloc(b->ctx, &b->cand, (size_t)(nb * K) * sizeof(muse_record)));
        b->cand_cap = nb * K;
    }
    if (!on_device && G > b->rec_host_cap) { // the exact feed: every group's record and selection key through pinned memory
        if (b->rec_host)
            hfree(b->ctx, b->rec_host);
        if (b->key_host)
            hfree(b->ctx, b->key_host);
        b->rec_host = nullptr;
        b->key_host = nullptr;
        b->rec_host_cap = 0;
        const int64_t cap = std::max<int64_t>(G, 64);
        HIP_TRY(hmalloc(b->ctx, &b->rec_host, (size_t)cap * sizeof(muse_record)));
        HIP_TRY(hmalloc(b->ctx, &b->key_host, (size_t)cap * sizeof(unsigned long long)));
        b->rec_host_cap = cap;
    }
    if (on_device && nb > b->cnt_host_cap) {
        if (b->cnt_host)
            hfree(b->ctx, b->cnt_host); // (hipHostFree(NULL) leaves a sticky error behind)
        b->cnt_host = nullptr;
        b->cnt_host_cap = 0;
        HIP_TRY(hmalloc(b->ctx, &b->cnt_host, (size_t)nb * sizeof(int)));
        b->cnt_host_cap = nb;
    }
    if (on_device && nb * K > b->cand_host_cap) {
        if (b->cand_host)
            hfree(b->ctx, b->cand_host);
        b->cand_host = nullptr;
        b->cand_host_cap = 0;
        HIP_TRY(hmalloc(b->ctx, &b->cand_host, (size_t)(nb * K) * sizeof(muse_record)));
        b->cand_host_cap = nb * K;
    }
    return MUSE_OK;
}

// the label-group map of a Run on the device (re-sent only when it changed)
int upload_group_ids(muse_batch *b, const int32_t *group_id, int64_t M)
{
    if (!group_id)
        return MUSE_OK;
    const bool same = b->gid_valid && (int64_t)b->gid_host.size() == M &&
                      memcmp(b->gid_host.data(), group_id, (size_t)M * sizeof(int32_t)) == 0;
    if (!same) {
        b->gid_host.assign(group_id, group_id + M);
        HIP_TRY(hipMemcpyAsync(b->gid_dev, b->gid_host.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice,
                               b->stream()));
        b->gid_valid = true;
    }
    return MUSE_OK;
}

// Small Runs: launch_group_reduce's outcome from one launch, read from coherent pinned memory as the kernel's per-slot stamps arrive
// (reduce_kernels.hip, small_groups_kernel / ungrouped_slots_kernel).  *out stays valid until the batch's next Run.
static bool small_run(int64_t M, int64_t G, bool grouped)
{
    return grouped ? M <= SMALL_GROUPS_MAX_M && G <= SMALL_GROUPS_MAX_G : G <= SMALL_UNGROUPED_MAX;
}
// the batch's pinned slot buffer with room for `need` slots (from the context's free list, or new) and a stamp no slot of this
// context has ever held
static int small_acquire(muse_batch *b, int64_t need, unsigned long long *token)
{
    muse_ctx *ctx = b->ctx;
    {
        std::lock_guard<std::mutex> lock(ctx->small_mu);
        *token = ++ctx->small_token;
        if (b->small_out && b->small_cap < need) { // (a Run(nil) behind grouped Runs: the larger buffer)
            ctx->small_free.emplace_back(b->small_out, b->small_cap);
            b->small_out = nullptr;
        }
        for (size_t i = 0; !b->small_out && i < ctx->small_free.size(); i++)
            if (ctx->small_free[i].second >= need) {
                b->small_out = ctx->small_free[i].first;
                b->small_cap = ctx->small_free[i].second;
                ctx->small_free.erase(ctx->small_free.begin() + (long)i);
            }
    }
    if (!b->small_out) {
        const int cap = need <= SMALL_GROUPS_MAX_G ? SMALL_GROUPS_MAX_G : need <= SMALL_UNGROUPED_MAX ? SMALL_UNGROUPED_MAX : SMALL_DIRECT_MAX_SLOTS;
        HIP_TRY(hipHostMalloc((void **)&b->small_out, (size_t)cap * sizeof(SmallSlot), hipHostMallocCoherent | hipHostMallocMapped));
        memset(b->small_out, 0, (size_t)cap * sizeof(SmallSlot));
        b->small_cap = cap;
    }
    return MUSE_OK;
}
// waits until *stamp == token: polls for 0.2 s, then lets the runtime wait for the stream once (a kernel that never delivers)
struct StampWait {
    muse_batch *b;
    struct timespec t0;
    bool synced = false;
    explicit StampWait(muse_batch *batch) : b(batch) { clock_gettime(CLOCK_MONOTONIC, &t0); }
    int operator()(const volatile unsigned long long *stamp, unsigned long long token)
    {
        for (unsigned spin = 1; *stamp != token; spin++) {
            __builtin_ia32_pause();
            if ((spin & 1023u) != 0)
                continue;
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) <= 200000000ll)
                continue;
            if (synced)
                return fail(MUSE_ERR_HIP, "Run: the records did not arrive");
            HIP_TRY(hipStreamSynchronize(b->stream()));
            synced = true;
            clock_gettime(CLOCK_MONOTONIC, &t0);
        }
        return MUSE_OK;
    }
};
static int small_reduce(muse_batch *b, const SelectParams &sp, const SmallSlot **out)
{
    unsigned long long token;
    int rc = small_acquire(b, sp.G, &token);
    if (rc)
        return rc;
    const volatile SmallSlot *slots = (const volatile SmallSlot *)b->small_out;
    HIP_TRY(launch_small_groups(sp, (SmallSlot *)b->small_out, token, b->stream()));
    StampWait wait(b);
    for (int g = sp.G - 1; g >= 0; g--) // (the last slot first: the others have mostly arrived by then)
        if ((rc = wait(&slots[g].stamp, token)))
            return rc;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *out = (const SmallSlot *)b->small_out;
    return MUSE_OK;
}
// Run(nil) over more series than the exact feed takes (> 65 536): each chunk of TOPN_CHUNK series selects its best K from keys it
// computes itself and writes those candidates' records straight into pinned slots (reduce_kernels.hip, topn_ungrouped_kernel) --
// one launch and a poll where group_final + topn + two copies + a synchronisation were (the same candidates in the same order).
static bool direct_topn(int64_t G, int K)
{
    const int64_t nb = (G + TOPN_CHUNK - 1) / TOPN_CHUNK;
    return K >= 1 && K <= TOPN_DEVICE_MAX && nb * K + (nb + 1) / 2 <= SMALL_DIRECT_MAX_SLOTS;
}
static int direct_topn_reduce(muse_batch *b, const SelectParams &sp, int K, std::vector<muse_record> &cands)
{
    const int64_t nb = ((int64_t)sp.G + TOPN_CHUNK - 1) / TOPN_CHUNK;
    unsigned long long token;
    int rc = small_acquire(b, nb * K + (nb + 1) / 2, &token);
    if (rc)
        return rc;
    SmallSlot *cand = (SmallSlot *)b->small_out;
    CountSlot *cnt = (CountSlot *)(cand + nb * K);
    HIP_TRY(launch_topn_ungrouped(sp, K, cand, cnt, token, b->stream()));
    StampWait wait(b);
    for (int64_t blk = 0; blk < nb; blk++) {
        const volatile CountSlot *c = cnt + blk;
        if ((rc = wait(&c->stamp, token)))
            return rc;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const int n = (int)c->count;
        for (int r = 0; r < n; r++) {
            const volatile SmallSlot *sl = cand + blk * K + r;
            if ((rc = wait(&sl->stamp, token)))
                return rc;
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            cands.push_back(muse_record{sl->series, sl->score, sl->lag, (int32_t)std::min<int64_t>(sl->series, 0x7fffffffLL)});
        }
    }
    return MUSE_OK;
}

int run_select(muse_batch *b, const int32_t *group_id, int32_t G_in, int64_t series_offset, int32_t max_lag,
                      int32_t top_n, double threshold, int32_t sign_filter, int32_t abs_scores,
                      std::vector<muse_record> &out, bool already_scored, bool prescreened)
{
    out.clear();
    muse_ctx *ctx = b->ctx;
    const int64_t M = b->g->M;
    if (sign_filter < -1 || sign_filter > 1)
        return fail(MUSE_ERR_INVALID, "sign_filter must be -1, 0 or 1");
    if (group_id && G_in < 0)
        return fail(MUSE_ERR_INVALID, "negative group count");
    // Batch.Run re-scores on every call (muse_batch.go:116-122)
    // (prescreened: muse_batch_run_many has run the screening pass for several batches at once and finished this one)
    const muse_batch::RunKey rkey = run_key(b, group_id, group_id ? (int64_t)G_in : 0, max_lag, top_n, threshold, sign_filter, abs_scores);
    const int32_t path = prescreened ? MUSE_RUN_PATH_SCREENED : screen_path(b, rkey, already_scored);
    const bool screened = path == MUSE_RUN_PATH_SCREENED;
    b->last_path = path;
    int rc = (already_scored || screened) ? MUSE_OK : muse_batch_score(b);
    if (rc)
        return rc;
    const int64_t G = group_id ? (int64_t)G_in : M;
    if (M == 0 || G == 0 || top_n <= 0)
        return MUSE_OK;
    if (G > 0x7fffffffLL)
        return fail(MUSE_ERR_UNSUPPORTED, "more than 2^31-1 groups on one device");
    // Up to EXACT_FEED_MAX_GROUPS groups every group's record comes back (G x 32 B through pinned memory) and the heap below is
    // fed ONE Score per group in group order: the reference's own feed (muse_batch.go:124-128, results.go:55-72), so exactly tied
    // scores -- every series that clamps to 1.0 ties -- survive at the TopN boundary and come back from Fetch as they do there,
    // for every binding of this entry point.  Beyond that the device pre-selects each chunk's best top_n (ties at the boundary
    // then go to the lower group id: docs/HISTORY.md 8.3).
    const bool on_device = select_on_device(top_n, G);
    const int K = on_device ? top_n : 1;
    const bool small = !on_device && !screened && small_run(M, G, group_id != nullptr); // (a screened Run's host-side checks wait on the stream)
    const bool direct = on_device && !screened && !group_id && direct_topn(G, K);
    rc = ensure_select_ws(b, M, small || direct ? 0 : G, group_id != nullptr, K, on_device && !direct);
    if (rc)
        return rc;
    rc = upload_group_ids(b, group_id, M);
    if (rc)
        return rc;
    b->last_screened = screened;
    if (screened && !prescreened) { // fp32 screening pass, then fp64 for the rows that can reach the top-N (needs the selection workspace)
        rc = score_screened(b, max_lag, top_n, threshold, sign_filter, abs_scores, group_id ? b->gid_dev : nullptr, G);
        if (rc)
            return rc;
    }
    SelectParams sp{};
    sp.mv = b->mv;
    sp.lag = b->lag;
    sp.M = M;
    sp.group_id = group_id ? b->gid_dev : nullptr;
    sp.G = (int)G;
    sp.abs_scores = abs_scores ? 1 : 0;
    sp.max_lag = max_lag;
    sp.threshold = threshold;
    sp.sign_filter = sign_filter;
    sp.series_offset = series_offset;
    sp.include = screened ? b->include : nullptr;
    std::vector<muse_record> cands;
    if (small) {
        const SmallSlot *slot;
        rc = small_reduce(b, sp, &slot);
        if (rc)
            return rc;
        for (int64_t g = 0; g < G; g++)
            if (slot[g].key != 0u)
                cands.push_back(muse_record{slot[g].series, slot[g].score, slot[g].lag, (int32_t)g});
    } else if (direct) {
        rc = direct_topn_reduce(b, sp, K, cands);
        if (rc)
            return rc;
    } else if (on_device) {
        HIP_TRY(launch_group_reduce(sp, b->gw, b->rec, b->selkey, b->stream()));
        const int64_t nb = (G + TOPN_CHUNK - 1) / TOPN_CHUNK;
        HIP_TRY(launch_topn(b->rec, b->selkey, (int)G, K, b->cand, b->cnt, b->stream()));
        const int *cnt = b->cnt_host;
        const muse_record *cand = b->cand_host;
        HIP_TRY(hipMemcpyAsync(b->cnt_host, b->cnt, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, b->stream()));
        HIP_TRY(hipMemcpyAsync(b->cand_host, b->cand, (size_t)(nb * K) * sizeof(muse_record), hipMemcpyDeviceToHost,
                               b->stream()));
        HIP_TRY(hipStreamSynchronize(b->stream()));
        for (int64_t blk = 0; blk < nb; blk++)
            for (int r = 0; r < cnt[(size_t)blk]; r++)
                cands.push_back(cand[(size_t)(blk * K + r)]);
        // a screened Run that had to re-evaluate a large part of the rows (few rows certainly pass the filters, or the
        // scores crowd around the cut) costs more than the plain fp64 pass: not again for this (immutable) set of rows
        if (screened && (int64_t)*b->refine_host * 4 > (M + 1) / 2)
            b->costly_key = rkey;
        if (screened && screen_guard_tripped(b)) // an estimate left its bound: this Run is redone entirely in fp64
            return run_select(b, group_id, G_in, series_offset, max_lag, top_n, threshold, sign_filter, abs_scores, out, false);
    } else {
        HIP_TRY(launch_group_reduce(sp, b->gw, b->rec, b->selkey, b->stream()));
        const muse_record *rec = b->rec_host;
        const unsigned long long *key = b->key_host;
        HIP_TRY(hipMemcpyAsync(b->rec_host, b->rec, (size_t)G * sizeof(muse_record), hipMemcpyDeviceToHost, b->stream()));
        HIP_TRY(hipMemcpyAsync(b->key_host, b->selkey, (size_t)G * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               b->stream()));
        HIP_TRY(hipStreamSynchronize(b->stream()));
        for (int64_t g = 0; g < G; g++)
            if (key[(size_t)g] != 0ull)
                cands.push_back(rec[(size_t)g]);
        if (screened && (int64_t)*b->refine_host * 4 > (M + 1) / 2)
            b->costly_key = rkey;
        if (screened && screen_guard_tripped(b))
            return run_select(b, group_id, G_in, series_offset, max_lag, top_n, threshold, sign_filter, abs_scores, out, false);
    }
    if (!group_id) // ungrouped: global order of the groups is the global series index
        for (auto &r : cands)
            r.group = (int32_t)std::min<int64_t>(r.series, 0x7fffffffLL);
    out = heap_select(std::move(cands), top_n);
    return MUSE_OK;
}

void emit(const std::vector<muse_record> &sel, int64_t *out_series, int32_t *out_lag, double *out_score,
                 int32_t *out_count, double *out_mean_abs)
{
    double sum = 0.0;
    for (size_t i = sel.size(); i-- > 0;) { // results.go:81-85 sums in pop order (ascending |score|)
        if (out_series)
            out_series[i] = sel[i].series;
        if (out_lag)
            out_lag[i] = sel[i].lag;
        if (out_score)
            out_score[i] = sel[i].score;
        sum += std::fabs(sel[i].score);
    }
    if (out_count)
        *out_count = (int32_t)sel.size();
    if (out_mean_abs) // results.go:86 (0/0 = NaN when empty)
        *out_mean_abs = sel.empty() ? std::numeric_limits<double>::quiet_NaN() : sum / (double)sel.size();
}

extern "C" int muse_batch_run(muse_batch *b, const int32_t *group_id, int32_t G, int32_t max_lag, int32_t top_n,
                              double threshold, int32_t sign_filter, int32_t abs_scores, int64_t *out_series,
                              int32_t *out_lag, double *out_score, int32_t *out_count, double *out_mean_abs)
{
    if (!b)
        return fail(MUSE_ERR_INVALID, "NULL batch");
    std::vector<muse_record> sel;
    int rc = run_select(b, group_id, G, 0, max_lag, top_n, threshold, sign_filter, abs_scores, sel);
    if (rc)
        return rc;
    emit(sel, out_series, out_lag, out_score, out_count, out_mean_abs);
    return MUSE_OK;
}

extern "C" int muse_batch_run_shard(muse_batch *b, const int32_t *group_id, int32_t G, int64_t series_offset,
                                    int32_t max_lag, int32_t top_n, double threshold, int32_t sign_filter,
                                    int32_t abs_scores, muse_record *out_records, int32_t *out_count)
{
    if (!b || !out_count || (top_n > 0 && !out_records))
        return fail(MUSE_ERR_INVALID, "NULL argument");
    std::vector<muse_record> sel;
    int rc = run_select(b, group_id, G, series_offset, max_lag, top_n, threshold, sign_filter, abs_scores, sel);
    if (rc)
        return rc;
    for (size_t i = 0; i < sel.size(); i++)
        out_records[i] = sel[i];
    *out_count = (int32_t)sel.size();
    return MUSE_OK;
}

// Sharded Run whose label groups may straddle shards ("... or the per-group partial maxima are merged before
// top-N"): this shard's winner per label group, unfiltered, plus the group's state on this shard (SelectParams::partial)
extern "C" int muse_batch_run_groups(muse_batch *b, const int32_t *group_id, int32_t G, int64_t series_offset,
                                     int32_t abs_scores, muse_record *out_records, uint8_t *out_state)
{
    if (!b || !group_id || G < 0 || (G > 0 && (!out_records || !out_state)))
        return fail(MUSE_ERR_INVALID, "bad arguments (label groups are required: ungrouped Runs shard with muse_batch_run_shard)");
    muse_ctx *ctx = b->ctx;
    b->last_path = MUSE_RUN_PATH_FP64;
    b->last_screened = false;
    int rc = muse_batch_score(b);
    if (rc)
        return rc;
    const int64_t M = b->g->M;
    for (int32_t g = 0; g < G; g++) {
        out_records[g] = muse_record{-1, 0.0, 0, g};
        out_state[g] = 0;
    }
    if (M == 0 || G == 0)
        return MUSE_OK;
    const bool small = small_run(M, G, true);
    rc = small ? ensure_select_ws(b, M, 0, true, 1, false) : ensure_select_ws(b, M, G, true, 1, false);
    if (rc)
        return rc;
    rc = upload_group_ids(b, group_id, M);
    if (rc)
        return rc;
    SelectParams sp{};
    sp.mv = b->mv;
    sp.lag = b->lag;
    sp.M = M;
    sp.group_id = b->gid_dev;
    sp.G = G;
    sp.abs_scores = abs_scores ? 1 : 0;
    sp.series_offset = series_offset;
    sp.partial = 1;
    if (small) {
        const SmallSlot *slot;
        rc = small_reduce(b, sp, &slot);
        if (rc)
            return rc;
        for (int32_t g = 0; g < G; g++) {
            out_records[g] = muse_record{slot[g].series, slot[g].score, slot[g].lag, g};
            out_state[g] = (uint8_t)slot[g].key;
        }
        return MUSE_OK;
    }
    HIP_TRY(launch_group_reduce(sp, b->gw, b->rec, b->selkey, b->stream()));
    std::vector<unsigned long long> st((size_t)G);
    HIP_TRY(hipMemcpyAsync(out_records, b->rec, (size_t)G * sizeof(muse_record), hipMemcpyDeviceToHost, b->stream()));
    HIP_TRY(hipMemcpyAsync(st.data(), b->selkey, (size_t)G * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream()));
    HIP_TRY(hipStreamSynchronize(b->stream()));
    for (int32_t g = 0; g < G; g++)
        out_state[g] = (uint8_t)st[(size_t)g];
    return MUSE_OK;
}

// results.go:46-52 on the host (the merge of shards filters AFTER the group maxima are final)
static bool passed_host(double s, int32_t lag, int32_t max_lag, double threshold, int32_t sign_filter)
{
    return std::fabs((double)lag) <= (double)max_lag && std::fabs(s) >= threshold &&
           (sign_filter == 0 || (s > 0 && sign_filter == 1) || (s < 0 && sign_filter == -1));
}

// per label group: what the shards' records and states amount to (the group's winner; out_state 0 = no member anywhere,
// 1 = out_records[g] is the group's Score, 2 = the group's first member scores NaN, so the group's score is NaN)
static void merge_group_winners(const muse_record *records, const uint8_t *state, int32_t n_shards, int32_t G,
                                muse_record *out_records, uint8_t *out_state)
{
    for (int32_t g = 0; g < G; g++) {
        // shards are listed in ascending row order: the first one with a member holds the group's first member
        bool seen = false, nan_first = false, have = false;
        muse_record best{};
        best.series = -1;
        for (int32_t s = 0; s < n_shards; s++) {
            const size_t k = (size_t)s * (size_t)G + (size_t)g;
            if (state[k] == 0)
                continue;
            if (!seen) {
                seen = true;
                nan_first = state[k] == 2;
            }
            const muse_record &r = records[k];
            if (r.series < 0)
                continue;
            // muse_batch.go:87 / muse.go:86: a later series replaces the maximum only if strictly greater (by |score|)
            if (!have || std::fabs(r.score) > std::fabs(best.score)) {
                best = r;
                have = true;
            }
        }
        best.group = g;
        out_records[g] = best;
        out_state[g] = !seen ? 0 : (nan_first || !have) ? 2 : 1;
    }
}

extern "C" int muse_merge_group_winners(const muse_record *records, const uint8_t *state, int32_t n_shards, int32_t G,
                                        muse_record *out_records, uint8_t *out_state)
{
    if (n_shards < 0 || G < 0 || ((int64_t)n_shards * G > 0 && (!records || !state)) || (G > 0 && (!out_records || !out_state)))
        return fail(MUSE_ERR_INVALID, "bad shard records");
    merge_group_winners(records, state, n_shards, G, out_records, out_state);
    return MUSE_OK;
}

extern "C" int muse_merge_group_records(const muse_record *records, const uint8_t *state, int32_t n_shards, int32_t G,
                                        int32_t max_lag, int32_t top_n, double threshold, int32_t sign_filter,
                                        int64_t *out_series, int32_t *out_lag, double *out_score, int32_t *out_count,
                                        double *out_mean_abs)
{
    if (n_shards < 0 || G < 0 || ((int64_t)n_shards * G > 0 && (!records || !state)))
        return fail(MUSE_ERR_INVALID, "bad shard records");
    if (sign_filter < -1 || sign_filter > 1)
        return fail(MUSE_ERR_INVALID, "sign_filter must be -1, 0 or 1");
    std::vector<muse_record> win((size_t)G), cands;
    std::vector<uint8_t> st((size_t)G);
    merge_group_winners(records, state, n_shards, G, win.data(), st.data());
    for (int32_t g = 0; g < G; g++) // (an empty group, or one whose first member scores NaN, never passes Results.passed)
        if (st[(size_t)g] == 1 && passed_host(win[(size_t)g].score, win[(size_t)g].lag, max_lag, threshold, sign_filter))
            cands.push_back(win[(size_t)g]);
    std::vector<muse_record> sel = heap_select(std::move(cands), top_n);
    emit(sel, out_series, out_lag, out_score, out_count, out_mean_abs);
    return MUSE_OK;
}

extern "C" int muse_merge_records(const muse_record *records, int64_t count, int32_t top_n, int64_t *out_series,
                                  int32_t *out_lag, double *out_score, int32_t *out_count, double *out_mean_abs)
{
    if (count < 0 || (count > 0 && !records))
        return fail(MUSE_ERR_INVALID, "bad records");
    std::vector<muse_record> c(records, records + count);
    std::vector<muse_record> sel = heap_select(std::move(c), top_n);
    emit(sel, out_series, out_lag, out_score, out_count, out_mean_abs);
    return MUSE_OK;
}
