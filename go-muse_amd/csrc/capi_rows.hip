// capi_rows.hip -- Muse.Run (muse.go:46-92) as ONE call: the rows of one small label group from host memory against a
// template batch's reference, the group's winner back.
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
//
// Why it exists: muse_group_upload + muse_batch_create_like + muse_batch_run + two frees cost a dozen hipMalloc / hipFree
// (each a device-wide synchronisation) and three stream synchronisations per Muse.Run -- 110 us for 5 x 480 samples, all of it
// host side.  Here a context keeps a small pool of SLOTS (pinned staging, device rows behind the usual guard, score buffers,
// a pinned result record, an event); a call takes a slot, copies the rows into the staging buffer, and enqueues exactly
//     one host -> HBM copy, the fused kernel automatic selection takes for the length, one single-workgroup kernel that
//     reduces the group (reduce_kernels.hip, single_group_kernel) and writes the winner straight into pinned host memory,
//     one event
// on the SLOT's stream, waits for the event and returns the slot.  Steady state: no allocation, no hipFree, no device-wide
// synchronisation.  Many goroutines may drive one Muse (muse_test.go:203-214): every call in flight owns its slot, and
// since every slot owns a stream the callers' kernels -- a few dozen workgroups each -- run beside each other instead of
// queueing on one stream (round 5: sixteen callers gained 1.85 x over one; profiles/r06_cold_path.txt for now).  FFT lengths
// whose kernels work in the context's shared scratch buffer (n >= 8192) stay on the context's stream.
#include "capi_internal.h"

using namespace muse;

namespace {
constexpr size_t ROWS_GUARD = 8192;            // = capi_group.hip's GROUP_GUARD: padded series read in front of row 0
constexpr size_t ROWS_SLOT_MIN_ELEMS = 1u << 16;   // 512 KB: 5 ... 50 series of 480 ... 1000 samples without ever growing
constexpr size_t ROWS_SLOT_MAX_ELEMS = 1u << 24;   // 128 MB: larger groups take the general path (the copies dominate there)
constexpr size_t ROWS_ZERO_COPY_BYTES = 256u << 10; // up to here the kernel reads the pinned staging buffer itself
constexpr size_t ROWS_SLOT_KEEP_ELEMS = 1u << 20;   // 8 MB: a slot that grew beyond hands its buffers back when it is returned
constexpr int ROWS_SLOTS_KEPT = 16;            // slots kept per context when idle (more callers in flight: created and freed)
} // namespace

struct RowsSlot {
    muse_group g;            // rows = the slot's device buffer; N / stride / M set per call
    muse_batch b;            // spectrum tables rebound per call (the template's)
    double *dev = nullptr;   // allocation base (guard in front of g.rows)
    double *host = nullptr;  // pinned staging (ROWS_GUARD zeroed elements in front of it)
    double *host_dev = nullptr; // the same memory as the device addresses it
    size_t cap_elems = 0;
    SingleGroupOut *out = nullptr; // pinned: the winner record, written by the device
    hipEvent_t done = nullptr;
    hipStream_t stream = nullptr;  // the slot's own stream
};

static void slot_destroy(RowsSlot *s)
{
    if (!s)
        return;
    muse_ctx *ctx = s->b.ctx;
    if (s->stream)
        (void)hipStreamSynchronize(s->stream);
    dfree(ctx, s->dev);
    if (s->host)
        hfree(ctx, s->host - ROWS_GUARD);
    if (s->out)
        (void)hipHostFree(s->out);
    if (s->done)
        (void)hipEventDestroy(s->done);
    dfree(ctx, s->g.hstats);
    dfree(ctx, s->b.mv);
    dfree(ctx, s->b.lag);
    dfree(ctx, s->b.ovf_count);
    dfree(ctx, s->b.ovf_list);
    if (s->b.handoff_host)
        hfree(ctx, s->b.handoff_host);
    if (s->stream)
        (void)hipStreamDestroy(s->stream);
    delete s;
}

// called by ctx_release with the context's streams idle
void rows_slots_free(muse_ctx *ctx)
{
    for (void *p : ctx->rows_slots)
        slot_destroy((RowsSlot *)p);
    ctx->rows_slots.clear();
}

static int slot_reserve(RowsSlot *s, size_t elems)
{
    if (elems <= s->cap_elems)
        return MUSE_OK;
    size_t cap = ROWS_SLOT_MIN_ELEMS;
    while (cap < elems)
        cap *= 2;
    muse_ctx *ctx = s->b.ctx;
    dfree(ctx, s->dev);
    if (s->host)
        hfree(ctx, s->host - ROWS_GUARD);
    s->dev = nullptr;
    s->host = nullptr;
    s->cap_elems = 0;
    s->g.rows = nullptr;
    HIP_TRY(dmalloc(ctx, &s->dev, (cap + ROWS_GUARD) * sizeof(double)));
    HIP_TRY(hipMemsetAsync(s->dev, 0, ROWS_GUARD * sizeof(double), s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream)); // (growth is rare; the call may go on to use the context's stream instead of the slot's)
    // (the staging buffer carries the same zeroed guard: the smallest groups are read by the kernel straight out of it)
    double *hbase = nullptr;
    HIP_TRY(hmalloc(ctx, &hbase, (cap + ROWS_GUARD) * sizeof(double)));
    memset(hbase, 0, ROWS_GUARD * sizeof(double));
    s->host = hbase + ROWS_GUARD;
    s->host_dev = nullptr;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, hbase, 0) == hipSuccess && dp)
        s->host_dev = (double *)dp + ROWS_GUARD;
    s->g.rows = s->dev + ROWS_GUARD;
    s->cap_elems = cap;
    return MUSE_OK;
}

static int slot_acquire(muse_ctx *ctx, size_t elems, RowsSlot **out)
{
    RowsSlot *s = nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->rows_mu);
        if (!ctx->rows_slots.empty()) {
            s = (RowsSlot *)ctx->rows_slots.back();
            ctx->rows_slots.pop_back();
        }
    }
    if (!s) {
        s = new (std::nothrow) RowsSlot();
        if (!s)
            return fail(MUSE_ERR_NOMEM, "host allocation failed");
        s->g.ctx = ctx;
        s->b.ctx = ctx;
        s->b.g = &s->g;
        // (coherent whatever HIP_HOST_COHERENT says: the caller polls the record's state word while the kernel that writes it is still running)
        hipError_t e = hipHostMalloc((void **)&s->out, sizeof(SingleGroupOut), hipHostMallocCoherent | hipHostMallocMapped);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
        if (e == hipSuccess)
            e = dmalloc(ctx, &s->b.ovf_count, 2 * sizeof(int));
        if (e != hipSuccess) {
            slot_destroy(s);
            return fail(MUSE_ERR_HIP, "slot set-up failed: %s", hipGetErrorString(e));
        }
    }
    int rc = slot_reserve(s, elems);
    if (rc) {
        slot_destroy(s);
        return rc;
    }
    *out = s;
    return MUSE_OK;
}

static void slot_return(muse_ctx *ctx, RowsSlot *s)
{
    if (s->cap_elems > ROWS_SLOT_KEEP_ELEMS) { // one burst of large groups must not pin 16 x 128 MB for the life of the context
        dfree(ctx, s->dev);
        hfree(ctx, s->host - ROWS_GUARD);
        s->dev = nullptr;
        s->host = s->host_dev = nullptr;
        s->g.rows = nullptr;
        s->cap_elems = 0;
    }
    {
        std::lock_guard<std::mutex> lock(ctx->rows_mu);
        if ((int)ctx->rows_slots.size() < ROWS_SLOTS_KEPT) {
            ctx->rows_slots.push_back(s);
            return;
        }
    }
    slot_destroy(s); // (the slot's work has completed: its event was waited for)
}

// the general path for groups too large for a slot: what the host mirrors did per Muse.Run before this entry point existed
static int run_rows_general(muse_batch *tmpl, const double *rows, const double *const *row_ptrs, int64_t M, int64_t row_stride,
                            int32_t abs_scores, muse_record *out_winner, uint8_t *out_state)
{
    muse_group *g = nullptr;
    muse_batch *b = nullptr;
    int rc;
    if (rows) {
        rc = muse_group_upload(tmpl->ctx, rows, M, tmpl->N, row_stride, &g);
    } else {
        rc = muse_group_create(tmpl->ctx, M, tmpl->N, &g);
        for (int64_t r = 0; r < M && !rc; r++)
            rc = muse_group_append(g, row_ptrs[r], 1, tmpl->N);
    }
    if (rc) {
        muse_group_free(g);
        return rc;
    }
    rc = muse_batch_create_like(tmpl, g, &b);
    if (!rc) {
        std::vector<int32_t> gid((size_t)M, 0);
        rc = muse_batch_run_groups(b, gid.data(), 1, 0, abs_scores, out_winner, out_state);
    }
    muse_batch_free(b);
    muse_group_free(g);
    return rc;
}

// rows: M x N with stride row_stride, or (rows == NULL) row_ptrs[r] -> the N samples of row r
static int run_rows(muse_batch *tmpl, const double *rows, const double *const *row_ptrs, int64_t M, int64_t row_stride,
                    int32_t abs_scores, muse_record *out_winner, uint8_t *out_state)
{
    if (!tmpl || !out_winner || !out_state || M < 0 || (M > 0 && !rows && !row_ptrs))
        return fail(MUSE_ERR_INVALID, "bad arguments");
    *out_winner = muse_record{-1, 0.0, 0, 0};
    *out_state = 0;
    if (M == 0) // muse.go:47-50: nothing to compare
        return MUSE_OK;
    const int32_t N = tmpl->N;
    if (rows && row_stride < N) // muse.go:68-70
        return fail(MUSE_ERR_LENGTH, "Encountered a comparison graph with differing length than the reference (%lld vs %d)",
                    (long long)row_stride, N);
    if (!rows)
        for (int64_t r = 0; r < M; r++)
            if (!row_ptrs[r])
                return fail(MUSE_ERR_INVALID, "row %lld is NULL", (long long)r);
    muse_ctx *ctx = tmpl->ctx;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    const size_t elems = (size_t)M * (size_t)N;
    if (elems > ROWS_SLOT_MAX_ELEMS)
        return run_rows_general(tmpl, rows, row_ptrs, M, row_stride, abs_scores, out_winner, out_state);
    RowsSlot *s = nullptr;
    rc = slot_acquire(ctx, elems, &s);
    if (rc)
        return rc;
    // the slot's group and batch take this call's shape and the template's reference
    s->g.N = N;
    s->g.stride = N;
    s->g.M = M;
    s->g.cap = M;
    s->g.hstats_rows = 0; // (other rows than the slot's last call: no kept statistics)
    s->b.N = tmpl->N;
    s->b.n = tmpl->n;
    s->b.logn = tmpl->logn;
    s->b.sp = tmpl->sp;
    adopt_spectrum(&s->b);
    s->b.handoff_M = -1; // (no kernel-selection memory across unrelated groups)
    // kernels that work in the context's shared scratch buffer are serialised on the context's stream
    s->b.own_stream = tmpl->n >= GENERIC_LDS_MAX_N ? nullptr : s->stream;
    const hipStream_t st = s->b.stream();
    if (M > s->b.score_cap) { // (grown in steps that small groups never reach twice)
        dfree(ctx, s->b.mv);
        dfree(ctx, s->b.lag);
        s->b.mv = nullptr;
        s->b.lag = nullptr;
        s->b.score_cap = 0;
        const int64_t cap = std::max<int64_t>(M, 4096);
        hipError_t ea = dmalloc(ctx, &s->b.mv, (size_t)cap * sizeof(double));
        if (ea == hipSuccess)
            ea = dmalloc(ctx, &s->b.lag, (size_t)cap * sizeof(int));
        if (ea != hipSuccess) {
            slot_destroy(s);
            return fail(MUSE_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(ea));
        }
        s->b.score_cap = cap;
    }
    if (rows && row_stride == N) {
        memcpy(s->host, rows, elems * sizeof(double));
    } else {
        for (int64_t r = 0; r < M; r++)
            memcpy(s->host + (size_t)r * (size_t)N, rows ? rows + (size_t)r * (size_t)row_stride : row_ptrs[r], (size_t)N * sizeof(double));
    }
    s->out->state = ~0ull;
    // The smallest groups are not copied at all: the fused kernel reads its rows once, so it reads them straight from the
    // pinned staging buffer over PCIe (a copy command in front of the kernel costs more than its 19 KB take to cross)
    hipError_t e = hipSuccess;
    if (s->host_dev && elems * sizeof(double) <= ROWS_ZERO_COPY_BYTES && !ctx->rows_always_copy.load(std::memory_order_relaxed)) {
        s->g.rows = s->host_dev;
    } else {
        s->g.rows = s->dev + ROWS_GUARD;
        e = hipMemcpyAsync(s->g.rows, s->host, elems * sizeof(double), hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) {
        rc = muse_batch_score(&s->b); // the fused kernel automatic selection takes for this length (and its redo launch, if any)
        if (!rc)
            e = launch_single_group(s->b.mv, s->b.lag, M, abs_scores ? 1 : 0, 0, s->out, st);
    }
    if (e != hipSuccess || rc) {
        (void)hipStreamSynchronize(st); // nothing of this call may still be using the slot
        slot_return(ctx, s);
        return rc ? rc : fail(MUSE_ERR_HIP, "muse_batch_run_rows: %s", hipGetErrorString(e));
    }
    // The reduction kernel's last act is the record (rec, a system-scope fence, then state) in coherent pinned memory: the
    // caller polls the state word instead of recording and waiting for an event -- one command-processor packet and one
    // runtime wait fewer per Muse.Run.  A kernel that never delivers (a device fault) is found by the stream synchronisation
    // the poll falls back to.
    {
        volatile unsigned long long *flag = (volatile unsigned long long *)&s->out->state;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spin = 1; *flag == ~0ull; spin++) {
            __builtin_ia32_pause();
            if ((spin & 1023u) == 0) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 200000000ll) // 0.2 s: hand over to the runtime
                    break;
            }
        }
        if (*flag == ~0ull) {
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                slot_destroy(s);
                return fail(MUSE_ERR_HIP, "muse_batch_run_rows: %s", hipGetErrorString(e));
            }
        }
    }
    const unsigned long long stt = *(volatile unsigned long long *)&s->out->state;
    *out_winner = s->out->rec;
    slot_return(ctx, s);
    if (stt > 2ull)
        return fail(MUSE_ERR_HIP, "muse_batch_run_rows: the result record did not arrive");
    *out_state = (uint8_t)stt;
    return MUSE_OK;
}

extern "C" int muse_batch_run_rows(muse_batch *tmpl, const double *rows, int64_t M, int64_t row_stride, int32_t abs_scores,
                                   muse_record *out_winner, uint8_t *out_state)
{
    if (M > 0 && !rows)
        return fail(MUSE_ERR_INVALID, "bad arguments");
    return run_rows(tmpl, rows, nullptr, M, row_stride, abs_scores, out_winner, out_state);
}

// the same with one pointer per row (each to the template's N samples): the series of a Muse.Run are separate slices
// (muse.go:46, compGraphs []*Series) -- they are gathered straight into the slot's pinned buffer, not packed by the caller first
extern "C" int muse_batch_run_row_ptrs(muse_batch *tmpl, const double *const *rows, int64_t M, int32_t abs_scores,
                                       muse_record *out_winner, uint8_t *out_state)
{
    if (M > 0 && !rows)
        return fail(MUSE_ERR_INVALID, "bad arguments");
    return run_rows(tmpl, nullptr, rows, M, 0, abs_scores, out_winner, out_state);
}

extern "C" int muse_test_rows_always_copy(muse_ctx *ctx, int32_t always_copy)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    ctx->rows_always_copy.store(always_copy != 0);
    return MUSE_OK;
}
