// capi_group.hip -- the resident comparison Group (Group.Add, group.go:31-56): staging, uploads, reads
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
#include "capi_internal.h"

using namespace muse;


// ------------------------------------------------------------------- group
// Every group allocation starts with GROUP_GUARD readable (zeroed) elements in front of row 0: the kernels for zero-padded
// series (xcorr_small.hip) read up to n - N samples in front of a row without clamping and mask them afterwards.
constexpr size_t GROUP_GUARD = 8192;
static_assert(GROUP_GUARD >= (size_t)SMALL_MAX_N / 2, "xcorr_small.hip reads up to n - N < n / 2 samples in front of row 0");
static int group_create(muse_ctx *ctx, int64_t capacity_rows, int32_t N, bool f32, muse_group **out);
extern "C" int muse_group_create(muse_ctx *ctx, int64_t capacity_rows, int32_t N, muse_group **out)
{
    return group_create(ctx, capacity_rows, N, false, out);
}
extern "C" int muse_group_create_f32(muse_ctx *ctx, int64_t capacity_rows, int32_t N, muse_group **out)
{
    // (the float32-row loaders are built into the kernels automatic selection takes for FFT lengths 512 ... 16384)
    if (N <= 256 || N > 16384)
        return fail(MUSE_ERR_UNSUPPORTED, "float32-storage groups are built for series of length 257 .. 16384 (got %d)", N);
    return group_create(ctx, capacity_rows, N, true, out);
}
// a fresh allocation's guard is zeroed on the copy stream (no device-wide synchronisation): `uploaded` orders every reader behind it
static hipError_t zero_guard(muse_group *g, void *base)
{
    hipError_t e = hipMemsetAsync(base, 0, GROUP_GUARD * g->elem(), g->ctx->copy_stream);
    if (e == hipSuccess)
        e = hipEventRecord(g->uploaded, g->ctx->copy_stream);
    if (e == hipSuccess)
        g->upload_pending = true;
    return e;
}

static int group_create(muse_ctx *ctx, int64_t capacity_rows, int32_t N, bool f32, muse_group **out)
{
    if (!out)
        return fail(MUSE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (N < 1 || capacity_rows < 0)
        return fail(MUSE_ERR_INVALID, "bad group shape (%lld x %d)", (long long)capacity_rows, N);
    muse_group *g = new (std::nothrow) muse_group();
    if (!g)
        return fail(MUSE_ERR_NOMEM, "host allocation failed");
    g->ctx = ctx;
    g->N = N;
    g->stride = N;
    g->cap = capacity_rows;
    g->f32 = f32;
    if (hipEventCreateWithFlags(&g->uploaded, hipEventDisableTiming) != hipSuccess) {
        delete g;
        return fail(MUSE_ERR_HIP, "hipEventCreate failed");
    }
    if (capacity_rows > 0) {
        // (from the context's allocation cache: a block a freed group handed back carries that group's rows and guard --
        // the guard is zeroed again here, the rows are only ever read below M)
        void *mem = nullptr;
        hipError_t e = dmalloc(ctx, &mem, ((size_t)capacity_rows * (size_t)N + GROUP_GUARD) * g->elem());
        if (e == hipSuccess)
            e = zero_guard(g, mem);
        if (e != hipSuccess) {
            dfree(ctx, mem);
            (void)hipEventDestroy(g->uploaded);
            delete g;
            return fail(MUSE_ERR_NOMEM, "hipMalloc of %lld x %d samples failed: %s", (long long)capacity_rows, N,
                        hipGetErrorString(e));
        }
        mem = (char *)mem + GROUP_GUARD * g->elem();
        (f32 ? (void *&)g->rows32 : (void *&)g->rows) = mem;
    }
    ctx->refs.fetch_add(1);
    *out = g;
    return MUSE_OK;
}

// Growth never waits for the device: the new block's guard is zeroed and the rows are copied over ON THE COPY STREAM (behind
// every upload into the old block that is still in flight), `uploaded` orders the readers behind that, and the old block stays
// alive until the group goes -- a score pass enqueued earlier may still be reading it.
static int group_reserve(muse_group *g, int64_t rows)
{
    if (rows <= g->cap)
        return MUSE_OK;
    int64_t ncap = std::max<int64_t>(rows, g->cap * 2);
    void *nr = nullptr;
    hipError_t e = dmalloc(g->ctx, &nr, ((size_t)ncap * (size_t)g->N + GROUP_GUARD) * g->elem());
    if (e != hipSuccess)
        return fail(MUSE_ERR_NOMEM, "hipMalloc of %lld rows failed: %s", (long long)ncap, hipGetErrorString(e));
    HIP_TRY(zero_guard(g, nr));
    nr = (char *)nr + GROUP_GUARD * g->elem();
    if (g->M > 0) {
        // (rows still packed in the staging buffer and not sent yet are not in the old block: only the sent ones are copied)
        const int64_t sent = g->M - (g->staged - g->flushed);
        if (sent > 0)
            HIP_TRY(hipMemcpyAsync(nr, g->base(), (size_t)sent * (size_t)g->N * g->elem(), hipMemcpyDeviceToDevice, g->ctx->copy_stream));
        HIP_TRY(hipEventRecord(g->uploaded, g->ctx->copy_stream));
        g->upload_pending = true;
    }
    if (g->base())
        g->retired.push_back((char *)g->base() - GROUP_GUARD * g->elem());
    (g->f32 ? (void *&)g->rows32 : (void *&)g->rows) = nr;
    g->cap = ncap;
    return MUSE_OK;
}

constexpr size_t STAGE_BYTES = 32u << 20;      // one staging buffer
constexpr size_t STAGE_FLUSH_BYTES = 1u << 20; // packed rows go out in pieces of at least this size

// the staging pair, borrowed from the context's pool on first use
static int group_stage_buffers(muse_group *g)
{
    if (g->stage[0])
        return MUSE_OK;
    g->stage_rows = std::max<int64_t>(1, (int64_t)(STAGE_BYTES / ((size_t)g->N * g->elem())));
    for (int i = 0; i < 2; i++) {
        double *buf = nullptr;
        {
            std::lock_guard<std::mutex> lock(g->ctx->stage_mu);
            if (!g->ctx->stage_pool.empty()) {
                buf = g->ctx->stage_pool.back();
                g->ctx->stage_pool.pop_back();
            }
        }
        if (!buf)
            HIP_TRY(hipHostMalloc((void **)&buf, std::max(STAGE_BYTES, (size_t)g->N * g->elem()), hipHostMallocDefault));
        g->stage[i] = buf;
        HIP_TRY(hipEventCreateWithFlags(&g->stage_done[i], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g->stage_done[i], g->ctx->copy_stream));
    }
    return MUSE_OK;
}

// enqueue the upload of the packed rows not sent yet (asynchronous); a full buffer is left for the other one
static int group_flush(muse_group *g)
{
    if (g->staged > g->flushed) {
        const int64_t k = g->staged - g->flushed, first = g->M - k;
        HIP_TRY(hipMemcpyAsync((char *)g->base() + (size_t)(first * g->stride) * g->elem(),
                               (char *)g->stage[g->cur] + (size_t)g->flushed * (size_t)g->N * g->elem(),
                               (size_t)k * (size_t)g->N * g->elem(), hipMemcpyHostToDevice, g->ctx->copy_stream));
        HIP_TRY(hipEventRecord(g->stage_done[g->cur], g->ctx->copy_stream));
        HIP_TRY(hipEventRecord(g->uploaded, g->ctx->copy_stream));
        g->upload_pending = true;
        g->flushed = g->staged;
    }
    if (g->staged == g->stage_rows && g->stage_rows > 0) {
        g->staged = g->flushed = 0;
        g->cur ^= 1;
        HIP_TRY(hipEventSynchronize(g->stage_done[g->cur])); // the other buffer's last upload has landed
    }
    return MUSE_OK;
}

// packed rows enqueued for upload, and `stream` (the context's compute stream by default) ordered behind every upload
// enqueued so far: call before anything on that stream reads the rows
int group_ready(muse_group *g, hipStream_t stream)
{
    if (g->win_rows)
        return fail(MUSE_ERR_INVALID, "the group has an open staging window (muse_group_stage without its commits)");
    int rc = group_flush(g);
    if (rc)
        return rc;
    if (g->upload_pending) {
        HIP_TRY(hipStreamWaitEvent(stream ? stream : g->ctx->stream, g->uploaded, 0));
        g->upload_pending = false;
    }
    return MUSE_OK;
}

extern "C" int muse_group_append(muse_group *g, const double *rows, int64_t count, int64_t row_stride)
{
    if (!g || (!rows && count > 0) || count < 0)
        return fail(MUSE_ERR_INVALID, "bad append arguments");
    if (count == 0)
        return MUSE_OK;
    if (row_stride < g->N) // group.go:45-51: one length per group
        return fail(MUSE_ERR_LENGTH, "Timeseries has length %lld, but current group has length %d",
                    (long long)row_stride, g->N);
    if (g->win_rows)
        return fail(MUSE_ERR_INVALID, "the group has an open staging window");
    int rc = use_device(g->ctx);
    if (rc)
        return rc;
    const size_t row_bytes = (size_t)g->N * sizeof(double);
    // small appends are staged from the SECOND one on: a group that is uploaded in one call (Muse.Run builds one
    // per call) never needs the staging pair
    bool small = (size_t)count * row_bytes < STAGE_BYTES / 4 && row_bytes <= STAGE_BYTES;
    if (small && !g->stage[0] && g->small_appends++ == 0)
        small = false;
    if (g->f32) // float32 storage: every append is narrowed on the host into the pinned staging pair (half the PCIe bytes too)
        small = true;
    if (small) {
        rc = group_stage_buffers(g);
        if (rc)
            return rc;
    }
    if (!small) { // a slab: upload it directly (synchronously: the caller's memory is not retained)
        rc = group_flush(g);
        if (rc)
            return rc;
        rc = group_reserve(g, g->M + count);
        if (rc)
            return rc;
        // (on the copy stream: the caller's memory is not retained, so the call waits for the copy -- but not for a score
        // pass that may be running on the compute stream over the rows uploaded earlier)
        HIP_TRY(hipMemcpy2DAsync(g->rows + g->M * g->stride, (size_t)g->stride * sizeof(double), rows,
                                 (size_t)row_stride * sizeof(double), row_bytes, (size_t)count, hipMemcpyHostToDevice,
                                 g->ctx->copy_stream));
        HIP_TRY(hipStreamSynchronize(g->ctx->copy_stream));
        g->M += count;
        return MUSE_OK;
    }
    const int64_t flush_rows = std::max<int64_t>(1, (int64_t)(STAGE_FLUSH_BYTES / ((size_t)g->N * g->elem())));
    for (int64_t r = 0; r < count; r++) {
        if (g->M + 1 > g->cap) {
            // growing re-allocates and copies on the stream; staged rows are uploaded first
            rc = group_flush(g);
            if (!rc)
                rc = group_reserve(g, g->M + 1);
            if (rc)
                return rc;
        }
        if (g->f32) {
            float *dst = (float *)g->stage[g->cur] + g->staged * g->N;
            const double *src = rows + r * row_stride;
            for (int32_t j = 0; j < g->N; j++)
                dst[j] = (float)src[j];
        } else {
            memcpy(g->stage[g->cur] + g->staged * g->N, rows + r * row_stride, row_bytes);
        }
        g->staged++;
        g->M++;
        if (g->staged == g->stage_rows || g->staged - g->flushed >= flush_rows) {
            rc = group_flush(g);
            if (rc)
                return rc;
        }
    }
    return MUSE_OK;
}

// ---- the caller packs rows straight into pinned memory (no staging copy inside the library, any number of filling threads)
extern "C" int muse_group_stage(muse_group *g, int64_t count, double **window, int64_t *granted)
{
    if (!g || !window || !granted || count < 0)
        return fail(MUSE_ERR_INVALID, "bad staging arguments");
    *window = nullptr;
    *granted = 0;
    if (count == 0)
        return MUSE_OK;
    if (g->f32)
        return fail(MUSE_ERR_UNSUPPORTED, "staging windows hold float64 rows (float32-storage groups narrow inside muse_group_append)");
    if (g->win_rows)
        return fail(MUSE_ERR_INVALID, "the group already has an open staging window");
    int rc = use_device(g->ctx);
    if (rc)
        return rc;
    rc = group_stage_buffers(g);
    if (rc)
        return rc;
    rc = group_flush(g);
    if (rc)
        return rc;
    if (g->staged > 0) { // a window starts a buffer of its own
        g->staged = g->flushed = 0;
        g->cur ^= 1;
        HIP_TRY(hipEventSynchronize(g->stage_done[g->cur]));
    } else {
        HIP_TRY(hipEventSynchronize(g->stage_done[g->cur])); // (a window the previous call filled: its copies have landed)
    }
    const int64_t k = std::min<int64_t>(count, g->stage_rows);
    rc = group_reserve(g, g->M + k);
    if (rc)
        return rc;
    g->win_rows = k;
    g->win_committed = 0;
    *window = g->stage[g->cur];
    *granted = k;
    return MUSE_OK;
}

extern "C" int muse_group_commit(muse_group *g, int64_t first, int64_t count)
{
    if (!g || first < 0 || count < 0)
        return fail(MUSE_ERR_INVALID, "bad commit arguments");
    if (count == 0)
        return MUSE_OK;
    std::lock_guard<std::mutex> lock(g->win_mu); // (the filling threads commit their own pieces)
    if (!g->win_rows || first + count > g->win_rows || g->win_committed + count > g->win_rows)
        return fail(MUSE_ERR_INVALID, "commit of rows [%lld, %lld) outside the open window of %lld rows", (long long)first,
                    (long long)(first + count), (long long)g->win_rows);
    int rc = use_device(g->ctx);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(g->rows + (g->M + first) * g->stride, g->stage[g->cur] + first * g->N,
                           (size_t)count * (size_t)g->N * sizeof(double), hipMemcpyHostToDevice, g->ctx->copy_stream));
    g->win_committed += count;
    if (g->win_committed == g->win_rows) { // the window is complete: its rows join the group
        HIP_TRY(hipEventRecord(g->stage_done[g->cur], g->ctx->copy_stream));
        HIP_TRY(hipEventRecord(g->uploaded, g->ctx->copy_stream));
        g->upload_pending = true;
        g->M += g->win_rows;
        g->staged = g->flushed = g->win_rows;
        g->win_rows = g->win_committed = 0;
        if (g->staged == g->stage_rows) { // (as group_flush leaves a full buffer; the next user waits for its event)
            g->staged = g->flushed = 0;
            g->cur ^= 1;
        }
    }
    return MUSE_OK;
}

extern "C" int muse_group_upload(muse_ctx *ctx, const double *rows, int64_t M, int32_t N, int64_t row_stride,
                                 muse_group **out)
{
    int rc = muse_group_create(ctx, M, N, out);
    if (rc)
        return rc;
    rc = muse_group_append(*out, rows, M, row_stride);
    if (rc) {
        muse_group_free(*out);
        *out = nullptr;
    }
    return rc;
}

extern "C" int muse_group_fill_synthetic(muse_group *g, int64_t first, int64_t count, int64_t global_first,
                                         uint64_t seed, uint32_t flags, double *ref_out)
{
    if (!g || first < 0 || count < 0 || first > g->M)
        return fail(MUSE_ERR_INVALID, "bad synthetic fill range");
    int rc = use_device(g->ctx);
    if (rc)
        return rc;
    rc = group_ready(g);
    if (rc)
        return rc;
    rc = group_reserve(g, first + count);
    if (!rc)
        rc = group_ready(g); // (a growth copies the old rows over on the copy stream: the fill waits for it)
    if (rc)
        return rc;
    if (g->f32)
        HIP_TRY(launch_synth_f32(g->rows32, g->stride, first, count, global_first, g->N, seed, flags, g->ctx->stream));
    else
        HIP_TRY(launch_synth(g->rows, g->stride, first, count, global_first, g->N, seed, flags, g->ctx->stream));
    g->M = std::max(g->M, first + count);
    g->hstats_rows = std::min(g->hstats_rows, first); // (rows rewritten: their kept statistics go)
    g->rewrites++;
    if (ref_out) {
        double *d = nullptr;
        HIP_TRY(hipMalloc(&d, (size_t)g->N * sizeof(double)));
        hipError_t e = launch_synth_ref(d, g->N, seed, g->ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ref_out, d, (size_t)g->N * sizeof(double), hipMemcpyDeviceToHost, g->ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(g->ctx->stream);
        (void)hipFree(d);
        HIP_TRY(e);
    }
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    return MUSE_OK;
}

extern "C" int muse_group_shape(muse_group *g, int64_t *M, int32_t *N)
{
    if (!g)
        return fail(MUSE_ERR_INVALID, "NULL group");
    if (M)
        *M = g->M;
    if (N)
        *N = g->N;
    return MUSE_OK;
}

extern "C" int muse_group_read(muse_group *g, int64_t first, int64_t count, double *out)
{
    if (!g || !out || first < 0 || count < 0 || first + count > g->M)
        return fail(MUSE_ERR_INVALID, "bad read range");
    if (count == 0)
        return MUSE_OK;
    int rc = use_device(g->ctx);
    if (rc)
        return rc;
    rc = group_ready(g);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(g->ctx->copy_stream));
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    if (g->f32) { // widened exactly: the checker sees the values the kernels see
        std::vector<float> tmp((size_t)count * (size_t)g->N);
        HIP_TRY(hipMemcpy2D(tmp.data(), (size_t)g->N * sizeof(float), g->rows32 + first * g->stride,
                            (size_t)g->stride * sizeof(float), (size_t)g->N * sizeof(float), (size_t)count,
                            hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); i++)
            out[i] = (double)tmp[i];
        return MUSE_OK;
    }
    HIP_TRY(hipMemcpy2D(out, (size_t)g->N * sizeof(double), g->rows + first * g->stride,
                        (size_t)g->stride * sizeof(double), (size_t)g->N * sizeof(double), (size_t)count,
                        hipMemcpyDeviceToHost));
    return MUSE_OK;
}

void group_release(muse_group *g)
{
    if (!g || g->refs.fetch_sub(1) != 1)
        return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->copy_stream);
    (void)hipStreamSynchronize(g->ctx->stream);
    if (g->uploaded)
        (void)hipEventDestroy(g->uploaded);
    if (g->base())
        dfree(g->ctx, (char *)g->base() - GROUP_GUARD * g->elem());
    for (void *old : g->retired)
        dfree(g->ctx, old);
    dfree(g->ctx, g->hstats);
    for (int i = 0; i < 2; i++) {
        if (g->stage[i]) { // back to the context's pool (the stream is idle: no upload reads it any more)
            std::lock_guard<std::mutex> lock(g->ctx->stage_mu);
            g->ctx->stage_pool.push_back(g->stage[i]);
        }
        if (g->stage_done[i])
            (void)hipEventDestroy(g->stage_done[i]);
    }
    muse_ctx *ctx = g->ctx;
    delete g;
    ctx_release(ctx);
}

extern "C" int muse_group_free(muse_group *g)
{
    group_release(g);
    return MUSE_OK;
}
