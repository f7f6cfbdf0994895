// capi_context.hip -- errors, the context (streams, tables, timing), the measurement hooks
// Part of the implementation of the C ABI declared in include/muse_hip.h (capi_internal.h: the handles and the helpers the
// parts share).  Host-side orchestration only; there is no CPU compute fallback anywhere: without a gfx950 device every
// compute entry point returns MUSE_ERR_NO_DEVICE.
#include "capi_internal.h"

using namespace muse;


// ------------------------------------------------------------------ errors
thread_local std::string g_last_error;

int fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

extern "C" int muse_abi_version(void) { return MUSE_HIP_ABI_VERSION; }
extern "C" const char *muse_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *muse_status_string(int s)
{
    switch (s) {
    case MUSE_OK: return "ok";
    case MUSE_ERR_INVALID: return "invalid argument";
    case MUSE_ERR_LENGTH: return "series length mismatch";
    case MUSE_ERR_ZERO_STD: return "Invalid input query, Standard deviation of zero";
    case MUSE_ERR_NO_DEVICE: return "no usable gfx950 device";
    case MUSE_ERR_HIP: return "HIP runtime error";
    case MUSE_ERR_UNSUPPORTED: return "unsupported FFT length";
    case MUSE_ERR_NOMEM: return "out of memory";
    case MUSE_ERR_EMPTY: return "Reference series length must be greater than zero";
    default: return "unknown status";
    }
}

// xcorr.go:19-24
extern "C" int64_t muse_next_pow2(double val)
{
    if (val <= 0)
        return 0;
    return (int64_t)std::pow(2.0, std::ceil(std::log(val) / std::log(2.0)));
}

int use_device(muse_ctx *ctx)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    HIP_TRY(hipSetDevice(ctx->device));
    return MUSE_OK;
}

// ------------------------------------------------------------ allocation cache
// Size classes: powers of two up to 1 MB, multiples of 1 MB beyond (a group of M x N rows always asks for the same class).
static size_t pool_class(size_t bytes)
{
    if (bytes <= 256)
        return 256;
    if (bytes <= ((size_t)1 << 20)) {
        size_t c = 256;
        while (c < bytes)
            c <<= 1;
        return c;
    }
    return (bytes + (((size_t)1 << 20) - 1)) & ~(((size_t)1 << 20) - 1);
}

hipError_t pool_alloc(muse_ctx *ctx, bool host, void **out, size_t bytes)
{
    MemPool &mp = host ? ctx->host_pool : ctx->dev_pool;
    const size_t cls = pool_class(bytes);
    *out = nullptr;
    {
        std::lock_guard<std::mutex> lock(mp.mu);
        auto it = mp.idle.find(cls);
        if (it != mp.idle.end()) {
            *out = it->second;
            mp.idle.erase(it);
            mp.idle_bytes -= cls;
            return hipSuccess;
        }
    }
    void *p = nullptr;
    hipError_t e = host ? hipHostMalloc(&p, cls, hipHostMallocDefault) : hipMalloc(&p, cls);
    if (e != hipSuccess) { // out of memory with blocks cached: give them back and try once more
        (void)hipGetLastError();
        pool_drain(ctx);
        e = host ? hipHostMalloc(&p, cls, hipHostMallocDefault) : hipMalloc(&p, cls);
    }
    if (e != hipSuccess)
        return e;
    {
        std::lock_guard<std::mutex> lock(mp.mu);
        mp.size_of[p] = cls;
    }
    *out = p;
    return hipSuccess;
}

void pool_free(muse_ctx *ctx, bool host, void *p)
{
    if (!p)
        return;
    MemPool &mp = host ? ctx->host_pool : ctx->dev_pool;
    {
        std::lock_guard<std::mutex> lock(mp.mu);
        auto it = mp.size_of.find(p);
        if (it != mp.size_of.end()) {
            const size_t cls = it->second;
            if (cls <= mp.block_cap && mp.idle_bytes + cls <= mp.idle_cap) {
                mp.idle.emplace(cls, p);
                mp.idle_bytes += cls;
                return;
            }
            mp.size_of.erase(it);
        }
    }
    // (not one of the pool's blocks, or the cache is full)
    if (host)
        (void)hipHostFree(p);
    else
        (void)hipFree(p);
}

void pool_drain(muse_ctx *ctx)
{
    for (int h = 0; h < 2; h++) {
        MemPool &mp = h ? ctx->host_pool : ctx->dev_pool;
        std::vector<void *> blocks;
        {
            std::lock_guard<std::mutex> lock(mp.mu);
            for (auto &kv : mp.idle) {
                blocks.push_back(kv.second);
                mp.size_of.erase(kv.second);
            }
            mp.idle.clear();
            mp.idle_bytes = 0;
        }
        for (void *b : blocks) {
            if (h)
                (void)hipHostFree(b);
            else
                (void)hipFree(b);
        }
    }
}

// ----------------------------------------------------------------- context
void fill_twiddle(std::vector<double2> &v, size_t i, long long num, long long den)
{
    const long double PI2 = 6.283185307179586476925286766559005768L;
    num %= den;
    const long double a = -PI2 * (long double)num / (long double)den;
    v[i] = make_double2((double)cosl(a), (double)sinl(a));
}

extern "C" int muse_ctx_create(int32_t device, muse_ctx **out)
{
    if (!out)
        return fail(MUSE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(MUSE_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    }
    if (device < 0 || device >= count)
        return fail(MUSE_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MUSE_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    muse_ctx *ctx = new (std::nothrow) muse_ctx();
    if (!ctx)
        return fail(MUSE_ERR_NOMEM, "host allocation failed");
    ctx->device = device;
    // what an idle context may keep cached: 1 GB of HBM in blocks of up to 256 MB, 192 MB of pinned host memory in blocks of up to 64 MB
    ctx->dev_pool.idle_cap = (size_t)1 << 30;
    ctx->dev_pool.block_cap = (size_t)256 << 20;
    ctx->host_pool.idle_cap = (size_t)192 << 20;
    ctx->host_pool.block_cap = (size_t)64 << 20;
    ctx->num_cus = prop.multiProcessorCount;
    ctx->hbm = (int64_t)prop.totalGlobalMem;
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    if (hipDeviceGetPCIBusId(ctx->pci, (int)sizeof(ctx->pci), device) != hipSuccess)
        snprintf(ctx->pci, sizeof(ctx->pci), "%04x:%02x:%02x.0", prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    std::vector<double2> t1(16 * 256), t2(16 * 16), tm(GENERIC_MAX_N / 2);
    for (int k = 0; k < 16; k++)
        for (int t = 0; t < 256; t++)
            fill_twiddle(t1, (size_t)k * 256 + t, (long long)k * t, 4096);
    for (int k = 0; k < 16; k++)
        for (int c = 0; c < 16; c++)
            fill_twiddle(t2, (size_t)k * 16 + c, (long long)k * c, 256);
    for (int k = 0; k < GENERIC_MAX_N / 2; k++)
        fill_twiddle(tm, (size_t)k, k, GENERIC_MAX_N);
    HIP_TRY(hipMalloc(&ctx->tw1, t1.size() * sizeof(double2)));
    HIP_TRY(hipMalloc(&ctx->tw2, t2.size() * sizeof(double2)));
    HIP_TRY(hipMalloc(&ctx->twm, tm.size() * sizeof(double2)));
    HIP_TRY(hipMemcpy(ctx->tw1, t1.data(), t1.size() * sizeof(double2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->tw2, t2.data(), t2.size() * sizeof(double2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->twm, tm.data(), tm.size() * sizeof(double2), hipMemcpyHostToDevice));
    {
        std::vector<float2> tmf(tm.size());
        for (size_t k = 0; k < tm.size(); k++)
            tmf[k] = make_float2((float)tm[k].x, (float)tm[k].y);
        HIP_TRY(hipMalloc(&ctx->twmf, tmf.size() * sizeof(float2)));
        HIP_TRY(hipMemcpy(ctx->twmf, tmf.data(), tmf.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    {   // generalised-pass factors for delta = u / 256 (fold_device.h): W_512^u, W_1024^u, W_2048^u, W_2048^(u+256), W_4096^(u+256q)
        const auto fill_g = [](std::vector<double2> &g, size_t stride, size_t idx, long long u) {
            fill_twiddle(g, 0 * stride + idx, u, 512);
            fill_twiddle(g, 1 * stride + idx, u, 1024);
            fill_twiddle(g, 2 * stride + idx, u, 2048);
            fill_twiddle(g, 3 * stride + idx, u + 256, 2048);
            for (int q = 0; q < 4; q++)
                fill_twiddle(g, (size_t)(4 + q) * stride + idx, u + 256 * q, 4096);
        };
        std::vector<double2> g2(8 * 16), g3a(8 * 256), g3b(8 * 256);
        for (int j = 0; j < 16; j++)
            fill_g(g2, 16, (size_t)j, 16 * j);
        for (int t = 0; t < 256; t++) {
            fill_g(g3a, 256, (size_t)t, (t >> 4) + 16 * (t & 15));
            fill_g(g3b, 256, (size_t)t, t);
        }
        HIP_TRY(hipMalloc(&ctx->g2, g2.size() * sizeof(double2)));
        HIP_TRY(hipMalloc(&ctx->g3a, g3a.size() * sizeof(double2)));
        HIP_TRY(hipMalloc(&ctx->g3b, g3b.size() * sizeof(double2)));
        HIP_TRY(hipMemcpy(ctx->g2, g2.data(), g2.size() * sizeof(double2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->g3a, g3a.data(), g3a.size() * sizeof(double2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->g3b, g3b.data(), g3b.size() * sizeof(double2), hipMemcpyHostToDevice));
        // xcorr_small.hip's passes behind the second one: phase m / L, m = j mod L, for L = 16 R1 (and L = 256 R1 = S, n = 8192):
        // W_(2L)^m, W_(4L)^m, W_(8L)^m, W_(8L)^(m+L), W_(16L)^(m+qL), lane-ordered
        for (int k = 0; k < 5; k++) {
            const int n = k < 3 ? (512 << k) : (2048 << (k - 1)), S = n / 16;
            std::vector<double2> gs;
            for (int L = (k < 3 ? S : S / 16); L <= S; L *= 16) {
                const size_t o = gs.size();
                gs.resize(o + (size_t)8 * L);
                for (int m = 0; m < L; m++) {
                    fill_twiddle(gs, o + (size_t)0 * L + m, m, 2 * L);
                    fill_twiddle(gs, o + (size_t)1 * L + m, m, 4 * L);
                    fill_twiddle(gs, o + (size_t)2 * L + m, m, 8 * L);
                    fill_twiddle(gs, o + (size_t)3 * L + m, m + L, 8 * L);
                    for (int q = 0; q < 4; q++)
                        fill_twiddle(gs, o + (size_t)(4 + q) * L + m, m + q * L, 16 * L);
                }
            }
            HIP_TRY(hipMalloc(&ctx->gsmall[k], gs.size() * sizeof(double2)));
            HIP_TRY(hipMemcpy(ctx->gsmall[k], gs.data(), gs.size() * sizeof(double2), hipMemcpyHostToDevice));
        }
    }
    {   // xcorr_real.hip's 16 x 1024 split of the 16384-point transform: the twiddles W_16384^(j k1) behind the register pass
        std::vector<double2> ws((size_t)15 * 1024);
        for (int k1 = 1; k1 < 16; k1++)
            for (int j = 0; j < 1024; j++)
                fill_twiddle(ws, (size_t)(k1 - 1) * 1024 + j, (long long)j * k1, 16384);
        HIP_TRY(hipMalloc(&ctx->wsplit, ws.size() * sizeof(double2)));
        HIP_TRY(hipMemcpy(ctx->wsplit, ws.data(), ws.size() * sizeof(double2), hipMemcpyHostToDevice));
    }
    std::vector<float2> t1f(t1.size()), t2f(t2.size());
    for (size_t i = 0; i < t1.size(); i++)
        t1f[i] = make_float2((float)t1[i].x, (float)t1[i].y);
    for (size_t i = 0; i < t2.size(); i++)
        t2f[i] = make_float2((float)t2[i].x, (float)t2[i].y);
    HIP_TRY(hipMalloc(&ctx->tw1f, t1f.size() * sizeof(float2)));
    HIP_TRY(hipMalloc(&ctx->tw2f, t2f.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(ctx->tw1f, t1f.data(), t1f.size() * sizeof(float2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->tw2f, t2f.data(), t2f.size() * sizeof(float2), hipMemcpyHostToDevice));
    *out = ctx;
    return MUSE_OK;
}

extern "C" int muse_device_count(int32_t *count)
{
    if (!count)
        return fail(MUSE_ERR_INVALID, "count is NULL");
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MUSE_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    }
    int usable = 0;
    for (int d = 0; d < n; d++) { // device ordinals are HIP's: count the leading run of gfx950 devices
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            break;
        usable++;
    }
    if (!usable)
        return fail(MUSE_ERR_NO_DEVICE, "no gfx950 device visible");
    *count = usable;
    return MUSE_OK;
}

void ctx_release(muse_ctx *ctx)
{
    if (!ctx || ctx->refs.fetch_sub(1) != 1)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
#ifdef MUSE_REAL64_STAMPS
    if (ctx->dbg_stamps && getenv("MUSE_STAMPS_OUT")) {
        std::vector<unsigned long long> h((size_t)ctx->num_cus * 4 * 16 * 16);
        if (hipMemcpy(h.data(), ctx->dbg_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(getenv("MUSE_STAMPS_OUT"), "w")) {
                for (int wg = 0; wg < ctx->num_cus * 4; wg++)
                    for (int w = 0; w < 16; w++) {
                        fprintf(f, "%d %d", wg, w);
                        for (int i = 0; i < 16; i++)
                            fprintf(f, " %llu", h[((size_t)wg * 16 + w) * 16 + i]);
                        fprintf(f, "\n");
                    }
                fclose(f);
            }
        }
    }
#endif
    (void)hipFree(ctx->dbg_stamps);
    rows_slots_free(ctx);
    huge_free(ctx);
    for (auto &p : ctx->small_free)
        (void)hipHostFree(p.first);
    ctx->small_free.clear();
    pool_drain(ctx);
    for (auto *ev : {&ctx->events, &ctx->redo_events})
        for (auto &e : *ev) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    (void)hipFree(ctx->tw1);
    (void)hipFree(ctx->tw2);
    (void)hipFree(ctx->twm);
    (void)hipFree(ctx->twmf);
    (void)hipFree(ctx->tw1f);
    (void)hipFree(ctx->g2);
    (void)hipFree(ctx->g3a);
    (void)hipFree(ctx->g3b);
    for (int k = 0; k < 5; k++)
        (void)hipFree(ctx->gsmall[k]);
    (void)hipFree(ctx->wsplit);
    for (int k = 0; k < 3; k++)
        (void)hipFree(ctx->twl[k]);
    (void)hipFree(ctx->zscratch);
    (void)hipFree(ctx->gscratch);
    for (double *b : ctx->stage_pool)
        (void)hipHostFree(b);
    (void)hipFree(ctx->many_tab);
    (void)hipFree(ctx->tw2f);
    if (ctx->probe_stream) {
        if (ctx->probe_buf) // (a probe still running ends within microseconds of its stop flag)
            *((volatile int *)(ctx->probe_buf + 2 * PROBE_WINDOWS) + 1) = 1;
        (void)hipStreamSynchronize(ctx->probe_stream);
        (void)hipStreamDestroy(ctx->probe_stream);
    }
    if (ctx->probe_buf)
        (void)hipHostFree(ctx->probe_buf);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int muse_ctx_destroy(muse_ctx *ctx)
{
    ctx_release(ctx);
    return MUSE_OK;
}

extern "C" int muse_ctx_synchronize(muse_ctx *ctx)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MUSE_OK;
}

extern "C" int muse_ctx_trim(muse_ctx *ctx)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    pool_drain(ctx);
    std::vector<std::pair<unsigned char *, int>> slots;
    {
        std::lock_guard<std::mutex> lock(ctx->small_mu); // (the record buffers of small Runs no batch holds at the moment)
        slots.swap(ctx->small_free);
    }
    for (auto &p : slots)
        (void)hipHostFree(p.first);
    return MUSE_OK;
}

extern "C" int muse_test_pool_stats(muse_ctx *ctx, int64_t *dev_idle_bytes, int64_t *dev_idle_blocks, int64_t *host_idle_bytes,
                                    int64_t *host_idle_blocks)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    {
        std::lock_guard<std::mutex> lock(ctx->dev_pool.mu);
        if (dev_idle_bytes)
            *dev_idle_bytes = (int64_t)ctx->dev_pool.idle_bytes;
        if (dev_idle_blocks)
            *dev_idle_blocks = (int64_t)ctx->dev_pool.idle.size();
    }
    {
        std::lock_guard<std::mutex> lock(ctx->host_pool.mu);
        if (host_idle_bytes)
            *host_idle_bytes = (int64_t)ctx->host_pool.idle_bytes;
        if (host_idle_blocks)
            *host_idle_blocks = (int64_t)ctx->host_pool.idle.size();
    }
    return MUSE_OK;
}

extern "C" int muse_test_xcorr_repeat(muse_ctx *ctx, int32_t repeat)
{
    if (!ctx || repeat < 1 || repeat > 1000)
        return fail(MUSE_ERR_INVALID, "repeat must be 1 .. 1000");
    ctx->xcorr_repeat = repeat;
    return MUSE_OK;
}

extern "C" int muse_ctx_device_info(muse_ctx *ctx, char *name, int32_t name_cap, int32_t *cus, int64_t *hbm)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    if (name && name_cap > 0)
        snprintf(name, (size_t)name_cap, "%s", ctx->name);
    if (cus)
        *cus = ctx->num_cus;
    if (hbm)
        *hbm = ctx->hbm;
    return MUSE_OK;
}

extern "C" int muse_ctx_set_kernel(muse_ctx *ctx, int32_t variant)
{
    if (!ctx || !(variant == 0 || variant == 1 || variant == 7 || variant == 10 || variant == 11 || variant == 12 || variant == 13 || variant == 14 || variant == 15))
        return fail(MUSE_ERR_INVALID, "bad kernel variant (0 auto, 1 generic, 7 rescaling n=4096, 10 default n=4096, 11 Stockham, 12 half-round, 13 long series, 14 real transform, 15 real transform on the 16 x 1024 split)");
    ctx->variant = variant;
    return MUSE_OK;
}

extern "C" int muse_ctx_set_screening(muse_ctx *ctx, int32_t enable)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    ctx->screening = enable != 0;
    ctx->screen_min_rows = enable > 1 ? enable : 0;
    return MUSE_OK;
}

extern "C" int muse_ctx_kernel_timing(muse_ctx *ctx, int32_t enable)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    ctx->timing = enable != 0;
    return MUSE_OK;
}

static int drain_events(std::vector<std::pair<hipEvent_t, hipEvent_t>> &ev, double &ms_sum, int64_t &count)
{
    for (auto &e : ev) {
        HIP_TRY(hipEventSynchronize(e.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
        ms_sum += (double)ms;
        count += 1;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    ev.clear();
    return MUSE_OK;
}

extern "C" int muse_ctx_kernel_time(muse_ctx *ctx, double *total_ms, int64_t *launches)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    rc = drain_events(ctx->events, ctx->total_ms, ctx->launches);
    if (rc)
        return rc;
    if (total_ms)
        *total_ms = ctx->total_ms;
    if (launches)
        *launches = ctx->launches;
    ctx->total_ms = 0.0;
    ctx->launches = 0;
    return MUSE_OK;
}

extern "C" int muse_ctx_redo_time(muse_ctx *ctx, double *total_ms, int64_t *brackets)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    rc = drain_events(ctx->redo_events, ctx->redo_ms, ctx->redo_launches);
    if (rc)
        return rc;
    if (total_ms)
        *total_ms = ctx->redo_ms;
    if (brackets)
        *brackets = ctx->redo_launches;
    ctx->redo_ms = 0.0;
    ctx->redo_launches = 0;
    return MUSE_OK;
}

extern "C" int muse_ctx_device_pci_bus_id(muse_ctx *ctx, char *out, int32_t cap)
{
    if (!ctx || !out || cap < 16)
        return fail(MUSE_ERR_INVALID, "muse_ctx_device_pci_bus_id: NULL argument or a buffer under 16 bytes");
    snprintf(out, (size_t)cap, "%s", ctx->pci);
    return MUSE_OK;
}

// ---- measurement hook: the shader clock held while other kernels of the process run (diag_kernels.hip)
// ends a running probe early (host flag in the pinned buffer: no GPU call)
extern "C" int muse_test_clock_probe_stop(muse_ctx *ctx)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    if (ctx->probe_buf)
        *((volatile int *)(ctx->probe_buf + 2 * PROBE_WINDOWS) + 1) = 1;
    return MUSE_OK;
}

extern "C" int muse_test_wave_argmax(muse_ctx *ctx, const double *ccA, const double *ccB, double *out24)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!ccA || !ccB || !out24)
        return fail(MUSE_ERR_INVALID, "wave argmax probe: null pointer");
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (2 * 4096 + 24) * sizeof(double)));
    hipError_t e = hipMemcpy(d, ccA, 4096 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d + 4096, ccB, 4096 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launch_wave_argmax_probe(d, d + 4096, d + 8192, nullptr);
    if (e == hipSuccess)
        e = hipMemcpy(out24, d + 8192, 24 * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess)
        return fail(MUSE_ERR_HIP, hipGetErrorString(e));
    return MUSE_OK;
}

extern "C" int muse_test_clock_probe_start(muse_ctx *ctx, double window_ms, double total_ms)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!(window_ms >= 0.05) || !(total_ms >= window_ms) || total_ms > 60000.0)
        return fail(MUSE_ERR_INVALID, "clock probe: window >= 0.05 ms, window <= total <= 60 s");
    if (!ctx->probe_stream)
        HIP_TRY(hipStreamCreateWithFlags(&ctx->probe_stream, hipStreamNonBlocking));
    if (!ctx->probe_buf) // (pinned and device-visible: the probe writes it directly, no copy behind a kernel that is still running)
        // (coherent: the host's stop flag must reach a kernel that is already running, and the kernel's window count the host,
        // whatever HIP_HOST_COHERENT says)
        HIP_TRY(hipHostMalloc((void **)&ctx->probe_buf, (2 * PROBE_WINDOWS + 1) * sizeof(unsigned long long),
                              hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipStreamSynchronize(ctx->probe_stream));
    memset(ctx->probe_buf, 0, (2 * PROBE_WINDOWS + 1) * sizeof(unsigned long long));
    HIP_TRY(launch_clock_probe(ctx->probe_buf, (int *)(ctx->probe_buf + 2 * PROBE_WINDOWS), PROBE_WINDOWS, window_ms, total_ms,
                               ctx->probe_stream));
    // return once the probe is RESIDENT (its first window has landed in the pinned buffer): launched behind a grid that fills
    // the chip it would only start when that grid has drained, and sample an idle GPU
    volatile int *cnt = (volatile int *)(ctx->probe_buf + 2 * PROBE_WINDOWS);
    for (int spin = 0; *cnt == 0 && spin < 20000; spin++) { // <= ~2 s
        struct timespec ts = {0, 100000};
        nanosleep(&ts, nullptr);
    }
    if (*cnt == 0) {
        *(cnt + 1) = 1; // the stop flag: a probe that becomes resident later ends at once instead of sampling for total_ms
        return fail(MUSE_ERR_HIP, "clock probe did not start");
    }
    return MUSE_OK;
}

// waits for the probe; mhz[] (capacity cap) receives the clock of every window in order, *windows their number
extern "C" int muse_test_clock_probe_read(muse_ctx *ctx, double *mhz, int32_t cap, int32_t *windows)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!ctx->probe_stream || !ctx->probe_buf || !windows)
        return fail(MUSE_ERR_INVALID, "clock probe was not started");
    HIP_TRY(hipStreamSynchronize(ctx->probe_stream));
    const int n = *(const int *)(ctx->probe_buf + 2 * PROBE_WINDOWS);
    *windows = n;
    for (int w = 0; w < n && w < cap && mhz; w++) {
        const double ticks = (double)ctx->probe_buf[2 * w], real = (double)ctx->probe_buf[2 * w + 1];
        mhz[w] = real > 0.0 ? ticks / real * 100.0 : 0.0;
    }
    return MUSE_OK;
}
