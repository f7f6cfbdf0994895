// muse_capi.hip -- implementation of the C ABI declared in include/muse_hip.h.
// Host-side orchestration only: device memory, streams, launches, and the
// final (tiny) top-N heap that mirrors go-muse's Results (results.go:55-87).
// There is no CPU compute fallback anywhere in this file: without a gfx950
// device every compute entry point returns MUSE_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "muse_hip.h"
#include "muse_hip_test.h"
#include "xcorr_kernels.h"

using namespace muse;

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

static int fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(MUSE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                        __FILE__, __LINE__);                                                       \
    } while (0)

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

// ----------------------------------------------------------------- handles
constexpr int PROBE_WINDOWS = 4096; // clock probe (muse_test_clock_probe_*): windows its pinned buffer holds; the window count and the stop flag sit behind them
struct muse_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // every kernel of the context
    hipStream_t copy_stream = nullptr; // host -> HBM uploads of muse_group_append: run beside a score pass
    int num_cus = 0;
    int64_t hbm = 0;
    char name[64] = {0};
    double2 *tw1 = nullptr, *tw2 = nullptr, *twm = nullptr;
    double2 *g2 = nullptr, *g3a = nullptr, *g3b = nullptr; // folded-twiddle tables (xcorr_r16_fold.hip)
    double2 *twl[3] = {nullptr, nullptr, nullptr};          // xcorr_long.hip (n = 16384, 32768, 65536): [4096] W_n^(m2), built on first use
    double2 *gsmall[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // the same for xcorr_small.hip: n = 512, 1024, 2048: [8][n/16]; n = 8192: [8][32] + [8][512]; n = 16384: [8][64] + [8][1024]
    float2 *tw1f = nullptr, *tw2f = nullptr, *twmf = nullptr; // fp32 copies for the screening kernels
    // many-reference pass (muse_batch_score_many): parked spectra + device pointer tables
    // pinned staging buffers (32 MB each) lent to groups that receive many small appends; allocated once
    // (hipHostMalloc of 32 MB costs milliseconds) and returned when the group is released
    // work buffers of the generic / Stockham kernels for n >= 8192: one allocation per context, grown on demand
    // (every kernel that uses it runs on the context's single stream)
    double2 *gscratch = nullptr;
    size_t gscratch_elems = 0;
    std::vector<double *> stage_pool;
    std::mutex stage_mu;
    double2 *zscratch = nullptr;
    int zslots = 0;
    void *many_tab = nullptr; // R x {xcp, mv, lag} pointers
    std::vector<void *> many_host; // host image of many_tab (outlives the asynchronous copy)
    int many_cap = 0;
    double screen_delta = 1e-4;
    // filter-and-refine Run (run_select), OPT-IN (muse_ctx_set_screening): 1 = Runs over large groups screen in fp32 and
    // re-evaluate in fp64 only the rows that can reach the top-N; 0 (default) = every Run scores all rows in fp64, the
    // arithmetic of the reference (xcorr.go:160-197)
    int screening = 0;
    // smaller groups: the plain fp64 pass is as fast (tools/screen_crossover.py: the crossover is at ~25 000 rows of 4096
    // samples).  Default: M * n >= 32768 * 4096 samples; an explicit row count (muse_ctx_set_screening(ctx, rows)) overrides.
    int64_t screen_min_rows = 0;
    double screen_e_scale = 1.0;     // test hook (muse_test_set_screen_bound_scale): scales the error bound, to exercise the guard
    int variant = 0;
    // measurement hook (muse_test_clock_probe_*): a one-wave kernel on its own stream sampling the shader clock
    hipStream_t probe_stream = nullptr;
    unsigned long long *probe_buf = nullptr; // pinned host memory: [2 * PROBE_WINDOWS] ticks + the window count behind them
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events, redo_events;
    double total_ms = 0.0, redo_ms = 0.0;
    int64_t launches = 0, redo_launches = 0;
    char pci[32] = {0}; // PCI bus id of the device ("0000:05:00.0"): tells two contexts on one GPU from two GPUs
    // Handles may be released in any order (Go finalizers, Python GC): the
    // context lives until it is destroyed AND its last group/batch is freed.
    std::atomic<int> refs{1};
};

struct muse_group {
    std::atomic<int> refs{1}; // the handle itself + one per batch built on it
    muse_ctx *ctx = nullptr;
    double *rows = nullptr; // float64 storage (the default: what the reference holds)
    float *rows32 = nullptr; // float32 storage (muse_group_create_f32, opt-in): exactly one of the two is used
    bool f32 = false;
    size_t elem() const { return f32 ? sizeof(float) : sizeof(double); }
    void *base() const { return f32 ? (void *)rows32 : (void *)rows; }
    int64_t cap = 0, M = 0, stride = 0; // M counts staged rows too
    int32_t N = 0;
    // Small appends (Group.Add calls muse_group_append once per Series) are packed into
    // two pinned staging buffers and uploaded asynchronously on the context's stream, one
    // buffer in flight while the other fills; kernels on that stream are ordered behind.
    double *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    int cur = 0;
    int64_t staged = 0;     // rows waiting in stage[cur]
    int64_t stage_rows = 0; // capacity of one staging buffer, in rows
    int small_appends = 0;  // the first small append goes straight to the device (Muse.Run: one upload per group)
    // uploads run on the context's copy stream; `uploaded` is recorded behind the last one enqueued and the compute stream
    // waits for it (hipStreamWaitEvent) before a kernel reads the rows: an append of NEW rows overlaps a running score pass
    hipEvent_t uploaded = nullptr;
    bool upload_pending = false;
};

// The reference spectrum and the tables derived from it: shared (reference-counted) by the batches
// created with muse_batch_create_like -- Muse.Run builds one small group per call against ONE reference.
struct muse_spectrum {
    std::atomic<int> refs{1};
    double2 *X = nullptr, *xc = nullptr, *xcp = nullptr;
    float2 *xcf = nullptr;
    double *xs = nullptr;
    double *c1 = nullptr; // n == 4096, N < 4096: indicator correlation (xcorr_r16_fast.hip, PADDED)
    double xmax = -1.0;   // max |X[f]| (lazily, by the first screened Run): scales the fp32 error bound
};

struct muse_batch {
    muse_ctx *ctx = nullptr;
    muse_group *g = nullptr;
    int32_t N = 0, n = 0, logn = 0;
    muse_spectrum *sp = nullptr; // owner of the five tables below (the pointers are copies)
    double *c1 = nullptr;
    double2 *X = nullptr, *xc = nullptr;
    double2 *xcp = nullptr; // n == 4096: xc in the lane order of xcorr_r16_fast.hip
    float2 *xcf = nullptr; // fp32 conj(X)/n (screening kernel)
    double *xs = nullptr;  // padded time-domain reference (exact re-evaluation)
    int *ovf_count = nullptr;
    // automatic kernel selection learns from the previous pass over the same (immutable) rows: the number of
    // pairs the default N = 4096 kernel handed to the rescaling kernel lands here (pinned, asynchronous copy)
    int *handoff_host = nullptr;
    int64_t handoff_M = -1;
    long long *ovf_list = nullptr;
    int64_t ovf_cap = 0;
    double *mv = nullptr;
    int *lag = nullptr;
    int64_t score_cap = 0;
    // selection workspace
    int *gid_dev = nullptr;
    int64_t gid_cap = 0;
    std::vector<int32_t> gid_host;
    bool gid_valid = false;
    GroupWork gw{nullptr, nullptr, nullptr};
    muse_record *rec = nullptr;
    unsigned long long *selkey = nullptr;
    int64_t grp_cap = 0;
    muse_record *cand = nullptr;
    int *cnt = nullptr;
    int64_t cand_cap = 0, cnt_cap = 0;
    // pinned host images of cand / cnt: the device top-N pre-selection comes back in two truly asynchronous
    // copies and one synchronisation
    muse_record *cand_host = nullptr;
    int *cnt_host = nullptr;
    int64_t cand_host_cap = 0, cnt_host_cap = 0;
    // filter-and-refine Run
    unsigned *scr_flags = nullptr;      // [M] SCR_* bits of the screening pass
    double *scr_var = nullptr;          // [M] sample variances from the screening pass
    unsigned char *include = nullptr;   // [M] rows re-evaluated in fp64 (the only ones the selection may take)
    unsigned long long *scr_keys = nullptr;
    unsigned long long *scr_gmay = nullptr, *scr_gkplus = nullptr; // label groups: per-group bounds
    int *scr_gcert = nullptr;
    int64_t scr_gcap = 0;
    int64_t scr_cap = 0, scr_keys_cap = 0;
    int *refine_host = nullptr;         // pinned: pairs re-evaluated by the last screened Run
    double *est_save = nullptr;         // estimates of the listed rows (2 per pair), for the guard of the bound
    int64_t est_cap = 0;
    unsigned long long *err_dev = nullptr, *err_host = nullptr; // largest | |estimate| - |fp64 score| | of the last screened Run
    double last_E = 0.0;                // the bound that Run assumed
    // a screened Run with these filters over this many rows re-evaluated too many of them: the same Run is not
    // screened again (other filters on the same batch still are); a tripped guard switches the batch off for good
    struct RunKey {
        int64_t M = -1, G = 0;
        int32_t max_lag = 0, top_n = 0, sign_filter = 0, abs_scores = 0, grouped = 0;
        double threshold = 0.0;
        bool operator==(const RunKey &o) const
        {
            return M == o.M && G == o.G && max_lag == o.max_lag && top_n == o.top_n && sign_filter == o.sign_filter &&
                   abs_scores == o.abs_scores && grouped == o.grouped && threshold == o.threshold;
        }
    };
    RunKey costly_key;                  // (M = -1: none)
    bool guard_off = false;
    int32_t last_path = 0;              // MUSE_RUN_PATH_* of the last Run
    bool scores_exact = true;           // mv / lag hold fp64 results for every row (false after a screened Run)
    bool last_screened = false;         // the last Run took the filter-and-refine path
    int64_t guard_trips = 0;            // Runs redone in fp64 because an estimate left its bound
    uint64_t guard_salt = 0;            // varies the guard's row sample from Run to Run
};

static int use_device(muse_ctx *ctx)
{
    if (!ctx)
        return fail(MUSE_ERR_INVALID, "NULL context");
    HIP_TRY(hipSetDevice(ctx->device));
    return MUSE_OK;
}

// HIP-event bracket of ONE kernel launch on the context's stream (muse_ctx_kernel_timing): begin() right in front of the
// launch, end() right behind it; a bracket that never reaches end() (an error return in between) destroys its events.
struct LaunchTimer {
    muse_ctx *ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool redo; // the bracket of the launches that redo listed pairs behind a fused launch (muse_ctx_redo_time)
    explicit LaunchTimer(muse_ctx *c, bool redo_ = false) : ctx(c), redo(redo_) {}
    LaunchTimer(const LaunchTimer &) = delete;
    LaunchTimer &operator=(const LaunchTimer &) = delete;
    hipError_t begin()
    {
        if (!ctx->timing)
            return hipSuccess;
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess)
            e = hipEventCreate(&e1);
        if (e == hipSuccess)
            e = hipEventRecord(e0, ctx->stream);
        return e;
    }
    hipError_t end()
    {
        if (!e0 || !e1)
            return hipSuccess;
        const hipError_t e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) {
            (redo ? ctx->redo_events : ctx->events).emplace_back(e0, e1);
            e0 = e1 = nullptr;
        }
        return e;
    }
    ~LaunchTimer()
    {
        if (e0)
            (void)hipEventDestroy(e0);
        if (e1)
            (void)hipEventDestroy(e1);
    }
};

// ----------------------------------------------------------------- context
static void fill_twiddle(std::vector<double2> &v, size_t i, long long num, long long den)
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

static void ctx_release(muse_ctx *ctx)
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
    if (!ctx || !(variant == 0 || variant == 1 || variant == 7 || variant == 10 || variant == 11 || variant == 12 || variant == 13))
        return fail(MUSE_ERR_INVALID, "bad kernel variant (0 auto, 1 generic, 7 rescaling n=4096, 10 default n=4096, 11 Stockham, 12 half-round, 13 long series)");
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
        void *mem = nullptr;
        hipError_t e = hipMalloc(&mem, ((size_t)capacity_rows * (size_t)N + GROUP_GUARD) * g->elem());
        if (e == hipSuccess)
            e = hipMemset(mem, 0, GROUP_GUARD * g->elem());
        if (e != hipSuccess) {
            (void)hipFree(mem);
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

static int group_reserve(muse_group *g, int64_t rows)
{
    if (rows <= g->cap)
        return MUSE_OK;
    int64_t ncap = std::max<int64_t>(rows, g->cap * 2);
    void *nr = nullptr;
    hipError_t e = hipMalloc(&nr, ((size_t)ncap * (size_t)g->N + GROUP_GUARD) * g->elem());
    if (e == hipSuccess)
        e = hipMemset(nr, 0, GROUP_GUARD * g->elem());
    if (e != hipSuccess) {
        (void)hipFree(nr);
        return fail(MUSE_ERR_NOMEM, "hipMalloc of %lld rows failed: %s", (long long)ncap, hipGetErrorString(e));
    }
    nr = (char *)nr + GROUP_GUARD * g->elem();
    HIP_TRY(hipStreamSynchronize(g->ctx->copy_stream)); // uploads into the old allocation have landed
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));      // no kernel is still reading it
    if (g->M > 0) {
        HIP_TRY(hipMemcpyAsync(nr, g->base(), (size_t)g->M * (size_t)g->N * g->elem(), hipMemcpyDeviceToDevice,
                               g->ctx->stream));
        HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    }
    if (g->base())
        (void)hipFree((char *)g->base() - GROUP_GUARD * g->elem());
    (g->f32 ? (void *&)g->rows32 : (void *&)g->rows) = nr;
    g->cap = ncap;
    return MUSE_OK;
}

// enqueue the staged rows' upload (asynchronous); the buffer is reusable after stage_done
static int group_flush(muse_group *g)
{
    if (!g->staged)
        return MUSE_OK;
    const int64_t first = g->M - g->staged;
    HIP_TRY(hipMemcpyAsync((char *)g->base() + (size_t)(first * g->stride) * g->elem(), g->stage[g->cur],
                           (size_t)g->staged * (size_t)g->N * g->elem(), hipMemcpyHostToDevice, g->ctx->copy_stream));
    HIP_TRY(hipEventRecord(g->stage_done[g->cur], g->ctx->copy_stream));
    HIP_TRY(hipEventRecord(g->uploaded, g->ctx->copy_stream));
    g->upload_pending = true;
    g->staged = 0;
    g->cur ^= 1;
    HIP_TRY(hipEventSynchronize(g->stage_done[g->cur])); // the other buffer's last upload has landed
    return MUSE_OK;
}

// staged rows enqueued for upload, and the compute stream ordered behind every upload enqueued so far: call before
// anything on the compute stream reads the rows
static int group_ready(muse_group *g)
{
    int rc = group_flush(g);
    if (rc)
        return rc;
    if (g->upload_pending) {
        HIP_TRY(hipStreamWaitEvent(g->ctx->stream, g->uploaded, 0));
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
    int rc = use_device(g->ctx);
    if (rc)
        return rc;
    const size_t row_bytes = (size_t)g->N * sizeof(double);
    constexpr size_t STAGE_BYTES = 32u << 20;
    // small appends are staged from the SECOND one on: a group that is uploaded in one call (Muse.Run builds one
    // per call) never needs the staging pair
    bool small = (size_t)count * row_bytes < STAGE_BYTES / 4 && row_bytes <= STAGE_BYTES;
    if (small && !g->stage[0] && g->small_appends++ == 0)
        small = false;
    if (g->f32) // float32 storage: every append is narrowed on the host into the pinned staging pair (half the PCIe bytes too)
        small = true;
    if (small && !g->stage[0]) { // borrow the staging pair from the context's pool
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
                HIP_TRY(hipHostMalloc((void **)&buf, STAGE_BYTES, hipHostMallocDefault));
            g->stage[i] = buf;
            HIP_TRY(hipEventCreateWithFlags(&g->stage_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(g->stage_done[i], g->ctx->copy_stream));
        }
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
    for (int64_t r = 0; r < count; r++) {
        if (g->staged == g->stage_rows) {
            rc = group_flush(g);
            if (rc)
                return rc;
        }
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
    if (rc)
        return rc;
    if (g->f32)
        HIP_TRY(launch_synth_f32(g->rows32, g->stride, first, count, global_first, g->N, seed, flags, g->ctx->stream));
    else
        HIP_TRY(launch_synth(g->rows, g->stride, first, count, global_first, g->N, seed, flags, g->ctx->stream));
    g->M = std::max(g->M, first + count);
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

static void group_release(muse_group *g)
{
    if (!g || g->refs.fetch_sub(1) != 1)
        return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->copy_stream);
    (void)hipStreamSynchronize(g->ctx->stream);
    if (g->uploaded)
        (void)hipEventDestroy(g->uploaded);
    if (g->base())
        (void)hipFree((char *)g->base() - GROUP_GUARD * g->elem());
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

// ------------------------------------------------------------------- batch
static int ilog2(int64_t n)
{
    int l = 0;
    while (((int64_t)1 << l) < n)
        l++;
    return l;
}

// device reference spectrum for (ref, N) at FFT length n: fills X, xc
static int build_spectrum(muse_ctx *ctx, const double *ref_host, int N, int n, int normalize, double x_scale,
                          double xc_scale, double2 *X, double2 *xc, float2 *xcf, double *xs, int *zero_std)
{
    double *dref = nullptr;
    int *dstat = nullptr;
    double2 *dscr = nullptr; // n > 8192: global work buffer for the radix-2 passes
    HIP_TRY(hipMalloc(&dref, (size_t)N * sizeof(double)));
    hipError_t e = hipMalloc(&dstat, sizeof(int));
    if (e == hipSuccess && n > GENERIC_LDS_MAX_N)
        e = hipMalloc(&dscr, (size_t)n * sizeof(double2));
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
    (void)hipFree(dref);
    (void)hipFree(dstat);
    (void)hipFree(dscr);
    HIP_TRY(e);
    *zero_std = st;
    return MUSE_OK;
}

static hipError_t ensure_gscratch(muse_ctx *ctx, int64_t n, int slices_per_cu = GSCRATCH_SLICES_PER_CU)
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
static hipError_t ensure_twl(muse_ctx *ctx, int64_t n)
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

static void adopt_spectrum(muse_batch *b)
{
    b->X = b->sp->X;
    b->xc = b->sp->xc;
    b->xcp = b->sp->xcp;
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
    if (n > GENERIC_MAX_N)
        return fail(MUSE_ERR_UNSUPPORTED, "FFT length %lld > %d is not built", (long long)n, GENERIC_MAX_N);
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
    hipError_t e = hipMalloc(&b->ovf_count, 2 * sizeof(int)); // [0] overflow-pair count, [1] dynamic work counter
    if (e == hipSuccess)
        e = ensure_gscratch(ctx, n);
    muse_spectrum *sp = new (std::nothrow) muse_spectrum();
    if (!sp)
        e = hipErrorOutOfMemory;
    b->sp = sp;
    if (e == hipSuccess)
        e = hipMalloc(&sp->X, (size_t)(n / 2 + 1) * sizeof(double2));
    if (e == hipSuccess)
        e = hipMalloc(&sp->xc, (size_t)n * sizeof(double2));
    if (e == hipSuccess)
        e = hipMalloc(&sp->xcf, (size_t)n * sizeof(float2));
    if (e == hipSuccess)
        e = hipMalloc(&sp->xs, (size_t)n * sizeof(double));
    const bool long_n = n == 16384 || n == 32768 || n == 65536; // xcorr_long.hip: spectrum rows in lane order, sweep twiddles
    if (e == hipSuccess && (n == 4096 || long_n))
        e = hipMalloc(&sp->xcp, (size_t)n * sizeof(double2));
    if (e == hipSuccess && (n == 4096 || long_n) && N < n)
        e = hipMalloc(&sp->c1, (size_t)n * sizeof(double));
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
    if (long_n) {
        const int R1 = (int)(n / 4096);
        e = ensure_twl(ctx, n);
        if (e == hipSuccess)
            e = launch_lane_order_rows(b->xc, b->xcp, R1, ctx->stream);
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
    hipError_t e = hipMalloc(&b->ovf_count, 2 * sizeof(int));
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

static int ensure_scores(muse_batch *b)
{
    const int64_t M = b->g->M;
    if (M <= b->score_cap)
        return MUSE_OK;
    (void)hipFree(b->mv);
    (void)hipFree(b->lag);
    b->mv = nullptr;
    b->lag = nullptr;
    b->score_cap = 0;
    HIP_TRY(hipMalloc(&b->mv, (size_t)M * sizeof(double)));
    HIP_TRY(hipMalloc(&b->lag, (size_t)M * sizeof(int)));
    b->score_cap = M;
    return MUSE_OK;
}

// the launch parameters every fused kernel shares for batch b (group flushed, M > 0, scores allocated)
static FusedParams base_params(muse_batch *b)
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
    p.c1 = b->c1;
    p.twl = (b->logn >= 14 && b->logn <= 16) ? ctx->twl[b->logn - 14] : nullptr;
    p.tw1f = ctx->tw1f;
    p.twmf = ctx->twmf;
    p.tw2f = ctx->tw2f;
    p.xcf = b->xcf;
    p.xs = b->xs;
    p.screen_delta = ctx->screen_delta;
    return p;
}

extern "C" int muse_batch_score(muse_batch *b)
{
    if (!b)
        return fail(MUSE_ERR_INVALID, "NULL batch");
    muse_ctx *ctx = b->ctx;
    int rc = use_device(ctx);
    if (rc)
        return rc;
    rc = group_ready(b->g); // rows still in the staging buffer are uploaded (copy stream) ahead of the kernel
    if (rc)
        return rc;
    const int64_t M = b->g->M;
    if (M == 0)
        return MUSE_OK;
    rc = ensure_scores(b);
    if (rc)
        return rc;
    // long series work in the context's scratch buffer: its pointer must not be swapped (a concurrent
    // muse_batch_create growing it) between reading it and enqueueing the launch
    std::unique_lock<std::mutex> scratch_lock(ctx->stage_mu, std::defer_lock);
    if (b->n >= GENERIC_LDS_MAX_N)
        scratch_lock.lock();
    FusedParams p = base_params(b);
    b->scores_exact = true;
    // kernel selection: ctx->variant 0 = auto; the others are test hooks (muse_hip_test.h)
    int variant = KERNEL_GENERIC;
    if (b->n == 4096) {
        switch (ctx->variant) {
        case 0: case 10: variant = KERNEL_R16_FOLD; break; // fastest measured (profiles/)
        case 7: variant = KERNEL_R16_OCC3; break;          // rescales both series before the shared transform
        default: variant = KERNEL_GENERIC; break;
        }
        if (b->g->f32 && variant == KERNEL_GENERIC)
            return fail(MUSE_ERR_UNSUPPORTED, "the generic kernel does not read float32-storage groups");
        if (variant == KERNEL_R16_FOLD && b->N != 4096 && !b->c1) // (N < n needs the batch's correction table)
            variant = KERNEL_R16_OCC3;
        // a group of mixed-unit series (sigmas far apart inside most pairs) makes the default kernel hand most
        // pairs to kernel 7 anyway: once a pass over these rows has shown that, go there directly
        if (variant == KERNEL_R16_FOLD && ctx->variant == 0 && b->handoff_host && b->handoff_M == M &&
            (long long)*(volatile int *)b->handoff_host * 8 > p.npairs)
            variant = KERNEL_R16_OCC3;
    } else if (b->g->f32 && !(((b->n >= 512 && b->n <= 2048) || b->n == 8192 || b->n == 16384) && (ctx->variant == 0 || ctx->variant == 12))) {
        return fail(MUSE_ERR_UNSUPPORTED, "float32-storage groups run on the default kernels only (FFT lengths 512 ... 16384)");
    } else if (b->xcp && p.twl && (b->N == b->n || b->c1) && ((b->n >= 32768 && ctx->variant == 0) || (b->n >= 16384 && ctx->variant == 13))) {
        variant = KERNEL_LONG; // four-step, 4096-point rows on the n = 4096 kernel's transforms (xcorr_long.hip)
    } else if (((b->n >= 512 && b->n <= 2048) || b->n == 8192 || b->n == 16384) && (ctx->variant == 0 || ctx->variant == 12)) {
        variant = KERNEL_SMALL; // half-round transposes at 16 waves per CU (xcorr_small.hip)
    } else if (((b->n >= 512 && b->n <= 2048) || b->n >= 8192) && (ctx->variant == 0 || ctx->variant == 11)) {
        variant = KERNEL_STOCKHAM; // radix-16 Stockham through LDS / global scratch (xcorr_stockham.hip)
    }
    if (variant == KERNEL_GENERIC && b->n <= GENERIC_LDS_MAX_N)
        p.gscratch = nullptr; // the generic kernel takes a non-NULL scratch pointer as "work in global memory"
    LaunchTimer timer(ctx); // (brackets the fused launch alone: not the counter reset in front of it, not the redo launch behind it)
    LaunchTimer redo_timer(ctx, true); // the launch that redoes the listed pairs: its own sum (muse_ctx_redo_time)
    if (variant == KERNEL_R16_FOLD) {
        // pairs with a NaN/Inf series or with sigmas too far apart for one shared transform are listed by the kernel
        // (once per such series: 2 entries per pair) and redone by the rescaling kernel right behind it (no host round
        // trip: the count stays on the device and bounds the second launch's loop)
        if (2 * p.npairs > b->ovf_cap) {
            (void)hipFree(b->ovf_list);
            b->ovf_list = nullptr;
            b->ovf_cap = 0;
            HIP_TRY(hipMalloc(&b->ovf_list, (size_t)(2 * p.npairs) * sizeof(long long)));
            b->ovf_cap = 2 * p.npairs;
        }
        p.ovf_count = b->ovf_count;
        p.work_counter = b->ovf_count + 1;
        p.ovf_list = b->ovf_list;
        HIP_TRY(hipMemsetAsync(b->ovf_count, 0, 2 * sizeof(int), ctx->stream));
        HIP_TRY(timer.begin());
        HIP_TRY(launch_fused(p, variant, ctx->num_cus, ctx->stream));
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
        HIP_TRY(launch_fused(q, KERNEL_R16_OCC3, ctx->num_cus, ctx->stream));
        HIP_TRY(redo_timer.end());
        if (p.npairs >= 1024) { // (small groups: a pinned allocation costs more than it can save)
            if (!b->handoff_host)
                HIP_TRY(hipHostMalloc((void **)&b->handoff_host, sizeof(int), hipHostMallocDefault));
            *b->handoff_host = 0;
            b->handoff_M = M;
            HIP_TRY(hipMemcpyAsync(b->handoff_host, b->ovf_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        }
    } else if (variant == KERNEL_LONG) {
        // as above: NaN / Inf and sigma-spread pairs are listed (one entry per pair) and redone by the four-step kernel that
        // isolates and rescales the series first
        if (2 * p.npairs > b->ovf_cap) {
            (void)hipFree(b->ovf_list);
            b->ovf_list = nullptr;
            b->ovf_cap = 0;
            HIP_TRY(hipMalloc(&b->ovf_list, (size_t)(2 * p.npairs) * sizeof(long long)));
            b->ovf_cap = 2 * p.npairs;
        }
        p.ovf_count = b->ovf_count;
        p.ovf_list = b->ovf_list;
        HIP_TRY(hipMemsetAsync(b->ovf_count, 0, 2 * sizeof(int), ctx->stream));
        HIP_TRY(timer.begin());
        HIP_TRY(launch_fused(p, variant, ctx->num_cus, ctx->stream));
        HIP_TRY(timer.end());
        FusedParams q = p;
        q.pair_list = b->ovf_list;
        q.pair_count = b->ovf_count;
        q.npairs = std::min<long long>(p.npairs, (long long)ctx->num_cus * STOCKHAM_GLOBAL_WGS_PER_CU);
        HIP_TRY(redo_timer.begin());
        HIP_TRY(launch_fused(q, KERNEL_STOCKHAM, ctx->num_cus, ctx->stream));
        HIP_TRY(redo_timer.end());
    } else {
        HIP_TRY(timer.begin());
        HIP_TRY(launch_fused(p, variant, ctx->num_cus, ctx->stream));
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
    HIP_TRY(hipMemcpyAsync(lag, b->lag, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipMemcpyAsync(mv, b->mv, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return MUSE_OK;
}

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

static int ensure_select_ws(muse_batch *b, int64_t M, int64_t G, bool with_gid, int K, bool on_device)
{
    if (with_gid && M > b->gid_cap) {
        (void)hipFree(b->gid_dev);
        b->gid_dev = nullptr;
        b->gid_cap = 0;
        b->gid_valid = false;
        HIP_TRY(hipMalloc(&b->gid_dev, (size_t)M * sizeof(int)));
        b->gid_cap = M;
    }
    if (G > b->grp_cap) {
        (void)hipFree(b->gw.key);
        (void)hipFree(b->gw.first);
        (void)hipFree(b->gw.win);
        (void)hipFree(b->rec);
        (void)hipFree(b->selkey);
        b->gw = GroupWork{nullptr, nullptr, nullptr};
        b->rec = nullptr;
        b->selkey = nullptr;
        b->grp_cap = 0;
        HIP_TRY(hipMalloc(&b->gw.key, (size_t)G * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&b->gw.first, (size_t)G * sizeof(long long)));
        HIP_TRY(hipMalloc(&b->gw.win, (size_t)G * sizeof(long long)));
        HIP_TRY(hipMalloc(&b->rec, (size_t)G * sizeof(muse_record)));
        HIP_TRY(hipMalloc(&b->selkey, (size_t)G * sizeof(unsigned long long)));
        b->grp_cap = G;
    }
    const int64_t nb = (G + TOPN_CHUNK - 1) / TOPN_CHUNK;
    if (nb > b->cnt_cap) {
        (void)hipFree(b->cnt);
        b->cnt = nullptr;
        b->cnt_cap = 0;
        HIP_TRY(hipMalloc(&b->cnt, (size_t)nb * sizeof(int)));
        b->cnt_cap = nb;
    }
    if (nb * K > b->cand_cap) {
        (void)hipFree(b->cand);
        b->cand = nullptr;
        b->cand_cap = 0;
        HIP_TRY(hipMalloc(&b->cand, (size_t)(nb * K) * sizeof(muse_record)));
        b->cand_cap = nb * K;
    }
    if (on_device && nb > b->cnt_host_cap) { // (small selections copy the group records instead: no pinned memory)
        if (b->cnt_host)
            (void)hipHostFree(b->cnt_host); // (hipHostFree(NULL) leaves a sticky error behind)
        b->cnt_host = nullptr;
        b->cnt_host_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&b->cnt_host, (size_t)nb * sizeof(int), hipHostMallocDefault));
        b->cnt_host_cap = nb;
    }
    if (on_device && nb * K > b->cand_host_cap) {
        if (b->cand_host)
            (void)hipHostFree(b->cand_host);
        b->cand_host = nullptr;
        b->cand_host_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&b->cand_host, (size_t)(nb * K) * sizeof(muse_record), hipHostMallocDefault));
        b->cand_host_cap = nb * K;
    }
    return MUSE_OK;
}

// ---- filter-and-refine Run: ungrouped N = n = 4096 Runs under automatic kernel selection
static muse_batch::RunKey run_key(const muse_batch *b, const int32_t *group_id, int64_t G, int32_t max_lag, int32_t top_n,
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
static int32_t screen_path(const muse_batch *b, const muse_batch::RunKey &key, bool already_scored)
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
        (void)hipFree(b->scr_flags);
        (void)hipFree(b->scr_var);
        (void)hipFree(b->include);
        b->scr_flags = nullptr;
        b->scr_var = nullptr;
        b->include = nullptr;
        b->scr_cap = 0;
        HIP_TRY(hipMalloc(&b->scr_flags, (size_t)M * sizeof(unsigned)));
        HIP_TRY(hipMalloc(&b->scr_var, (size_t)M * sizeof(double)));
        HIP_TRY(hipMalloc(&b->include, (size_t)M));
        b->scr_cap = M;
    }
    const int64_t nkeys = screen_select_scratch(gid_dev ? G : M, top_n);
    if (gid_dev && G > b->scr_gcap) {
        (void)hipFree(b->scr_gmay);
        (void)hipFree(b->scr_gkplus);
        (void)hipFree(b->scr_gcert);
        b->scr_gmay = b->scr_gkplus = nullptr;
        b->scr_gcert = nullptr;
        b->scr_gcap = 0;
        HIP_TRY(hipMalloc(&b->scr_gmay, (size_t)G * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&b->scr_gkplus, (size_t)G * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&b->scr_gcert, (size_t)G * sizeof(int)));
        b->scr_gcap = G;
    }
    if (nkeys > b->scr_keys_cap) {
        (void)hipFree(b->scr_keys);
        b->scr_keys = nullptr;
        b->scr_keys_cap = 0;
        HIP_TRY(hipMalloc(&b->scr_keys, (size_t)nkeys * sizeof(unsigned long long)));
        b->scr_keys_cap = nkeys;
    }
    if (!b->refine_host)
        HIP_TRY(hipHostMalloc((void **)&b->refine_host, sizeof(int), hipHostMallocDefault));
    if (!b->err_host)
        HIP_TRY(hipHostMalloc((void **)&b->err_host, sizeof(unsigned long long), hipHostMallocDefault));
    if (!b->err_dev)
        HIP_TRY(hipMalloc(&b->err_dev, sizeof(unsigned long long)));
    if (4 * npairs > b->est_cap) { // two estimates per listed pair; the list holds the selection's pairs plus the guard sample
        (void)hipFree(b->est_save);
        b->est_save = nullptr;
        b->est_cap = 0;
        HIP_TRY(hipMalloc(&b->est_save, (size_t)(4 * npairs) * sizeof(double)));
        b->est_cap = 4 * npairs;
    }
    // the list takes the selection's pairs (at most npairs) plus the guard sample (about npairs / 1024, not de-duplicated
    // against the selection): 2 npairs entries, the same capacity the fp64 pass's hand-off list has
    if (2 * npairs > b->ovf_cap) {
        (void)hipFree(b->ovf_list);
        b->ovf_list = nullptr;
        b->ovf_cap = 0;
        HIP_TRY(hipMalloc(&b->ovf_list, (size_t)(2 * npairs) * sizeof(long long)));
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

static int score_screened(muse_batch *b, int32_t max_lag, int32_t top_n, double threshold, int32_t sign_filter,
                          int32_t abs_scores, const int *gid_dev = nullptr, int64_t G = 0)
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

// the kernel automatic selection takes for this batch's all-scores pass (bench.py names it in its roofline object)
extern "C" int muse_batch_kernel_name(muse_batch *b, char *name, int32_t cap)
{
    if (!b || !name || cap < 1)
        return fail(MUSE_ERR_INVALID, "NULL argument");
    // (the names rocprofv3 prints for the instantiations automatic selection launches: profiles/r*_counters.json is keyed by them)
    char k[96] = "xcorr_fused_generic";
    const bool padded = b->N < b->n;
    if (b->n == 4096)
        snprintf(k, sizeof(k), "xcorr_fused_n4096_fold<false, %s, %s>", padded ? "true" : "false", b->g->f32 ? "true" : "false");
    else if ((b->n >= 512 && b->n <= 2048) || b->n == 8192 || b->n == 16384)
        snprintf(k, sizeof(k), "xcorr_fused_small<%d, %s, false%s>", b->logn, padded ? "true" : "false", b->g->f32 ? ", true" : ", false");
    else if (b->n > 16384)
        snprintf(k, sizeof(k), "xcorr_fused_long<%d, %s, false>", b->logn, padded ? "true" : "false");
    snprintf(name, (size_t)cap, "%s", k);
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

// the label-group map of a Run on the device (re-sent only when it changed)
static int upload_group_ids(muse_batch *b, const int32_t *group_id, int64_t M)
{
    if (!group_id)
        return MUSE_OK;
    const bool same = b->gid_valid && (int64_t)b->gid_host.size() == M &&
                      memcmp(b->gid_host.data(), group_id, (size_t)M * sizeof(int32_t)) == 0;
    if (!same) {
        b->gid_host.assign(group_id, group_id + M);
        HIP_TRY(hipMemcpyAsync(b->gid_dev, b->gid_host.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice,
                               b->ctx->stream));
        b->gid_valid = true;
    }
    return MUSE_OK;
}

// after the synchronisation of a screened Run: did any re-evaluated row's estimate miss its fp64 score by more than the
// bound the selection assumed?  (Never observed -- the bound is ~3 600x the measured error -- but if it happens the bound
// cannot be trusted for the rows that were NOT re-evaluated either: the batch leaves the filter-and-refine path.)
static bool screen_guard_tripped(muse_batch *b)
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

static int run_select(muse_batch *b, const int32_t *group_id, int32_t G_in, int64_t series_offset, int32_t max_lag,
                      int32_t top_n, double threshold, int32_t sign_filter, int32_t abs_scores,
                      std::vector<muse_record> &out, bool already_scored = false, bool prescreened = false)
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
    const bool on_device = top_n <= TOPN_DEVICE_MAX && G > TOPN_CHUNK / 4;
    const int K = on_device ? top_n : 1;
    rc = ensure_select_ws(b, M, G, group_id != nullptr, K, on_device);
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
    HIP_TRY(launch_group_reduce(sp, b->gw, b->rec, b->selkey, ctx->stream));
    std::vector<muse_record> cands;
    if (on_device) {
        const int64_t nb = (G + TOPN_CHUNK - 1) / TOPN_CHUNK;
        HIP_TRY(launch_topn(b->rec, b->selkey, (int)G, K, b->cand, b->cnt, ctx->stream));
        const int *cnt = b->cnt_host;
        const muse_record *cand = b->cand_host;
        HIP_TRY(hipMemcpyAsync(b->cnt_host, b->cnt, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(b->cand_host, b->cand, (size_t)(nb * K) * sizeof(muse_record), hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
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
        std::vector<muse_record> rec((size_t)G);
        std::vector<unsigned long long> key((size_t)G);
        HIP_TRY(hipMemcpyAsync(rec.data(), b->rec, (size_t)G * sizeof(muse_record), hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipMemcpyAsync(key.data(), b->selkey, (size_t)G * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
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

static void emit(const std::vector<muse_record> &sel, int64_t *out_series, int32_t *out_lag, double *out_score,
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
    rc = ensure_select_ws(b, M, G, true, 1, false);
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
    HIP_TRY(launch_group_reduce(sp, b->gw, b->rec, b->selkey, ctx->stream));
    std::vector<unsigned long long> st((size_t)G);
    HIP_TRY(hipMemcpyAsync(out_records, b->rec, (size_t)G * sizeof(muse_record), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(st.data(), b->selkey, (size_t)G * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
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

extern "C" int muse_merge_group_records(const muse_record *records, const uint8_t *state, int32_t n_shards, int32_t G,
                                        int32_t max_lag, int32_t top_n, double threshold, int32_t sign_filter,
                                        int64_t *out_series, int32_t *out_lag, double *out_score, int32_t *out_count,
                                        double *out_mean_abs)
{
    if (n_shards < 0 || G < 0 || ((int64_t)n_shards * G > 0 && (!records || !state)))
        return fail(MUSE_ERR_INVALID, "bad shard records");
    if (sign_filter < -1 || sign_filter > 1)
        return fail(MUSE_ERR_INVALID, "sign_filter must be -1, 0 or 1");
    std::vector<muse_record> cands;
    for (int32_t g = 0; g < G; g++) {
        // shards are listed in ascending row order: the first one with a member holds the group's first member
        bool seen = false, nan_first = false, have = false;
        muse_record best{};
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
        if (!seen || nan_first || !have) // empty group, or its first member's score is NaN: never passes Results.passed
            continue;
        best.group = g;
        if (passed_host(best.score, best.lag, max_lag, threshold, sign_filter))
            cands.push_back(best);
    }
    std::vector<muse_record> sel = heap_select(std::move(cands), top_n);
    emit(sel, out_series, out_lag, out_score, out_count, out_mean_abs);
    return MUSE_OK;
}

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
    const bool long_n = (b0->n == 32768 || b0->n == 65536) && R >= 3 && !b0->g->f32 && ctx->variant == 0 && b0->logn >= 14 && ctx->twl[b0->logn - 14];
    bool one_pass = R > 1 &&
                    ((b0->n == 4096 && (ctx->variant == 0 || ctx->variant == 10)) ||
                     (small_n && !b0->g->f32 && (ctx->variant == 0 || ctx->variant == 12)) || long_n);
    for (int r = 0; r < R && one_pass; r++)
        one_pass = bs[r]->N == b0->N && (small_n || b0->N == b0->n || bs[r]->c1 != nullptr) && (!long_n || bs[r]->xcp != nullptr);
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
        tab[(size_t)r] = small_n ? bs[r]->xc : bs[r]->xcp;
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
        (void)hipFree(b0->ovf_list);
        b0->ovf_list = nullptr;
        b0->ovf_cap = 0;
        HIP_TRY(hipMalloc(&b0->ovf_list, (size_t)(2 * p.npairs) * sizeof(long long)));
        b0->ovf_cap = 2 * p.npairs;
    }
    p.ovf_count = b0->ovf_count;
    p.work_counter = b0->ovf_count + 1;
    p.ovf_list = b0->ovf_list;
    // (the long-series kernel works in the context's scratch buffer: its pointer must not be swapped between reading it and the launch)
    std::unique_lock<std::mutex> scratch_lock(ctx->stage_mu, std::defer_lock);
    if (long_n) {
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
    } else if (long_n)
        HIP_TRY(launch_fused_long(p, ctx->num_cus, ctx->stream));
    else
        HIP_TRY(launch_fused_multi(p, ctx->num_cus, ctx->stream));
    HIP_TRY(timer.end());
    // pairs holding a NaN/Inf series (listed once, by reference 0): redone per reference by the
    // kernel that isolates the dead series before the shared transform
    LaunchTimer redo_timer(ctx, true); // (one bracket around the R redo launches)
    if (!small_n)
        HIP_TRY(redo_timer.begin());
    for (int r = 0; r < R && !small_n; r++) {
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

// The filter-and-refine Run for R references over one group (muse_batch_run_many): ONE screening pass reads, reduces and
// forward-transforms every pair of series once and reports into each reference's arrays; keys, cut, compaction, fp64
// re-evaluation and guard then run per reference.  Sets `done` when the batches have been screened (otherwise nothing
// was touched and the caller takes the fp64 one-pass kernel).
static int screen_many(muse_batch *const *bs, int32_t R, const int32_t *group_id, int32_t G_in, int32_t max_lag,
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
    const bool on_device = top_n <= TOPN_DEVICE_MAX && G > TOPN_CHUNK / 4;
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

extern "C" int muse_batch_free(muse_batch *b)
{
    if (!b)
        return MUSE_OK;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->sp && b->sp->refs.fetch_sub(1) == 1) {
        (void)hipFree(b->sp->X);
        (void)hipFree(b->sp->xc);
        (void)hipFree(b->sp->xcp);
        (void)hipFree(b->sp->xcf);
        (void)hipFree(b->sp->xs);
        (void)hipFree(b->sp->c1);
        delete b->sp;
    }
    (void)hipFree(b->ovf_count);
    if (b->handoff_host)
        (void)hipHostFree(b->handoff_host);
    (void)hipFree(b->ovf_list);
    (void)hipFree(b->mv);
    (void)hipFree(b->lag);
    (void)hipFree(b->gid_dev);
    (void)hipFree(b->gw.key);
    (void)hipFree(b->gw.first);
    (void)hipFree(b->gw.win);
    (void)hipFree(b->rec);
    (void)hipFree(b->selkey);
    (void)hipFree(b->cand);
    if (b->cand_host)
        (void)hipHostFree(b->cand_host);
    if (b->cnt_host)
        (void)hipHostFree(b->cnt_host);
    (void)hipFree(b->cnt);
    (void)hipFree(b->scr_flags);
    (void)hipFree(b->scr_var);
    (void)hipFree(b->include);
    (void)hipFree(b->scr_keys);
    (void)hipFree(b->scr_gmay);
    (void)hipFree(b->scr_gkplus);
    (void)hipFree(b->scr_gcert);
    if (b->refine_host)
        (void)hipHostFree(b->refine_host);
    if (b->err_host)
        (void)hipHostFree(b->err_host);
    (void)hipFree(b->err_dev);
    (void)hipFree(b->est_save);
    muse_group *g = b->g;
    muse_ctx *ctx = b->ctx;
    delete b;
    group_release(g);
    ctx_release(ctx);
    return MUSE_OK;
}

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
    if (n > GENERIC_MAX_N || (!is_pow2(n) && n > 8192))
        return fail(MUSE_ERR_UNSUPPORTED, "FFT length %d is not built (powers of two up to %d, any n up to 8192)", n,
                    GENERIC_MAX_N);
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
    if (is_pow2(n) && n >= 2) {
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
extern "C" int muse_xcorr_groups(muse_group *gx, muse_group *gy, int32_t n, int32_t normalize, int32_t *lag, double *mv,
                                 int32_t *is_nil, double *cc)
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
    if (!is_pow2(n) || n < 512 || n > GENERIC_MAX_N) {
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
    hipError_t e = ensure_gscratch(ctx, n);
    if (e == hipSuccess && n >= 32768)
        e = ensure_twl(ctx, n);
    if (e != hipSuccess)
        return fail(MUSE_ERR_NOMEM, "scratch: %s", hipGetErrorString(e));
    double *dmv = nullptr, *dcc = nullptr;
    int *dlag = nullptr, *dnil = nullptr;
    auto cleanup = [&]() { (void)hipFree(dmv); (void)hipFree(dcc); (void)hipFree(dlag); (void)hipFree(dnil); };
    e = hipMalloc(&dmv, (size_t)M * sizeof(double));
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
    {
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
        p.gsmall = (p.logn >= 9 && p.logn <= 11) ? ctx->gsmall[p.logn - 9] : (p.logn == 13 || p.logn == 14) ? ctx->gsmall[p.logn - 10] : nullptr;
        p.gscratch = ctx->gscratch;
        p.gscratch_slices = (long long)(ctx->gscratch_elems / (size_t)n);
        p.twl = (p.logn >= 14 && p.logn <= 16) ? ctx->twl[p.logn - 14] : nullptr;
        p.mv = dmv;
        p.lag = dlag;
        p.nil_out = dnil;
        p.cc_out = dcc;
        LaunchTimer timer(ctx);
        e = timer.begin();
        if (e == hipSuccess)
            e = launch_two_sided(p, ctx->num_cus, ctx->stream);
        if (e == hipSuccess)
            e = timer.end();
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
    return MUSE_OK;
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
        HIP_TRY(hipHostMalloc((void **)&ctx->probe_buf, (2 * PROBE_WINDOWS + 1) * sizeof(unsigned long long), hipHostMallocDefault));
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
    if (*cnt == 0)
        return fail(MUSE_ERR_HIP, "clock probe did not start");
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
