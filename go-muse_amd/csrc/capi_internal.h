// capi_internal.h -- what the parts of the C-ABI implementation (capi_*.hip) share: the handle structures behind
// include/muse_hip.h's opaque types, error reporting, the launch timer and the helpers that cross file boundaries.
// Internal to libmuse_hip.so.
#pragma once
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
#include <map>
#include <new>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "muse_hip.h"
#include "muse_hip_test.h"
#include "xcorr_kernels.h"

using namespace muse;

extern thread_local std::string g_last_error; // muse_last_error(): per host thread (capi_context.hip)
int fail(int status, const char *fmt, ...);
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(MUSE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                        __FILE__, __LINE__);                                                       \
    } while (0)

// ----------------------------------------------------------------- handles
// Per-context cache of device and pinned-host allocations (capi_context.hip: dmalloc / dfree / hmalloc / hfree).  hipMalloc,
// hipFree (a device-wide synchronisation each) and hipHostMalloc (hundreds of microseconds) were most of a cold
// NewGroup -> Add -> NewBatch -> Run -> free cycle (profiles/r06_cold_path.txt); a freed block is kept by size class and handed to
// the next request of that class.  A block handed back must no longer be in use by the device (the handles synchronise
// their streams before they free, as they did in front of hipFree).
struct MemPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> size_of;            // every block this pool allocated (in use or cached): its class size
    std::multimap<size_t, void *> idle;                     // cached blocks by class size
    size_t idle_bytes = 0;
    size_t idle_cap = 0;                                    // cached bytes kept at most
    size_t block_cap = 0;                                   // larger blocks are never cached
};
// FFT lengths 2^17 ... 2^20 (xcorr_huge.hip, capi_huge.hip): twiddle tables per length, the work buffer of one batch of pairs,
// per-pair multiplier tables (two-sided xCorr), chunk sums / flags / tile maxima; grown on demand under muse_ctx::huge_mu
struct HugeWork {
    double2 *thi[4] = {nullptr, nullptr, nullptr, nullptr}, *tlo[4] = {nullptr, nullptr, nullptr, nullptr};
    double2 *Y = nullptr, *T = nullptr;
    double *part = nullptr, *snorm = nullptr, *sfin = nullptr, *sfin_x = nullptr, *amax = nullptr;
    size_t Y_bytes = 0, T_bytes = 0, part_bytes = 0, snorm_bytes = 0, sfin_bytes = 0, sfin_x_bytes = 0, amax_bytes = 0;
    // the all-scores pass runs its batches on TWO streams alternately (the tail of one batch's kernels under the head of the
    // next one's): a second work buffer and tile-maximum buffer, the second stream, and the events that fork / join it
    double2 *Y2 = nullptr;
    double *amax2 = nullptr;
    size_t Y2_bytes = 0, amax2_bytes = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
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
    double2 *wsplit = nullptr; // xcorr_real.hip: [15][1024] W_16384^(j k1) (FusedParams::wsplit)
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
    int xcorr_repeat = 1; // measurement hook (muse_test_xcorr_repeat)
    unsigned long long *dbg_stamps = nullptr; // diagnostic builds only (-DMUSE_REAL64_STAMPS): [CUs][16 waves][16 phases] cycle sums
    // measurement hook (muse_test_clock_probe_*): a one-wave kernel on its own stream sampling the shader clock
    hipStream_t probe_stream = nullptr;
    unsigned long long *probe_buf = nullptr; // pinned host memory: [2 * PROBE_WINDOWS] ticks + the window count behind them
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events, redo_events;
    double total_ms = 0.0, redo_ms = 0.0;
    int64_t launches = 0, redo_launches = 0;
    char pci[32] = {0}; // PCI bus id of the device ("0000:05:00.0"): tells two contexts on one GPU from two GPUs
    // muse_batch_run_rows (capi_rows.hip): idle slots (RowsSlot *) -- pinned staging, device rows, score buffers, a pinned
    // result record and an event each -- so that a Muse.Run allocates nothing in steady state
    MemPool dev_pool, host_pool;   // dmalloc / hmalloc
    HugeWork huge;                 // series longer than 65 536 samples
    int huge_batch_mb = 0;         // measurement hook (muse_test_huge_batch_mb): 0 = HUGE_BATCH_BYTES; negative: |mb| on ONE stream
    std::mutex huge_mu;
    std::mutex small_mu;           // small_free: the pinned record buffers of small Runs between batches (capi_run.hip)
    std::vector<std::pair<unsigned char *, int>> small_free; // (buffer, its capacity in slots)
    unsigned long long small_token = 0; // the stamp of the context's last small Run
    std::mutex timing_mu;          // events / redo_events (LaunchTimer::end from concurrent muse_batch_run_rows callers)
    std::atomic<bool> rows_always_copy{false}; // test hook (muse_test_rows_always_copy): never let a kernel read the pinned staging buffer
    std::vector<void *> rows_slots;
    std::mutex rows_mu;
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
    // FFT lengths above 65 536 (capi_huge.hip): the rows never change, so neither do a series' zNormalize statistics -- first sample,
    // mean of the shifted samples, 1 / sigma, flag per row, computed by the first pass that needs them and kept for every later Run
    // and every other reference (rows appended later are added; a re-allocation starts over)
    double *hstats = nullptr;
    int64_t hstats_cap = 0, hstats_rows = 0;
    uint64_t rewrites = 0;          // how often rows already in the group were rewritten (muse_group_fill_synthetic: rows are otherwise immutable)
    // allocations the group has outgrown: kept until the group goes (kernels enqueued before the growth may still read them),
    // so that growing never waits for the device (group_reserve)
    std::vector<void *> retired;
    int32_t N = 0;
    // Appends that are small (Group.Add calls muse_group_append once per Series) are packed into two pinned staging buffers
    // borrowed from the context's pool and uploaded asynchronously on the context's copy stream: a piece goes out whenever
    // STAGE_FLUSH_BYTES have gathered (so the copies run beside the caller's next Adds), the buffers alternate when one is
    // full; kernels are ordered behind the copies through `uploaded`.
    double *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    int cur = 0;
    int64_t staged = 0;     // rows packed into stage[cur] (already counted in M)
    int64_t flushed = 0;    // of those, the rows whose upload is enqueued
    int64_t stage_rows = 0; // capacity of one staging buffer, in rows
    int small_appends = 0;  // the first small append goes straight to the device (Muse.Run: one upload per group)
    // an open window (muse_group_stage): the caller is filling rows [0, win_rows) of stage[cur] itself -- from any number of
    // threads -- and commits them piece by piece (muse_group_commit); they become rows [M, M + win_rows) of the group
    int64_t win_rows = 0, win_committed = 0;
    std::mutex win_mu;
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
    double2 *xcw = nullptr; // n == 32768: FusedParams::xcw
    float2 *xcf = nullptr;
    double *xs = nullptr;
    double *c1 = nullptr; // n == 4096, N < 4096: indicator correlation (xcorr_r16_fast.hip, PADDED)
    double xmax = -1.0;   // max |X[f]| (lazily, by the first screened Run): scales the fp32 error bound
};

struct muse_batch {
    muse_ctx *ctx = nullptr;
    muse_group *g = nullptr;
    // the stream this batch's kernels and copies are enqueued on: the context's, except for the slot batches of
    // muse_batch_run_rows (capi_rows.hip), which own one each so that concurrent callers' kernels overlap
    hipStream_t own_stream = nullptr;
    hipStream_t stream() const { return own_stream ? own_stream : ctx->stream; }
    int32_t N = 0, n = 0, logn = 0;
    muse_spectrum *sp = nullptr; // owner of the five tables below (the pointers are copies)
    double *c1 = nullptr;
    double2 *X = nullptr, *xc = nullptr;
    double2 *xcp = nullptr; // n == 4096: xc in the lane order of xcorr_r16_fast.hip
    double2 *xcw = nullptr; // n == 32768: FusedParams::xcw (xcorr_real.hip)
    float2 *xcf = nullptr; // fp32 conj(X)/n (screening kernel)
    double *xs = nullptr;  // padded time-domain reference (exact re-evaluation)
    int *ovf_count = nullptr;
    // automatic kernel selection learns from the previous pass over the same (immutable) rows: the number of
    // pairs the default N = 4096 kernel handed to the rescaling kernel lands here (pinned, asynchronous copy)
    int *handoff_host = nullptr;
    int64_t handoff_M = -1;         // the rows the count was taken over: M of them, after handoff_rewrites rewrites of the group
    uint64_t handoff_rewrites = 0;
    long long *ovf_list = nullptr;
    int64_t ovf_cap = 0;
    double *mv = nullptr;
    int *lag = nullptr;
    int64_t score_cap = 0;
    // small Runs (launch_small_groups): small_cap slots of coherent pinned memory, handed back to the context on free
    unsigned char *small_out = nullptr;
    int small_cap = 0;
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
    // pinned host images of rec / selkey: a Run over up to EXACT_FEED_MAX_GROUPS groups feeds the heap one Score per group
    muse_record *rec_host = nullptr;
    unsigned long long *key_host = nullptr;
    int64_t rec_host_cap = 0;
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

int use_device(muse_ctx *ctx);

// HIP-event bracket of ONE kernel launch on the context's stream (muse_ctx_kernel_timing): begin() right in front of the
// launch, end() right behind it; a bracket that never reaches end() (an error return in between) destroys its events.
struct LaunchTimer {
    muse_ctx *ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool redo; // the bracket of the launches that redo listed pairs behind a fused launch (muse_ctx_redo_time)
    hipStream_t stream; // the stream the bracketed launch goes to
    explicit LaunchTimer(muse_ctx *c, bool redo_ = false, hipStream_t s = nullptr) : ctx(c), redo(redo_), stream(s ? s : c->stream) {}
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
            e = hipEventRecord(e0, stream);
        return e;
    }
    hipError_t end()
    {
        if (!e0 || !e1)
            return hipSuccess;
        const hipError_t e = hipEventRecord(e1, stream);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lock(ctx->timing_mu); // (muse_batch_run_rows: any number of callers on one context)
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

// ---- helpers shared by the parts
hipError_t pool_alloc(muse_ctx *ctx, bool host, void **out, size_t bytes); // capi_context.hip
void pool_free(muse_ctx *ctx, bool host, void *p);
void pool_drain(muse_ctx *ctx);                                            // frees every cached block (context teardown, tests)
template <class T> inline hipError_t dmalloc(muse_ctx *ctx, T **out, size_t bytes) { return pool_alloc(ctx, false, (void **)out, bytes); }
template <class T> inline hipError_t hmalloc(muse_ctx *ctx, T **out, size_t bytes) { return pool_alloc(ctx, true, (void **)out, bytes); }
inline void dfree(muse_ctx *ctx, void *p) { pool_free(ctx, false, p); }
inline void hfree(muse_ctx *ctx, void *p) { pool_free(ctx, true, p); }
void fill_twiddle(std::vector<double2> &v, size_t i, long long num, long long den);
void ctx_release(muse_ctx *ctx);                       // capi_context.hip: drops one reference, frees the context with the last
int group_ready(muse_group *g, hipStream_t stream = nullptr); // capi_group.hip: staged rows uploaded, the compute stream behind the copies
void group_release(muse_group *g);
void rows_slots_free(muse_ctx *ctx);                   // capi_rows.hip: the idle slots of muse_batch_run_rows (streams idle)
int ilog2(int64_t n);                                  // capi_batch.hip
// capi_huge.hip: FFT lengths above GENERIC_MAX_N up to HUGE_MAX_N (xcorr_huge.h)
void huge_free(muse_ctx *ctx);
int huge_reference(muse_ctx *ctx, const double *ref_host, int N, int n, double2 *X, double2 *table, int *zero_std);
int huge_score(muse_batch *b);
int huge_pairs(muse_ctx *ctx, const double *xrows, int64_t xstride, int Nx, int normalize_x, double x_scale, const double *yrows,
               int64_t ystride, int Ny, int normalize_y, int64_t M, int n, double cc_scale, double *mv, int *lag, int *nil, double *cc);
void adopt_spectrum(muse_batch *b);                    // the batch's table pointers from its muse_spectrum
int build_spectrum(muse_ctx *ctx, const double *ref_host, int N, int n, int normalize, double x_scale,
                          double xc_scale, double2 *X, double2 *xc, float2 *xcf, double *xs, int *zero_std);
hipError_t ensure_gscratch(muse_ctx *ctx, int64_t n, int slices_per_cu = muse::GSCRATCH_SLICES_PER_CU);
hipError_t ensure_twl(muse_ctx *ctx, int64_t n);
int ensure_scores(muse_batch *b);
muse::FusedParams base_params(muse_batch *b);
int ensure_select_ws(muse_batch *b, int64_t M, int64_t G, bool with_gid, int K, bool on_device);
// the selection of a Run happens on the device (each chunk's best top_n) only beyond EXACT_FEED_MAX_GROUPS groups: capi_run.hip, run_select
inline bool select_on_device(int32_t top_n, int64_t G) { return top_n <= TOPN_DEVICE_MAX && G > EXACT_FEED_MAX_GROUPS; }
muse_batch::RunKey run_key(const muse_batch *b, const int32_t *group_id, int64_t G, int32_t max_lag, int32_t top_n,
                                  double threshold, int32_t sign_filter, int32_t abs_scores);
int32_t screen_path(const muse_batch *b, const muse_batch::RunKey &key, bool already_scored);
int score_screened(muse_batch *b, int32_t max_lag, int32_t top_n, double threshold, int32_t sign_filter,
                          int32_t abs_scores, const int *gid_dev = nullptr, int64_t G = 0);
int upload_group_ids(muse_batch *b, const int32_t *group_id, int64_t M);
bool screen_guard_tripped(muse_batch *b);
int run_select(muse_batch *b, const int32_t *group_id, int32_t G_in, int64_t series_offset, int32_t max_lag,
                      int32_t top_n, double threshold, int32_t sign_filter, int32_t abs_scores,
                      std::vector<muse_record> &out, bool already_scored = false, bool prescreened = false);
void emit(const std::vector<muse_record> &sel, int64_t *out_series, int32_t *out_lag, double *out_score,
                 int32_t *out_count, double *out_mean_abs);
int screen_many(muse_batch *const *bs, int32_t R, const int32_t *group_id, int32_t G_in, int32_t max_lag,
                       int32_t top_n, double threshold, int32_t sign_filter, int32_t abs_scores, bool &done);
