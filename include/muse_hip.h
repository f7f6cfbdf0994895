/*
 * muse_hip.h -- C ABI of libmuse_hip.so: the MI355X (gfx950) engine behind
 * go-muse's z-normalized cross-correlation hot path.
 *
 * The reference (aouyang1/go-muse, pure Go) has no FFI layer; the natural
 * seam is its exported batch API.  Each entry point below names the reference
 * interface it replaces (paths relative to /root/reference).  The Go-side
 * cgo binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns MUSE_OK (0) or a negative
 *     muse_status; text for the last error on the calling thread comes from
 *     muse_last_error().
 *   - no caller pointer is retained after a call returns (cgo rule): uploads
 *     copy.  Caller data is never mutated (the reference's zNormalize mutates
 *     Series values in place, xcorr.go:86,93; this engine does not).
 *   - one context = one GPU.  Several contexts may live in one process, on
 *     different devices or on the same one: the host mirrors shard a Group
 *     over a list of contexts inside ONE process (one host thread per
 *     context: muse.hpp Engine::List, muse_hip.go SetDevices, muse.py
 *     Batch(engines=...)); under torch.distributed / RCCL it is one process
 *     and one context per GPU (go-muse_amd/dist.py).  A handle may be used
 *     from any host thread.  Calls on DIFFERENT group / batch handles of one
 *     context may be in flight at the same time (muse_test.go:203-214 drives
 *     one Muse from many goroutines: every Muse.Run owns its group and batch);
 *     at most one call per handle, at most one muse_batch_score_many /
 *     _run_many per context, and kernel timing (muse_ctx_kernel_timing) only
 *     with a single caller.  All work of a context runs on its one stream.
 *   - there is NO CPU fallback: every compute entry point fails with
 *     MUSE_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef MUSE_HIP_H
#define MUSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUSE_HIP_ABI_VERSION 5

typedef enum muse_status {
    MUSE_OK = 0,
    MUSE_ERR_INVALID = -1,     /* bad argument / NULL handle                      */
    MUSE_ERR_LENGTH = -2,      /* length mismatch: muse_batch.go:24-28, muse.go:68-70, group.go:45-51 */
    MUSE_ERR_ZERO_STD = -3,    /* "Invalid input query": sigma(ref)==0, muse_batch.go:39-41 */
    MUSE_ERR_NO_DEVICE = -4,   /* no usable gfx950 device                         */
    MUSE_ERR_HIP = -5,         /* a HIP runtime call failed                       */
    MUSE_ERR_UNSUPPORTED = -6, /* FFT length outside the built kernels (n > 65536) */
    MUSE_ERR_NOMEM = -7,
    MUSE_ERR_EMPTY = -8        /* empty reference: muse.go:24-26                  */
} muse_status;

/* SignFilter values: results.go:20-26 */
#define MUSE_SIGN_ANY 0
#define MUSE_SIGN_POS 1
#define MUSE_SIGN_NEG (-1)

typedef struct muse_ctx muse_ctx;
typedef struct muse_group muse_group;
typedef struct muse_batch muse_batch;

/* One top-N candidate as exchanged between shards. 24 bytes. */
typedef struct muse_record {
    int64_t series; /* global series (row) index of the group's winner */
    double score;   /* clamped score: |mv| (Batch) or signed mv (Muse)  */
    int32_t lag;
    int32_t group;  /* group id (or series index when ungrouped, truncated) */
} muse_record;

int muse_abi_version(void);
const char *muse_last_error(void);
const char *muse_status_string(int status);

/* ------------------------------------------------------------ context */
/* Streams, twiddle tables, scratch.  device = HIP device ordinal.      */
int muse_ctx_create(int32_t device, muse_ctx **out);
/* Number of usable (gfx950) devices, ordinals 0 .. count-1: a host that shards a Group over the GPUs of a node
 *  creates one context per device and one host thread per context. */
int muse_device_count(int32_t *count);
int muse_ctx_destroy(muse_ctx *ctx);
int muse_ctx_synchronize(muse_ctx *ctx);
/* A context keeps the device and pinned-host blocks its groups and batches hand back (by size class, at most 1 GB of HBM
 * and 192 MB of pinned memory) and gives them to the next group / batch of the same shape: a steady
 * NewGroup -> Add -> NewBatch -> Run -> free cycle then makes no hipMalloc / hipFree / hipHostMalloc (each a device-wide
 * synchronisation or hundreds of microseconds).  muse_ctx_trim returns every cached block to the system. */
int muse_ctx_trim(muse_ctx *ctx);
/* name: >= 64 bytes.  Any out pointer may be NULL. */
int muse_ctx_device_info(muse_ctx *ctx, char *name, int32_t name_cap,
                         int32_t *compute_units, int64_t *hbm_bytes);
/* PCI bus id of the context's device ("0000:05:00.0"; cap >= 16): what tells N contexts on N GPUs from N contexts on one
 * (the sharded Runs of SURVEY 8e report it per shard; bench.py prints it per rank). */
int muse_ctx_device_pci_bus_id(muse_ctx *ctx, char *out, int32_t cap);
/* Filter-and-refine Run: OPT-IN (off by default: every Run scores every series with the float64 kernel, the
 * arithmetic of the reference, xcorr.go:160-197).  enable = 1: Runs over groups of >= 32768 * 4096 samples after
 * padding, where it starts to pay; enable = n > 1: Runs over >= n series; 0: off.  When enabled, a muse_batch_run /
 * muse_batch_run_shard / muse_batch_run_many (with or without label groups) over that many series of length
 * 257 .. 65536 screens every series with an fp32 transform (a bound E on its error is derived from the reference's
 * spectrum), re-evaluates in fp64 exactly those rows whose optimistic selection key reaches the top_n-th best
 * pessimistic key, and selects among the re-evaluated rows only: the records returned are the ones the all-fp64 Run
 * returns (a run-time guard re-does the Run in fp64 if an estimate is found outside the bound).
 * muse_batch_read_scores after such a Run re-scores every row in fp64 first.
 * Not every Run can take it: muse_batch_run_groups and muse_batch_run_rows always score in float64, and the host mirrors'
 * Batch.Run goes through muse_batch_run_groups whenever it has at most 65 536 label groups (it feeds Results one Score per
 * group, the reference's own order among exactly tied scores) -- so with the mirrors screening only ever applies to Runs
 * over more label groups than that (Run(nil) over a large Group). */
int muse_ctx_set_screening(muse_ctx *ctx, int32_t enable);
/* Which path the last muse_batch_run / muse_batch_run_shard on this batch took: *screened = 1 for filter-and-refine,
 * *refined_pairs = pairs of series it re-evaluated in fp64.  Any out pointer may be NULL. */
int muse_batch_last_run_info(muse_batch *b, int32_t *screened, int64_t *refined_pairs);
/* ... and why: MUSE_RUN_PATH_FP64 (screening off or the Run not eligible), _SCREENED, _FP64_COSTLY (an earlier
 * screened Run with exactly these filters over these rows re-evaluated more than a quarter of the pairs: the plain
 * fp64 pass is cheaper for it; other filters on the same batch are still screened), _FP64_GUARD (an estimate was once
 * found outside the bound: the batch stays on the fp64 path). */
#define MUSE_RUN_PATH_FP64 0
#define MUSE_RUN_PATH_SCREENED 1
#define MUSE_RUN_PATH_FP64_COSTLY 2
#define MUSE_RUN_PATH_FP64_GUARD 3
int muse_batch_last_run_path(muse_batch *b, int32_t *path);
/* HIP-event timing of the fused kernel on the stream it is launched on:
 * enable, run, then read (sum of launch durations in ms, launch count). */
int muse_ctx_kernel_timing(muse_ctx *ctx, int32_t enable);
int muse_ctx_kernel_time(muse_ctx *ctx, double *total_ms, int64_t *launches);
/* The same for what muse_ctx_kernel_time leaves out: the launches that REDO the pairs a fused launch listed (a NaN / Inf
 * series, sigmas too far apart for one shared transform) with the kernel that isolates and rescales first -- one bracket per
 * pass (many references: one around the R redo launches).  Microseconds for an empty list, a second pass for a group of
 * mixed-unit series. */
int muse_ctx_redo_time(muse_ctx *ctx, double *total_ms, int64_t *brackets);
/* Name of the kernel automatic selection takes for this batch's all-scores pass (name: >= 64 bytes). */
int muse_batch_kernel_name(muse_batch *b, char *name, int32_t name_cap);

/* -------------------------------------------------------------- group */
/* Device-resident row-major M x N float64 comparison matrix; replaces the
 * per-Series []float64 storage behind Group.Add (group.go:31-56: one length
 * per group).  Never mutated by any run. */
int muse_group_create(muse_ctx *ctx, int64_t capacity_rows, int32_t N,
                      muse_group **out);
/* OPT-IN float32-STORAGE group: the rows are kept as float32 in HBM -- half the bytes the float64 group
 * streams per Run -- and widened exactly to float64 as the kernels consume them; all arithmetic stays float64.
 * muse_group_append narrows the caller's float64 samples (round to nearest) on the way in, so scores are those of
 * the reference applied to the ROUNDED inputs, not to the caller's: they no longer match the float64 reference to
 * 1e-6 in general (relative input perturbation 6e-8 per sample), which is why this is never the default.
 * Built for series of length 257 .. 16384 (FFT lengths 512 ... 16384: the float32-row loaders live in the kernels
 * automatic selection takes for them); MUSE_ERR_UNSUPPORTED otherwise.  Such a group works with muse_batch_create /
 * _score(s) / _run / _run_shard / _run_groups / _score_many (one pass per reference); the opt-in filter-and-refine
 * Run and the two-sided xCorr do not apply to it.  muse_group_read returns the stored values widened to float64. */
int muse_group_create_f32(muse_ctx *ctx, int64_t capacity_rows, int32_t N,
                          muse_group **out);
/* Appends count rows read from host memory (row_stride in doubles, >= N).
 * This is what Group.Add calls once per Series (count = 1) or per slab. */
int muse_group_append(muse_group *g, const double *rows, int64_t count,
                      int64_t row_stride);
/* The same without a staging copy inside the library (Group.Add "stages series into a C-allocated pinned
 * buffer"): muse_group_stage opens a WINDOW of pinned host memory for up to `count` more rows -- *granted <= count of them,
 * row-major, N doubles per row, at *window -- which the caller fills itself, from as many threads (goroutines) as it likes:
 * the copy out of the Series' own slices (series.go:20) is the only one the host makes.  muse_group_commit(first, k) hands
 * rows [first, first + k) of the window to the copy stream as soon as they are filled (any thread, any order, every row of
 * the window exactly once); the upload of the first pieces runs beside the filling of the later ones, and with the last
 * commit the rows join the group as rows [M, M + granted).  Two windows alternate: the next muse_group_stage returns while
 * the previous window is still crossing PCIe.  One window at a time per group; no other call on the group between
 * _stage and its last _commit.  float64 groups only (MUSE_ERR_UNSUPPORTED for float32-storage groups). */
int muse_group_stage(muse_group *g, int64_t count, double **window, int64_t *granted);
int muse_group_commit(muse_group *g, int64_t first, int64_t count);
/* create + append in one call */
int muse_group_upload(muse_ctx *ctx, const double *rows, int64_t M, int32_t N,
                      int64_t row_stride, muse_group **out);
/* Fills rows [first, first+count) ON DEVICE with the rect+noise workload of
 * SURVEY 8d (after example_test.go:15-20), keyed by (seed, global row index =
 * global_first + local row, sample index): bench/test utility; rows become
 * part of the group (size grows to first+count if needed). ref_out (N doubles,
 * host, may be NULL) receives the reference series of the workload.
 * By default 1 row in 1024 is an exact copy of the reference (score 1, lag 0) and 1 in 1024 a constant row
 * (sigma == 0 path), as SURVEY 8d prescribes; flags switch either off (a workload whose top-N is NOT a tie
 * among copies). */
#define MUSE_SYNTH_NO_COPIES 1u
#define MUSE_SYNTH_NO_CONSTANTS 2u
int muse_group_fill_synthetic(muse_group *g, int64_t first, int64_t count,
                              int64_t global_first, uint64_t seed,
                              uint32_t flags, double *ref_out);
int muse_group_shape(muse_group *g, int64_t *M, int32_t *N);
/* D2H copy of rows [first, first+count) (dense, N doubles per row): lets a
 * checker feed byte-identical inputs to a CPU oracle. */
int muse_group_read(muse_group *g, int64_t first, int64_t count, double *out);
int muse_group_free(muse_group *g);

/* -------------------------------------------------------------- batch */
/* NewBatch (muse_batch.go:23-52) / New (muse.go:23-42): validates
 * N == group length (MUSE_ERR_LENGTH), n = nextPowOf2(N), reference spectrum
 * FFT(zeroPad(zNormalize(ref)/(N-1), n)) computed on the device and kept
 * resident.  MUSE_ERR_ZERO_STD when sigma(ref) == 0; MUSE_ERR_EMPTY when
 * N < 1.  ref is copied, not mutated.
 * Divergence: N == 1 is rejected with MUSE_ERR_INVALID.  The reference accepts it (New / NewBatch divide by
 * N - 1 = 0: every score is NaN and never passes Results.passed), so the observable outcome -- no scores -- is the
 * same, but here it is an error at creation instead of an empty result. */
int muse_batch_create(muse_ctx *ctx, muse_group *g, const double *ref,
                      int32_t N, muse_batch **out);
/* A batch for another group against the SAME reference: shares src's
 * spectrum tables (reference-counted), so it costs no transform and two
 * small allocations.  Muse.Run (muse.go:46-92) scores one small group per
 * call against the Muse's one reference: create the template once in New,
 * then one _create_like + _run + _free per call.  MUSE_ERR_LENGTH when the
 * group's length differs from the reference's (muse.go:68-70). */
int muse_batch_create_like(muse_batch *src, muse_group *g, muse_batch **out);
/* Muse.Run (muse.go:46-92) in ONE call: the M rows of one label group, read from host memory (row_stride in doubles,
 * >= the reference's length: MUSE_ERR_LENGTH otherwise, muse.go:68-70), are scored against tmpl's reference -- tmpl is the
 * batch New made once (muse_batch_create over an empty group); its own group is not touched -- and the group's winner comes
 * back as muse_batch_run_groups reports it for G = 1: *out_winner = the member with the largest |clamped score| among the
 * members whose score is a number (series = its row index, -1 if none; the first one on ties, muse.go:86), *out_state = 1
 * the first member's score is a number (out_winner is the Score Muse.Run passes to Results.Update), 2 the first member
 * scores NaN (x > NaN never replaces it: the group's score is NaN and never passes Results.passed), 0 for M = 0
 * (muse.go:47-50).  abs_scores = 0: signed score clamped to [-1, 1] (Muse.Run); 1: |score| clamped to 1 (a Batch of one
 * label group).  rows are copied before the call returns and never mutated.
 * One host -> HBM copy, the fused kernel of the length, one reduction kernel that writes the record into pinned host
 * memory, one event: no allocation, no free and no device-wide synchronisation in steady state (a pool of slots per
 * context); any number of host threads may call it on one tmpl at the same time (muse_test.go:203-214). */
int muse_batch_run_rows(muse_batch *tmpl, const double *rows, int64_t M, int64_t row_stride,
                        int32_t abs_scores, muse_record *out_winner, uint8_t *out_state);
/* The same with one pointer per row (rows[r] -> the N samples of row r): the comparison series of a Muse.Run are separate
 * slices (muse.go:46, compGraphs []*Series), gathered here straight into the call's pinned buffer instead of being packed
 * by the caller first.  (cgo: an array of Go pointers may only cross when the slices are pinned, runtime.Pinner;
 * INTEGRATION.md keeps the packed form for Go.) */
int muse_batch_run_row_ptrs(muse_batch *tmpl, const double *const *rows, int64_t M,
                            int32_t abs_scores, muse_record *out_winner, uint8_t *out_state);
int muse_batch_fft_len(muse_batch *b, int32_t *n);
/* The batch's x (muse_batch.go:47): n/2+1 complex128, interleaved re,im. */
int muse_batch_spectrum(muse_batch *b, double *out);
/* The hot loop of Batch.scoreSingle / Muse.Run (muse_batch.go:68-73,
 * muse.go:64-71): one fused kernel launch computes, for every series of the
 * group, exactly what xCorrWithX returns (xcorr.go:160-197) minus the cc
 * slice: lag and signed max value; sigma==0 series give (0, 0.0).  Results
 * stay on the device (asynchronous; see muse_ctx_synchronize). */
int muse_batch_score(muse_batch *b);
/* muse_batch_score + D2H of the per-series results (lag[M], mv[M]). */
int muse_batch_scores(muse_batch *b, int32_t *lag, double *mv);
/* Batch.Run + Results.Update + Results.Fetch (muse_batch.go:99-130,
 * results.go:46-87); abs_scores = 0 gives the Muse.Run post-processing
 * (muse.go:72-90: signed score clamped to [-1,1], group max by |score|).
 *   group_id : host int32[M], label-group of each series in [0,G), or NULL =
 *              every series its own group (Run(nil) with distinct labels).
 *              Groups are fed to Results in group-id order; inside a group the
 *              first series (lowest index) wins ties (muse_batch.go:87).
 *   outputs  : up to top_n entries in Fetch order (descending |score|),
 *              out_series = index of each group's winning series;
 *              *out_mean_abs = mean |score| (NaN when empty, results.go:86). */
int muse_batch_run(muse_batch *b, const int32_t *group_id, int32_t G,
                   int32_t max_lag, int32_t top_n, double threshold,
                   int32_t sign_filter, int32_t abs_scores,
                   int64_t *out_series, int32_t *out_lag, double *out_score,
                   int32_t *out_count, double *out_mean_abs);
/* Sharded form of the same run: this context holds rows
 * [series_offset, series_offset + M) of a larger group and every label group
 * lives on ONE shard.  Returns this shard's top-N candidates (<= top_n
 * records, descending |score|); the host gathers the records of all shards
 * (RCCL all_gather of top_n * 24 B per rank) and passes them to
 * muse_merge_records.  group ids are global. */
int muse_batch_run_shard(muse_batch *b, const int32_t *group_id, int32_t G,
                         int64_t series_offset, int32_t max_lag, int32_t top_n,
                         double threshold, int32_t sign_filter,
                         int32_t abs_scores, muse_record *out_records,
                         int32_t *out_count);
/* Final Results.Update/Fetch over gathered shard candidates (host only, no
 * GPU): records in any order; they are fed to the top-N heap in group order. */
int muse_merge_records(const muse_record *records, int64_t count, int32_t top_n,
                       int64_t *out_series, int32_t *out_lag, double *out_score,
                       int32_t *out_count, double *out_mean_abs);
/* Sharded Run whose label groups may STRADDLE shards ("... or the per-group partial maxima are merged
 * before top-N"; e.g. Run(["graph"]) over a Group whose graphs are interleaved, cut into contiguous row ranges): every
 * shard reports, per label group and unfiltered, its winner among the members whose score is a number
 * (out_records[g].series = global index, -1 if none) and out_state[g] = 0 no member on this shard / 1 members, the first
 * one's score is a number / 2 the first member's score is NaN (muse_batch.go:87: x > NaN never replaces it, so a group
 * whose FIRST member overall scores NaN yields NaN).  G records and states per shard; group ids are global and
 * required (ungrouped Runs are exact with muse_batch_run_shard). */
int muse_batch_run_groups(muse_batch *b, const int32_t *group_id, int32_t G,
                          int64_t series_offset, int32_t abs_scores,
                          muse_record *out_records, uint8_t *out_state);
/* The per-group part of that merge alone (host only): records / state hold n_shards x G entries, shard-major, shards in
 * ascending row order; out_records[g] = the group's winner (the first shard with a member decides the NaN rule, the maximum by
 * |score| wins, the earlier shard on ties: muse_batch.go:87), out_state[g] = 0 the group has no member / 1 out_records[g] is its
 * Score / 2 its first member scores NaN, so its score is NaN.  This is what a host feeds through Results.Update, one Score per
 * label group in group order (muse_batch.go:124-128), to reproduce the reference's heap HISTORY and with it the order among
 * exactly tied scores -- also into a Results that earlier Runs have filled (results.go:55-72).  n_shards = 1 turns one
 * muse_batch_run_groups result into that feed. */
int muse_merge_group_winners(const muse_record *records, const uint8_t *state,
                             int32_t n_shards, int32_t G,
                             muse_record *out_records, uint8_t *out_state);
/* Merge of those per-shard records (host only): records / state hold n_shards x G entries, shard-major, shards in
 * ascending row order.  Per group: the first shard with a member decides the NaN rule, the maximum by |score| wins
 * (the earlier shard on ties: muse_batch.go:87 keeps the earlier series); then Results.passed, the top-N heap and
 * Fetch exactly as muse_merge_records. */
int muse_merge_group_records(const muse_record *records, const uint8_t *state,
                             int32_t n_shards, int32_t G, int32_t max_lag,
                             int32_t top_n, double threshold, int32_t sign_filter,
                             int64_t *out_series, int32_t *out_lag, double *out_score,
                             int32_t *out_count, double *out_mean_abs);
/* Many references against one resident group (the
 * README.md:10-13 use case iterates references and groupings over a fixed
 * set of series, i.e. one NewBatch + Run per reference against the same
 * Group).  The R batches must share the context and the group; each is
 * scored exactly as muse_batch_score would, but in ONE pass over the rows:
 * every pair of series is read and forward-transformed once and correlated
 * against all R reference spectra (FFT lengths 512 ... 16384, i.e.
 * 256 < N <= 16384, from two references on; FFT lengths 32768 and 65536 from
 * three references on: the row spectra stay in the workgroup's scratch slice
 * and every reference takes product, second transform and argmax from there;
 * float32-storage groups: FFT length 4096; shorter series and forced kernel
 * variants score the batches one after the other).  Results land in each
 * batch's own buffers (muse_batch_scores / _run on a batch re-score it). */
int muse_batch_score_many(muse_batch *const *batches, int32_t R);
/* Copies back the (lag, signed mv) of the last scoring pass WITHOUT re-scoring
 * (muse_batch_scores = muse_batch_score + this). */
int muse_batch_read_scores(muse_batch *b, int32_t *lag, double *mv);
/* muse_batch_score_many followed by Batch.Run's selection for every batch
 * with the same grouping and Results settings: outputs are R consecutive
 * blocks of top_n entries (out_series[r*top_n + i], ...), out_count[r] and
 * out_mean_abs[r] per reference.  Any output may be NULL. */
int muse_batch_run_many(muse_batch *const *batches, int32_t R,
                        const int32_t *group_id, int32_t G, int32_t max_lag,
                        int32_t top_n, double threshold, int32_t sign_filter,
                        int32_t abs_scores, int64_t *out_series,
                        int32_t *out_lag, double *out_score,
                        int32_t *out_count, double *out_mean_abs);
int muse_batch_free(muse_batch *b);

/* ------------------------------------------- single-pair entry points */
/* xCorrWithX as exercised by xcorr_test.go:204-286: ref and y of length N,
 * FFT length n (any n >= N; n need not be a power of two -- then a direct
 * O(n^2) device kernel is used).  cc (n doubles, may be NULL) receives the
 * full correlation.  Returns MUSE_OK with *is_nil = 1 (lag 0, mv 0) where
 * the reference returns (nil, 0, 0). */
int muse_xcorr_with_x(muse_ctx *ctx, const double *ref, const double *y,
                      int32_t N, int32_t n, double *cc, int32_t *lag,
                      double *mv, int32_t *is_nil);
/* xCorr (xcorr.go:102-153): n is raised to max(n, lenx, leny); cc holds that
 * many doubles.  n is used as given (not rounded up: the reference transforms any length): powers of two up to 2^20 run
 * the batched kernels; any other n up to 8192 the direct kernel; any other n up to 2^19 is folded out of the correlation at
 * a power of two L >= 2 n, where nothing wraps around (cc_n[k] = r[k] + r[k - n]; the fold and the argmax run on the host). */
int muse_xcorr(muse_ctx *ctx, const double *x, int32_t lenx, const double *y,
               int32_t leny, int32_t n, int32_t normalize, double *cc,
               int32_t *lag, double *mv, int32_t *is_nil);
/* xCorr (xcorr.go:102-153) for M independent pairs in ONE launch: pair i = (row i of gx, row i of gy).
 * The two groups may hold series of different lengths (each is zero-padded on its own, xcorr.go:129-130); n is raised
 * to max(n, Nx, Ny) (xcorr.go:104-106).  Powers of two 2^17 ... 2^20 run the long-series kernels (xcorr_huge.hip: one series
 * per transform, every x its own multiplier table).  FFT lengths 512 ... 65536 that are powers of two run the batched kernels, in two
 * forms: n <= 4096 and n = 65536 -- z = (x read backwards) + i y, one forward transform, cc = Im FFT(Z^2) / 2n on the xCorrWithX
 * kernels' transforms (xcorr_small.hip; xcorr_two_sided.hip: n = 4096 and, on the four-step transform with one scratch slice per
 * workgroup, n = 65536); n = 8192, 16384, 32768 -- each series a REAL transform of n / 2 complex points, X parked in the
 * workgroup's scratch slice, cc = FFT_n(Y conj X) / n untangled and re-tangled at mirror pairs of bins (xcorr_real.hip);
 * scale 1 / (n (n - 1)) when normalized else 1 / n, exactly as xcorr.go:139-143.  At n = 65536 with Nx = Ny = n the rows are
 * read once and the spectrum squared unscaled; pairs whose series differ by more than 2^16 in scale (or whose magnitudes are
 * extreme) are listed on the device and redone by a second launch that takes the statistics first.  Finite samples whose SQUARES leave the float64 range (|x| >~ 1e154) are looked at again by a device
 * kernel and give what the reference's arithmetic gives: finite correlations (raw: the pair is recomputed at magnitude 1 and
 * scaled back by an exact power of two, so the results are numbers exactly as long as n max|x| max|y| stays inside the float64
 * range, as in xcorr.go:108-143), all zeros (normalized, sigma = +Inf) or NaN (the reference's own sums overflow).  Any other n (the reference's n = 5 tables, short series) goes pair by pair
 * through muse_xcorr's path.  The groups are not mutated (the reference's zNormalize mutates x and y in place).
 * Outputs (host): lag[M], mv[M]; is_nil[M] (may be NULL) = 1 where the reference returns (nil, 0, 0), i.e. normalize
 * and sigma(x) == 0 or sigma(y) == 0; cc (may be NULL): M x n correlations (rows of nil pairs are zero). */
int muse_xcorr_groups(muse_group *gx, muse_group *gy, int32_t n, int32_t normalize,
                      int32_t *lag, double *mv, int32_t *is_nil, double *cc);
/* The same from host memory: x_rows is M x lenx, y_rows M x leny, dense row-major; uploads, runs, frees. */
int muse_xcorr_batch(muse_ctx *ctx, const double *x_rows, const double *y_rows, int64_t M,
                     int32_t lenx, int32_t leny, int32_t n, int32_t normalize,
                     int32_t *lag, double *mv, int32_t *is_nil, double *cc);
/* nextPowOf2 (xcorr.go:19-24), same floating formula. */
int64_t muse_next_pow2(double val);

#ifdef __cplusplus
}
#endif
#endif
