// xcorr_kernels.h -- host-visible launch interface of the device code
// (internal to libmuse_hip.so; the public ABI is include/muse_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "muse_hip.h"

namespace muse {

enum { KERNEL_GENERIC = 0, KERNEL_R16_OCC3 = 6, KERNEL_STOCKHAM = 10, KERNEL_R16_FOLD = 11, KERNEL_SMALL = 12, KERNEL_LONG = 13, KERNEL_REAL = 14, KERNEL_REAL_SPLIT = 15 };

struct FusedParams {
    const double *rows; // M x N row-major, row stride `stride` elements (float64 groups)
    const float *rows32; // the same for float32-storage groups (muse_group_create_f32); exactly one of the two is set
    long long M;
    long long stride;
    long long npairs; // ceil(M/2): one workgroup pass handles two series
    int N;            // series length
    int n;            // FFT length (power of two >= N)
    int logn;
    int normalize_y;     // 1: zNormalize each series (xCorrWithX, xCorr normalize=true)
    const double2 *xc;   // n entries: conj(X_full[f]) * scale
    const double2 *tw1;  // [16][256] W_4096^(k*t)      (tuned kernel)
    const double2 *tw2;  // [16][16]  W_256^(k*c)       (tuned kernel)
    const double2 *xcp;  // [16][256] xc[256*k + (t>>4) + 16*(t&15)]     (xcorr_r16_fold.hip: xc in lane order)
    // xcorr_r16_fold.hip: the eight per-thread factors of a generalised 16-point pass (fold_device.h), delta = u / 256
    const double2 *g2;   // [8][16]  pass 2: u = 16 j
    const double2 *g3a;  // [8][256] first transform, pass 3: u = (t >> 4) + 16 (t & 15)
    const double2 *g3b;  // [8][256] second transform, pass 3: u = t
    const double2 *gsmall; // [8][n/16] xcorr_small.hip (n = 512, 1024, 2048): last-pass factors, delta = j / (n/16), lane-ordered
    // xcorr_real.hip, the 16384-point transform as 16 x 1024 (wave-local 1024-point transforms around ONE workgroup transpose):
    const double2 *gsmall_b; // [8][64]: the 1024-point transform's last-pass factors (the context's table for n = 1024)
    const double2 *wsplit;   // [15][1024] W_16384^(j k1), k1 = 1 .. 15: the twiddles between the register pass and the wave-local transforms
    const double2 *xcw;      // [2][8][1024] xc at the bins of a thread's lower eight registers, k = w + 16 c + 1024 r (j = 64 w + c), and at M - k
    const double2 *twl;  // [4096] W_n^(m2): the base of the sweeps' twiddles W_n^(m2 k1) of the long-series kernel (xcorr_long.hip forms the powers)
    const double *c1;    // [n] (n = 4096 and the long-series kernel) N < n: correlation of the valid-sample indicator with the reference (xcorr_r16_fold.hip)
    // many references in one pass (xcorr_fused_n4096_multi): device arrays of R pointers
    int R;
    const double2 *const *xcp_many; // R lane-ordered spectrum tables
    double *const *mv_many;         // R result vectors (M doubles each)
    int *const *lag_many;           // R lag vectors (M ints each)
    const double *const *c1_many;   // N < 4096: R indicator-correlation tables
    const float2 *const *xcf_many;  // screening pass for R references: R fp32 spectrum tables
    unsigned *const *flags_many;    //   R flag vectors
    double *const *var_many;        //   R variance vectors
    double2 *zscratch;              // zslots x 4096 complex: one parked spectrum per (resident) workgroup
    int zslots;
    const double2 *twm;  // [32768]   W_65536^k         (generic kernel; half period)
    double2 *gscratch;   // n > 8192: one n-element complex work buffer per workgroup (global, L2-resident)
    double *mv;          // out: M signed max values
    int *lag;            // out: M lags
    double *cc_out;      // optional (generic kernel only): M x n correlations
    int *nil_out;        // optional (generic kernel only): M flags, 1 = sigma == 0 -> (nil,0,0)
    unsigned long long *dbg; // diagnostic builds only (tools/ablate): per-workgroup phase cycle sums
    // fp32 screening kernel (xcorr_r16_screen.hip)
    const float2 *twmf;  // [32768]   W_65536^k, fp32 (screening pass, n = 512 .. 2048)
    const float2 *tw1f;  // [16][256] W_4096^(k*t), fp32
    const float2 *tw2f;  // [16][16]  W_256^(k*c), fp32
    const float2 *xcf;   // n entries: conj(X_full[f]) / n, fp32
    const double *xs;    // n entries: zeroPad(zNormalize(ref)/(N-1), n), time domain, fp64
    double screen_delta; // candidate window below the fp32 maximum (scaled units, max |cc| <= 1)
    unsigned *scr_flags; // screening pass (xcorr_screen_pass_n4096): per-row SCR_* bits, OR-ed in
    double *scr_var;     // screening pass: per-row sample variance (the estimate in mv is cc32 * 2^e: score = mv / sqrt(var))
    int scr_max_lag;     // screening pass: the Run's MaxLag (classifies the possible argmax lags)
    int scr_need_sign;   // screening pass (n = 4096): 0 = the Run's filters never look at the sign of a score
    int *ovf_count;      // pairs with too many candidates: redone by the fp64 kernel
    int *work_counter;   // dynamic pair hand-out (xcorr_r16_fold.hip): zeroed before the launch
    long long *ovf_list;
    // optional indirection for the fp64 kernels: process pair_list[0 .. *pair_count)
    const long long *pair_list;
    const int *pair_count;
    long long dense_total; // xcorr_r16_occ4.hip behind the default n = 4096 kernel: > 0 and *pair_count * 8 > dense_total -> redo ALL dense_total pairs
    long long gscratch_slices; // n-element slices `gscratch` holds: a kernel that works in it launches no more workgroups than fit
    // two-sided xCorr (xcorr_two_sided.hip): pair i = (x_i = xrows + i xstride, length Nx; y_i = rows + i stride, length N)
    const double *xrows;
    long long xstride;
    int Nx;
    // 1 / N and 1 / (N - 1), formed on the host by the launchers (with_reciprocals): as kernel arguments they live in scalar
    // registers; formed by the kernel (N is a run-time value in the builds for zero-padded series) they come out of the vector
    // divider, and wave-uniform values in vector registers were what those builds parked in scratch
    double invN, invNm1;
};
inline FusedParams with_reciprocals(FusedParams p)
{
    p.invN = 1.0 / (double)p.N;
    p.invNm1 = 1.0 / (double)(p.N - 1);
    return p;
}

hipError_t launch_fused(const FusedParams &p, int variant, int num_cus, hipStream_t stream);
hipError_t launch_fused_occ4(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_r16_occ4.hip (rescaling / pair-list kernel)
hipError_t launch_screen_pass(const FusedParams &p, int num_cus, hipStream_t stream);  // xcorr_r16_screen.hip (filter-and-refine Run)
hipError_t launch_screen_pass_many(const FusedParams &p, int num_cus, hipStream_t stream); // the same for R references in one pass
hipError_t launch_screen_pass_stk(const FusedParams &p, int num_cus, hipStream_t stream);  // xcorr_screen_stk.hip: n = 512, 1024, 2048
// per-row flags of the screening pass
enum : unsigned { SCR_IN = 1u, SCR_OUT = 2u, SCR_POS = 4u, SCR_NEG = 8u, SCR_REFINE = 16u, SCR_NAN = 32u };
hipError_t launch_fused_fold(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_r16_fold.hip (n == 4096, default)
hipError_t launch_fused_multi(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_r16_fold.hip (R references)
hipError_t launch_fused_small(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_small.hip (n = 512, 1024, 2048: default)
hipError_t launch_fused_stockham(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_stockham.hip (n = 512 .. 2048, 8192 .. 65536)
hipError_t launch_fused_long(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_long.hip (n = 65536: default; 16384, 32768)
hipError_t launch_real_split_tables(const double2 *xc, double2 *out, hipStream_t stream); // xcorr_real.hip: FusedParams::xcw for n = 32768
hipError_t launch_real8k_tables(const double2 *xc, double2 *out, hipStream_t stream); // xcorr_real.hip: n = 8192, xc at the threads' bins, lane-ordered ([16][256])
hipError_t launch_two_sided_real(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_real.hip: the two-sided xCorr at n = 32768
hipError_t launch_fused_real_split(const FusedParams &p, int num_cus, hipStream_t stream); // the same, the 16384-point transform as 16 x 1024 (test hook 15)
hipError_t launch_fused_real(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_real.hip (n = 32768: one real series per workgroup on the 16384-point transform)
hipError_t launch_two_sided(const FusedParams &p, int num_cus, hipStream_t stream);
// the same for n = 512 ... 2048, 8192, 16384 on xcorr_small.hip's transforms (called by launch_two_sided)
hipError_t launch_two_sided_small(const FusedParams &p, int num_cus, hipStream_t stream); // xcorr_two_sided.hip (xCorr, n = 512 .. 65536)
// out[4096 k1 + 256 k + t] = in[k1 + R1 (256 k + (t >> 4) + 16 (t & 15))]: the spectrum rows of the long-series kernel in lane order
hipError_t launch_lane_order_rows(const double2 *in, double2 *out, int R1, hipStream_t stream);
// out[256 k + t] = in[256 k + (t >> 4) + 16 (t & 15)], k < 16: a 4096-entry table in the lane order of xcorr_r16_fold.hip
hipError_t launch_lane_order(const double2 *in, double2 *out, hipStream_t stream);
// c1[k] = sum_{j >= pad} xs[(j + k) mod n]: what a series of ones at the valid (non-pad) positions correlates to
hipError_t launch_indicator_corr(const double *xs, int n, int pad, double *c1, hipStream_t stream);
hipError_t launch_ref_spectrum(const double *ref_dev, int N, int n, int logn, int normalize, double x_scale,
                               double xc_scale, const double2 *twm, double2 *X, double2 *xc, float2 *xcf, double *xs,
                               double2 *gscratch, int *status, hipStream_t stream);
constexpr int SMALL_MAX_N = 16384;       // largest FFT length of xcorr_small.hip (it reads n - N samples in front of a row unclamped:
                                         // FusedParams::rows must carry that many readable elements in front of row 0, capi_group.hip GROUP_GUARD)
constexpr int GENERIC_LDS_MAX_N = 8192;  // larger n: the radix-2 passes run in gscratch
constexpr int GENERIC_MAX_N = 65536;
constexpr int GENERIC_GLOBAL_WGS_PER_CU = 2;
constexpr int STOCKHAM_GLOBAL_WGS_PER_CU = 2; // xcorr_fused_stk_4step (the long series' redo kernel): resident workgroups per CU, one n-element slice each
constexpr int LONG_WGS_PER_CU = 4;            // xcorr_long.hip and xcorr_two_sided_long: resident workgroups per CU, one n-element slice each
// n-element complex slices of the context's scratch buffer per CU (capi_batch.hip, ensure_gscratch): every kernel that works in it
// launches at most this many workgroups per CU times the slices each of them uses, and checks FusedParams::gscratch_slices
constexpr int GSCRATCH_SLICES_PER_CU = 4;
static_assert(LONG_WGS_PER_CU <= GSCRATCH_SLICES_PER_CU && GENERIC_GLOBAL_WGS_PER_CU <= GSCRATCH_SLICES_PER_CU &&
                  2 * STOCKHAM_GLOBAL_WGS_PER_CU <= GSCRATCH_SLICES_PER_CU,
              "the scratch buffer is sized for GSCRATCH_SLICES_PER_CU slices per CU");
hipError_t launch_direct(const double *x, int lenx, const double *y, int leny, int n, int normalize_x,
                         int normalize_y, double x_scale, double cc_scale, double *cc, int *lag, double *mv,
                         int *status, hipStream_t stream);
// two-sided xCorr, pairs whose statistics overflowed (xcorr_kernels.hip): what the reference's arithmetic gives for each listed
// pair (code 0 NaN stands / 1 every cc zero / 2 recompute on copies scaled by (scale.x, scale.y)), and those copies
hipError_t launch_two_sided_rescue(const double *xrows, long long xstride, int Nx, const double *yrows, long long ystride, int Ny,
                                   int n, int normalize, const long long *list, int count, int *code, double2 *scale,
                                   hipStream_t stream);
hipError_t launch_scale_listed_rows(const double *src, long long stride, int N, const long long *list, const double2 *g, int which,
                                    int count, double *dst, hipStream_t stream);
hipError_t launch_synth(double *rows, long long stride, long long first, long long count, long long global_first,
                        int N, unsigned long long seed, unsigned flags, hipStream_t stream);
hipError_t launch_synth_f32(float *rows, long long stride, long long first, long long count, long long global_first,
                            int N, unsigned long long seed, unsigned flags, hipStream_t stream);
hipError_t launch_synth_ref(double *ref, int N, unsigned long long seed, hipStream_t stream);

// measurement hook (diag_kernels.hip): one wave sampling delta s_memtime / delta s_memrealtime in windows of window_ms for total_ms
hipError_t launch_clock_probe(unsigned long long *out, int *count, int max_windows, double window_ms, double total_ms,
                              hipStream_t stream);
// unit-test hook: foldk::wave_argmax_store on 2 x 4096 given values (diag_kernels.hip); out24 = 4 waves x {max, value, index} x 2 series
hipError_t launch_wave_argmax_probe(const double *ccA, const double *ccB, double *out24, hipStream_t stream);

// ---- group max / filter / top-N (reduce_kernels.hip)
constexpr int TOPN_CHUNK = 4096;    // groups per workgroup in the selection pass
constexpr int TOPN_DEVICE_MAX = 256; // larger top_n: winners are copied to the host instead
constexpr int EXACT_FEED_MAX_GROUPS = 65536; // up to this many groups a Run feeds the top-N heap one Score per group in group order (the reference's feed)

struct GroupWork {
    unsigned long long *key; // [G] max of bits(|score|) over members (atomicMax)
    long long *first;        // [G] lowest member index (atomicMin)
    long long *win;          // [G] lowest member index attaining key (atomicMin)
};

struct SelectParams {
    const double *mv;
    const int *lag;
    long long M;
    const int *group_id; // device, or nullptr: every series its own group
    int G;
    int abs_scores;
    int max_lag;
    double threshold;
    int sign_filter;
    long long series_offset;
    const unsigned char *include; // optional: rows that may be selected (filter-and-refine Run); nullptr = all
    // 1: this device holds ONE SHARD of a group whose label groups may continue on other shards (muse_batch_run_groups):
    // rec[g] = the winner among the shard's members whose score is a number (series -1 if there is none), nothing is
    // filtered, and selkey[g] = the group's state on this shard: 0 no member, 1 members and the first one's score is a
    // number, 2 the first member's score is NaN (if it is the group's first member overall the group's score is NaN:
    // x > NaN never replaces it, muse_batch.go:87)
    int partial;
};

// filter-and-refine Run: what the screening pass left per row, the Run's filters and the error bound of the estimate
struct ScreenSelect {
    const double *mv;      // fp32 estimate of the signed value at the fp32 argmax, times sigma
    const double *var;     // sample variance (score estimate = mv / sqrt(var))
    const unsigned *flags; // SCR_* bits
    long long M;
    double threshold;
    int sign_filter;
    int abs_scores;
    double E;              // |estimate - exact| <= E (score units)
    const int *group_id;   // label groups (device), or nullptr: every series its own group
    int G;
};
// per-group scratch of the grouped filter-and-refine selection (G entries each)
struct ScreenGroupWork {
    long long *first;
    unsigned long long *glo, *gmay, *gkplus;
    int *gcert;
};
// pessimistic keys -> the top_n-th best of them (the cut) -> pairs whose optimistic key reaches it (pair_list,
// *pair_count, include[row] = 1); keys: screen_select_scratch(M, top_n) entries of scratch
long long screen_select_scratch(long long G, int top_n);
// run-time guard of the bound: estimates of the listed rows saved before the fp64 kernel overwrites them, then the largest
// | |estimate| - |fp64 score| | over those rows (as the bits of a non-negative double, atomicMax)
// guard sample: appends one pair in 1024 (hash of pair and salt) to the list and marks its rows re-evaluated
hipError_t launch_screen_sample(long long npairs, long long M, unsigned long long salt, long long *pair_list, int *pair_count,
                                unsigned char *include, hipStream_t stream);
hipError_t launch_screen_save(const ScreenSelect &q, const long long *pair_list, const int *pair_count, double *est_save,
                              hipStream_t stream);
hipError_t launch_screen_check(const double *mv, long long M, const long long *pair_list, const int *pair_count,
                               const double *est_save, unsigned long long *err_bits, hipStream_t stream);
hipError_t launch_screen_select(const ScreenSelect &q, int top_n, unsigned long long *selkey, unsigned long long *keys,
                                const ScreenGroupWork &gw, long long *pair_list, int *pair_count, unsigned char *include,
                                hipStream_t stream);

// per-group winners -> rec[G] (+ selection keys selkey[G]: 0 = filtered out)
hipError_t launch_group_reduce(const SelectParams &sp, const GroupWork &gw, muse_record *rec,
                               unsigned long long *selkey, hipStream_t stream);
// small Runs in one launch (reduce_kernels.hip, small_groups_kernel): what launch_group_reduce computes, one 32-byte slot of pinned
// host memory per label group, each with its own stamp (= token) stored last
constexpr int SMALL_GROUPS_MAX_G = 2048;        // label groups (three 8-byte work arrays in LDS)
constexpr long long SMALL_GROUPS_MAX_M = 32768; // series
constexpr int SMALL_UNGROUPED_MAX = 32768;      // series of a Run without a label map (one slot each, no work arrays)
struct SmallSlot {
    long long series; // muse_record::series, ::score, ::lag of group g (::group = g)
    double score;
    int lag;
    unsigned key;     // the selection key: 0 not selectable / empty; partial mode: the group's state 0 / 1 / 2
    unsigned long long stamp;
};
static_assert(sizeof(SmallSlot) == 32, "one slot = one 32-byte block");
hipError_t launch_small_groups(const SelectParams &sp, SmallSlot *out, unsigned long long token, hipStream_t stream);
// Run(nil) over many series: per chunk of TOPN_CHUNK series the best K candidates (what group_final_kernel + topn_kernel select,
// in the same order) written into pinned slots cand[chunk * K + r], their number into cnt[chunk] (each with its stamp)
constexpr int SMALL_DIRECT_MAX_SLOTS = 131072;  // 4 MB of pinned slots
struct CountSlot {
    unsigned long long count;
    unsigned long long stamp;
};
hipError_t launch_topn_ungrouped(const SelectParams &sp, int K, SmallSlot *cand, CountSlot *cnt, unsigned long long token,
                                 hipStream_t stream);
// one label group in one launch (Muse.Run): the winner record and the group's state (reduce_kernels.hip)
struct SingleGroupOut {
    muse_record rec;
    unsigned long long state;
};
hipError_t launch_single_group(const double *mv, const int *lag, long long M, int abs_scores, long long series_offset,
                               SingleGroupOut *out, hipStream_t stream);
// per-chunk top-K extraction: cand[nblocks*K], cnt[nblocks]
hipError_t launch_topn(const muse_record *rec, const unsigned long long *selkey, int G, int K, muse_record *cand,
                       int *cnt, hipStream_t stream);

} // namespace muse
