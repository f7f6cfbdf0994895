// xcorr_huge.h -- launch interface of xcorr_huge.hip (FFT lengths 2^17 ... 2^20); internal to libmuse_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace muse {

constexpr int HUGE_MIN_LOGN = 17, HUGE_MAX_LOGN = 20;
constexpr int HUGE_MAX_N = 1 << HUGE_MAX_LOGN;
constexpr size_t HUGE_BATCH_BYTES = (size_t)128 << 20; // work buffer of one batch: half the Infinity Cache

enum : unsigned {
    HUGE_STAGE_STATS = 1u,        // huge_stats (when normalize is set)
    HUGE_STAGE_SWEEP1 = 2u,       // rows -> Y
    HUGE_STAGE_ROWS = 4u,         // Y -> forward, times table, forward -> Y
    HUGE_STAGE_ROWS_FORWARD = 8u, // Y -> forward -> table_out (and X_out)
    HUGE_STAGE_SWEEP2 = 16u,      // Y -> cc -> amax (and cc_out)
    HUGE_STAGE_FINAL = 32u,       // amax -> mv, lag, nil
    HUGE_STAGE_STATS_ONLY = 64u,  // huge_stats + huge_norm alone (snorm and sfin of the batch), nothing else
};

struct HugeParams {
    // the batch: series first .. first + count - 1 of `rows` (row stride `stride`, length N); solo = 1: one series per transform,
    // 0: series 2 i and 2 i + 1 share one (the real and the imaginary part of one complex signal)
    const double *rows;
    long long stride;
    long long first;
    int count;
    int N;
    int solo;
    int normalize;    // zNormalize each series (xcorr.go:84-95); 0: raw samples (xCorr with normalize = false)
    double pre_scale; // multiplies the (normalised) samples: 1 / (N - 1) for a reference (muse_batch.go:42), else 1
    int n, logn, R1;
    const double2 *thi, *tlo; // W_n^(1024 j), j < n / 1024; W_n^j, j < 1024
    const double2 *g2, *g3a, *g3b; // the n = 4096 kernel's tables (row transforms)
    const double2 *table;     // HUGE_STAGE_ROWS: multiplier rows in lane order, pair i at table + i * table_stride (0: shared)
    long long table_stride;
    double2 *Y;               // work buffer: n complex per pair of the batch
    double *part;             // [series of the batch][R1][2] chunk sums
    double *snorm;            // [series of the batch][4]: first sample, mean of the shifted samples, pre_scale / sigma, flag
                              // (the all-scores pass of a group points it into the group's kept statistics: capi_huge.hip)
    double *sfin;             // [series of the batch] flag: 0 ok, 1 sigma == 0, 2 NaN / Inf statistics
    const double *sfin_x;     // two-sided: the flags of the batch's x series (same slots), or nullptr
    double *amax;             // [pair][R1][8] tile maxima
    double *mv;               // outputs, indexed by first + slot
    int *lag;
    int *nil;                 // optional
    double *cc_out;           // optional: (first + slot) * n + lag index
    double2 *table_out;       // HUGE_STAGE_ROWS_FORWARD: pair i at table_out + i * n
    double table_scale;
#ifdef MUSE_HUGE_ABL
    int abl;                  // diagnostic builds only (tools/ablate/ab_huge.sh): parts of sweep 1 left out / made contiguous
#endif
    double2 *X_out;           // optional: bins 0 .. n / 2 in natural order, pair i at X_out + i * (n / 2 + 1)
};

hipError_t launch_huge(const HugeParams &p, unsigned stages, hipStream_t stream);

} // namespace muse
