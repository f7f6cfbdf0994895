// xcorr_huge.hip -- series longer than 65 536 samples: FFT lengths n = 2^17 ... 2^20 (the reference has no length limit:
// /root/reference/xcorr.go:19-24 nextPowOf2, :160-197 xCorrWithX, muse_batch.go:33-37).
//
// A transform this long no longer fits a CU (n complex = 2 ... 16 MB), so one pair is not one workgroup any more: the
// four-step split of xcorr_long.hip is cut into KERNELS, each of them spread over the whole chip, and a BATCH of pairs is
// sized so that its work buffer (n complex per pair, 128 MB per batch) stays in the 256 MB Infinity Cache between them:
//
//   n = R1 * 4096 (R1 = 32 ... 256), input index m1 4096 + m2, spectrum index k1 + R1 k2, lag index l1 4096 + l2
//     Z[k1 + R1 k2]    = sum_m2 W_4096^(m2 k2) [ W_n^(m2 k1) sum_m1 W_R1^(m1 k1) z[m1 4096 + m2] ]
//     cc[l1 4096 + l2] = sum_k1 W_R1^(k1 l1)  [ W_n^(k1 l2) sum_k2 W_4096^(k2 l2) Z[k1 + R1 k2] xc[k1 + R1 k2] ]
//
//   huge_stats   per series and 4096-sample chunk: shifted sums (zNormalize, xcorr.go:84-95)                 grid R1 x series
//   huge_sweep1  per pair and tile of 4096 / R1 columns m2: rows -> z-normalised, leading zero pad (xcorr.go:176-181), two series
//                as re / im of one complex signal; R1-point DFT over m1 (radix 16 in registers, one LDS transpose, radix R1 / 16),
//                twiddle W_n^(m2 k1) -> Y[k1][m2]                                                              grid R1 x pairs
//   huge_rows    per pair and row k1: the n = 4096 kernel's pair of transforms (long_device.h, row_transforms) with the
//                reference spectrum row folded in between (conj, mult, xcorr.go:184-185), in place                 grid R1 x pairs
//   huge_sweep2  per pair and tile of columns l2: twiddle W_n^(k1 l2), R1-point DFT over k1 -> cc (xcorr.go:186-187),
//                the tile's first maximum of |cc| per series (maxAbsIndex, xcorr.go:39-50)                          grid R1 x pairs
//   huge_final   per pair: the tiles' maxima -> the first index of the greatest |cc|, lag unwrap (xcorr.go:189-194) grid pairs
//
// Unlike the kernels for n <= 65536 the series are normalised BEFORE the transform (the statistics have their own pass: the
// rows are read twice, the second time out of the Infinity Cache), which is the reference's own order of operations and has
// three consequences: the two series of a pair enter the shared transform at unit variance (no sigma-spread hand-off), a
// NaN / Inf series is replaced by zeros before it can poison its partner (its score is NaN: every cc is NaN in the
// reference), and a zero-padded series needs no indicator correction.  One code path serves Batch.Run / Muse.Run (pairs of
// series against the batch's shared spectrum table) and the two-sided xCorr (xcorr.go:102-153: every x its own table, made by
// huge_rows<FORWARD>, one series per transform).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "long_device.h"
#include "xcorr_huge.h"

namespace muse {

using namespace occ4;
using namespace fold;
using namespace foldk;
using namespace lng;

namespace {

constexpr int FLAG_OK = 0, FLAG_ZERO = 1, FLAG_NAN = 2;

// W_n^p, p < n, from the two-level tables (one complex multiplication: ~1.5 ulp on a unit-modulus factor)
__device__ __forceinline__ double2 twiddle(const double2 *__restrict__ thi, const double2 *__restrict__ tlo, const unsigned p)
{
    return cmul(thi[p >> 10], tlo[p & 1023u]);
}

// workgroup sum in a fixed shape (DPP butterfly per wave, the four waves in order): the same bits in every workgroup that adds
// the same values
__device__ __forceinline__ double block_sum_fixed(const double x, double *red4, const int t)
{
    const double w = wave_sum_dpp(x);
    __syncthreads();
    if ((t & 63) == 0)
        red4[t >> 6] = w;
    __syncthreads();
    return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// series `slot` of the batch: row pointer and whether it exists
__device__ __forceinline__ const double *series_row(const HugeParams &p, const long long slot)
{
    return p.rows + (p.first + slot) * p.stride;
}

// ------------------------------------------------------------------------------------------------ statistics
__global__ __launch_bounds__(256) void huge_stats(const HugeParams p)
{
    __shared__ double red4[4];
    const int t = threadIdx.x, chunk = blockIdx.x;
    const long long slot = blockIdx.y;
    const double *__restrict__ row = series_row(p, slot);
    const double K = row[0];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int e = chunk * 4096 + 256 * i + t;
        if (e < p.N) {
            const double d = row[e] - K;
            s1 += d;
            s2 = fma(d, d, s2);
        }
    }
    const double a = block_sum_fixed(s1, red4, t);
    const double b = block_sum_fixed(s2, red4, t);
    if (t == 0) {
        p.part[(slot * p.R1 + chunk) * 2 + 0] = a;
        p.part[(slot * p.R1 + chunk) * 2 + 1] = b;
    }
}

// the series' shift (first sample, mean of the shifted samples), 1 / sigma (times pre_scale) and flag from its chunk sums, once per
// series: sweep 1 reads the four values instead of every one of its R1 workgroups per pair reducing the chunk sums again
// (a prologue of eight barriers in front of the row requests of a workgroup that has only one neighbour on its CU)
__global__ __launch_bounds__(256) void huge_norm(const HugeParams p)
{
    __shared__ double red4[4];
    const int t = threadIdx.x;
    const long long slot = blockIdx.x;
    const double a = t < p.R1 ? p.part[(slot * p.R1 + t) * 2 + 0] : 0.0;
    const double b = t < p.R1 ? p.part[(slot * p.R1 + t) * 2 + 1] : 0.0;
    const double s1 = block_sum_fixed(a, red4, t), s2 = block_sum_fixed(b, red4, t);
    if (t == 0) {
        const double invN = 1.0 / (double)p.N, invNm1 = 1.0 / (double)(p.N - 1);
        bool zero, nan;
        const double var = variance(Stat{s1, s2}, invN, invNm1, zero, nan);
        const int flag = nan ? FLAG_NAN : zero ? FLAG_ZERO : FLAG_OK;
        double *__restrict__ o = p.snorm + slot * 4;
        o[0] = series_row(p, slot)[0];
        o[1] = s1 * invN;
        o[2] = flag == FLAG_OK ? p.pre_scale / sqrt(var) : 0.0;
        o[3] = (double)flag;
        p.sfin[slot] = (double)flag;
    }
}

__device__ __forceinline__ void series_norm(const HugeParams &p, const long long slot, double &K, double &mean, double &inv, int &flag)
{
    K = 0.0;
    mean = 0.0;
    inv = p.pre_scale;
    flag = FLAG_OK;
    if (!p.normalize)
        return;
    const double *__restrict__ o = p.snorm + slot * 4;
    K = uniform(o[0]);
    mean = uniform(o[1]);
    inv = uniform(o[2]);
    flag = __builtin_amdgcn_readfirstlane((int)o[3]);
}

// Workgroup -> (pair, tile) of the sweeps.  A tile touches 128 B (sweep 1's reads) or 256 B of EVERY row of its pair: on
// its own a scatter over R1 DRAM pages.  Workgroups are handed to the eight XCDs round-robin by their linear index, so the
// map gives each XCD a run of R1 / 8 neighbouring tiles of one pair: the workgroups resident on an XCD at one time then cover
// 4 ... 8 KB of every row between them (measured: profiles/r06_long_series.txt).
__device__ __forceinline__ void sweep_tile(const int R1, long long &pair, int &tl)
{
    const unsigned lin = blockIdx.x, xcd = lin & 7u, slot = lin >> 3;
    const unsigned run = (unsigned)R1 >> 3;
    pair = slot / run;
    tl = (int)(xcd * run + slot % run);
}

// ------------------------------------------------------------------------------------------------ sweep 1
// R = R1 / 16 (2, 4, 8, 16): the R1-point DFT over m1 = q + R i is a radix-16 over i in registers, the factor W_R1^(q ka), one LDS
// transpose and a radix-R over q; thread (c, q) <-> (column, residue) before the transpose, (c, j) <-> (column, the outputs
// ka = j Q .. j Q + Q - 1 of the first step, Q = 16 / R) behind it.
// The transpose moves the real parts, then the imaginary parts, through ONE 32 KB buffer: with all 64 KB of a tile resident a CU
// holds two workgroups, and a copy-only build of sweep 1 moves its 268 MB in 79 us at two workgroups per CU but in 56 us at three
// (tools/ablate/ab_huge.sh; profiles/r06_long_series.txt).
template <int R>
__device__ __forceinline__ void column_dft(double2 (&v)[16], double *tile, const double2 *__restrict__ thi, const int t, const int ABL = 0)
{
    constexpr int TW = 256 / R, Q = 16 / R;
    const int c = t % TW, q = t / TW;
    if (!(ABL & 16))
        dft16_nr(v); // u_q[ka] at v[BR16(ka)]
    if (!(ABL & 32)) {
#pragma unroll
        for (int ka = 1; ka < 16; ka++) // W_R1^(q ka) = W_n^(4096 q ka) = thi[4 q ka]
            v[BR16(ka)] = cmul(v[BR16(ka)], thi[4 * q * ka]);
    }
    const int j = q; // (the same split of t: column c, group j)
    double re[16];
#pragma unroll
    for (int ka = 0; ka < 16; ka++)
        tile[(ka * R + q) * TW + c] = v[BR16(ka)].x;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < Q; m++)
#pragma unroll
        for (int s = 0; s < R; s++)
            re[m + s * Q] = tile[((j * Q + m) * R + s) * TW + c];
    __syncthreads();
#pragma unroll
    for (int ka = 0; ka < 16; ka++)
        tile[(ka * R + q) * TW + c] = v[BR16(ka)].y;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < Q; m++)
#pragma unroll
        for (int s = 0; s < R; s++)
            v[m + s * Q] = make_double2(re[m + s * Q], tile[((j * Q + m) * R + s) * TW + c]);
    if (!(ABL & 16))
        sweep_dft<R>(v); // output kb of (ka = j Q + m) at v[m + brev<R>(kb) Q]: element k1 = ka + 16 kb
    __syncthreads();
}

#ifndef MUSE_HUGE_S1_OCC
#define MUSE_HUGE_S1_OCC 3
#endif
template <int R>
__global__ __launch_bounds__(256, MUSE_HUGE_S1_OCC) void huge_sweep1(const HugeParams p)
{
    constexpr int TW = 256 / R, Q = 16 / R, R1 = 16 * R;
    __shared__ double tile[4096];
    const int t = threadIdx.x;
    long long pair;
    int tl;
    sweep_tile(R1, pair, tl);
    const long long sA = p.solo ? pair : 2 * pair, sB = sA + 1;
    const bool hasB = !p.solo && sB < p.count;
    const double *__restrict__ ra = series_row(p, sA);
    const double *__restrict__ rb = series_row(p, hasB ? sB : sA);
    double KA, mA, iA, KB = 0.0, mB = 0.0, iB = 0.0;
    int fA, fB = FLAG_OK;
    series_norm(p, sA, KA, mA, iA, fA);
    if (hasB)
        series_norm(p, sB, KB, mB, iB, fB);
    const int c = t % TW, q = t / TW;
    const int m2 = tl * TW + c, pad = p.n - p.N;
    double2 v[16];
#ifdef MUSE_HUGE_ABL
    const int ABL = p.abl;
#else
    constexpr int ABL = 0;
#endif
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int e = (q + R * i) * 4096 + m2 - pad; // (leading zeros: xcorr.go:176-181)
        if (ABL & 4)
            e = tl * 4096 + 256 * i + t; // (diagnostic: the same bytes, contiguous per workgroup)
        double xa = 0.0, xb = 0.0;
        if (e >= 0) {
            xa = ((ra[e] - KA) - mA) * iA;
            if (hasB)
                xb = ((rb[e] - KB) - mB) * iB;
        }
        // (a flagged series enters as zeros: 1 / sigma is 0 for it, but NaN * 0 and Inf * 0 are NaN)
        v[i] = make_double2(fA == FLAG_OK ? xa : 0.0, fB == FLAG_OK ? xb : 0.0);
    }
    if (!(ABL & 8))
        column_dft<R>(v, tile, p.thi, t, ABL);
    double2 *__restrict__ Y = p.Y + (size_t)pair * (size_t)p.n;
    const int j = q;
#pragma unroll
    for (int m = 0; m < Q; m++)
#pragma unroll
        for (int kb = 0; kb < R; kb++) {
            const int k1 = j * Q + m + 16 * kb;
            const double2 z = v[m + brev<R>(kb) * Q];
            const double2 zz = (k1 == 0 || (ABL & 1)) ? z : cmul(z, twiddle(p.thi, p.tlo, (unsigned)(m2 * k1)));
            if (ABL & 2)
                Y[(size_t)tl * 4096 + 256 * (m * R + kb) + t] = zz; // (diagnostic: the same bytes, contiguous per workgroup)
            else
                Y[(size_t)k1 * 4096 + m2] = zz;
        }
}

// ------------------------------------------------------------------------------------------------ rows
// FORWARD: the row's spectrum as the multiplier table of another pass (the reference of a batch, the x of a two-sided pair):
// table[k1 4096 + 256 b + t] = conj(spectrum) * table_scale in the lane order row_transforms reads (xc for k3 = b at 256 b + t:
// the bin k1 + R1 (256 b + (t >> 4) + 16 (t & 15))), and optionally the bins 0 .. n / 2 in natural order (muse_batch_spectrum).
template <bool FORWARD>
__global__ __launch_bounds__(256, 3) void huge_rows(const HugeParams p) // (at four workgroups per CU the pair of transforms parks 13 registers in scratch)
{
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + XW * wave;
    const int k1 = blockIdx.x;
    const long long pair = blockIdx.y;
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();
    // (addresses as in xcorr_long.hip: a scalar base formed where it is used plus one 32-bit lane offset -- nothing 64-bit per
    // lane beside the sixteen complex registers of the transforms)
    typedef d2v __attribute__((address_space(1))) *gd2;
    const auto opaque = [](int x) __attribute__((always_inline)) {
        asm volatile("" : "+v"(x));
        return x;
    };
    double2 *const row = uniform_ptr(p.Y + (size_t)pair * (size_t)p.n + (size_t)k1 * 4096);
    double2 v[16];
    {
        const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const d2v z = *((gd2)scalar_ptr_at(row, 256 * i) + tl);
            v[i] = make_double2(z.x, z.y);
        }
    }
    if (FORWARD) {
        row_forward(v, xbuf, xw, g2s, p.g3a, t, wave, false);
        double2 *__restrict__ out = p.table_out + (size_t)pair * (size_t)p.n + (size_t)k1 * 4096;
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const double2 z = v[BR16(b)];
            out[256 * b + t] = make_double2(z.x * p.table_scale, -z.y * p.table_scale);
            if (p.X_out) {
                const long long k = k1 + (long long)p.R1 * (256 * b + (t >> 4) + 16 * (t & 15));
                if (k <= p.n / 2)
                    p.X_out[(size_t)pair * (size_t)(p.n / 2 + 1) + (size_t)k] = z;
            }
        }
    } else {
        const double2 *const xrow = uniform_ptr(p.table + (size_t)pair * (size_t)p.table_stride + (size_t)k1 * 4096);
        row_transforms(v, xbuf, xw, g2s, p.g3a, p.g3b, xrow, opaque(t), wave, false);
        const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
        for (int m = 0; m < 16; m++)
            *((gd2)scalar_ptr_at(row, 256 * m) + tl) = d2v{v[BR16(m)].x, v[BR16(m)].y};
    }
}

// ------------------------------------------------------------------------------------------------ sweep 2
#ifndef MUSE_HUGE_S2_OCC
#define MUSE_HUGE_S2_OCC 3
#endif
template <int R>
__global__ __launch_bounds__(256, MUSE_HUGE_S2_OCC) void huge_sweep2(const HugeParams p)
{
    constexpr int TW = 256 / R, Q = 16 / R;
    __shared__ double tile[4096];
    __shared__ double redm[8];
    __shared__ int redi[8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    long long pair;
    int tl;
    sweep_tile(16 * R, pair, tl);
    const long long sA = p.solo ? pair : 2 * pair, sB = sA + 1;
    const bool hasB = !p.solo && sB < p.count;
    const int c = t % TW, q = t / TW;
    const int l2 = tl * TW + c;
    const double2 *__restrict__ Y = p.Y + (size_t)pair * (size_t)p.n;
    double2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k1 = q + R * i;
        const double2 z = Y[(size_t)k1 * 4096 + l2];
        v[i] = k1 == 0 ? z : cmul(z, twiddle(p.thi, p.tlo, (unsigned)(k1 * l2)));
    }
    column_dft<R>(v, tile, p.thi, t);
    // the thread's sixteen lags in ascending order (l1 = j Q + m + 16 kb): strictly greater keeps the first
    const int j = q;
    double ba = 0.0, bb = 0.0, sa = 0.0, sb = 0.0;
    int ia = 0x7fffffff, ib = 0x7fffffff;
#pragma unroll
    for (int kb = 0; kb < R; kb++)
#pragma unroll
        for (int m = 0; m < Q; m++) {
            const int l1 = j * Q + m + 16 * kb;
            const double2 z = v[m + brev<R>(kb) * Q];
            const int idx = l1 * 4096 + l2;
            if (p.cc_out) {
                p.cc_out[(size_t)(p.first + sA) * (size_t)p.n + (size_t)idx] = z.x;
                if (hasB)
                    p.cc_out[(size_t)(p.first + sB) * (size_t)p.n + (size_t)idx] = z.y;
            }
            if (fabs(z.x) > ba) {
                ba = fabs(z.x);
                sa = z.x;
                ia = idx;
            }
            if (fabs(z.y) > bb) {
                bb = fabs(z.y);
                sb = z.y;
                ib = idx;
            }
        }
    const double cc0a = v[0].x, cc0b = v[0].y; // (tile 0, thread 0: lag index 0)
    // the tile's first maximum per series
    const double wa = wave_max(ba), wb = wave_max(bb);
    if (lane == 0) {
        redm[wave] = wa;
        redm[4 + wave] = wb;
    }
    __syncthreads();
    const double MA = fmax(fmax(redm[0], redm[1]), fmax(redm[2], redm[3]));
    const double MB = fmax(fmax(redm[4], redm[5]), fmax(redm[6], redm[7]));
    int ca = (ba == MA && MA > 0.0) ? ia : 0x7fffffff;
    int cb = (bb == MB && MB > 0.0) ? ib : 0x7fffffff;
    ca = wave_min_i(ca);
    cb = wave_min_i(cb);
    if (lane == 0) {
        redi[wave] = ca;
        redi[4 + wave] = cb;
    }
    __syncthreads();
    const int IA = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
    const int IB = min(min(redi[4], redi[5]), min(redi[6], redi[7]));
    double *__restrict__ out = p.amax + ((size_t)pair * (size_t)p.R1 + (size_t)tl) * 8;
    if (IA == 0x7fffffff ? t == 0 : (ia == IA && ba == MA)) {
        out[0] = IA == 0x7fffffff ? 0.0 : MA;
        out[1] = IA == 0x7fffffff ? 0.0 : sa;
        out[2] = IA == 0x7fffffff ? 1e300 : (double)IA;
    }
    if (IB == 0x7fffffff ? t == 0 : (ib == IB && bb == MB)) {
        out[3] = IB == 0x7fffffff ? 0.0 : MB;
        out[4] = IB == 0x7fffffff ? 0.0 : sb;
        out[5] = IB == 0x7fffffff ? 1e300 : (double)IB;
    }
    if (t == 0) {
        out[6] = cc0a;
        out[7] = cc0b;
    }
}

// ------------------------------------------------------------------------------------------------ final
__global__ __launch_bounds__(256) void huge_final(const HugeParams p)
{
    __shared__ double redm[4];
    __shared__ double redx[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long pair = blockIdx.x;
    const long long sA = p.solo ? pair : 2 * pair;
    const int nser = (p.solo || sA + 1 >= p.count) ? 1 : 2;
    for (int s = 0; s < nser; s++) {
        const double *__restrict__ in = p.amax + ((size_t)pair * (size_t)p.R1 + (size_t)(t < p.R1 ? t : 0)) * 8 + 3 * s;
        const double m = t < p.R1 ? in[0] : 0.0, sv = t < p.R1 ? in[1] : 0.0, ix = t < p.R1 ? in[2] : 1e300;
        const double wm = wave_max(m);
        __syncthreads();
        if (lane == 0)
            redm[wave] = wm;
        __syncthreads();
        const double M = fmax(fmax(redm[0], redm[1]), fmax(redm[2], redm[3]));
        double cand = (m == M && M > 0.0) ? ix : 1e300;
        cand = -wave_max(-cand); // (indices are exact in a double: the minimum is the first index)
        if (lane == 0)
            redx[wave] = cand;
        __syncthreads();
        const double I = fmin(fmin(redx[0], redx[1]), fmin(redx[2], redx[3]));
        const bool none = !(I < 1e299);
        if (none ? t == 0 : (t < p.R1 && ix == I && m == M)) {
            const long long series = p.first + sA + s;
            const int idx = none ? 0 : (int)I;
            double mv = none ? p.amax[(size_t)pair * (size_t)p.R1 * 8 + 6 + s] : sv; // nothing above 0: index 0, mv = cc[0]
            int lag = idx > p.n / 2 ? idx - p.n : idx;                                 // xcorr.go:192-194
            int flag = p.normalize ? (int)p.snorm[(sA + s) * 4 + 3] : FLAG_OK;
            const int fx = p.sfin_x ? (int)p.sfin_x[sA + s] : FLAG_OK;                 // two-sided: the pair's x
            if (fx == FLAG_ZERO || (fx == FLAG_OK && flag == FLAG_ZERO)) {             // xcorr.go:107-127, 164-172: (nil, 0, 0)
                mv = 0.0;
                lag = 0;
                flag = FLAG_ZERO;
            } else if (fx == FLAG_NAN || flag == FLAG_NAN) {                           // every cc is NaN in the reference
                mv = __builtin_nan("");
                lag = 0;
                flag = FLAG_NAN;
            }
            p.mv[series] = mv;
            p.lag[series] = lag;
            if (p.nil)
                p.nil[series] = flag == FLAG_ZERO ? 1 : 0;
        }
        __syncthreads();
    }
}

template <int R>
hipError_t launch_sweeps(const HugeParams &p, const int which, const dim3 grid2, hipStream_t stream)
{
    const dim3 grid(grid2.x * grid2.y); // (one dimension: sweep_tile maps the linear index)
    if (which == 1)
        hipLaunchKernelGGL((huge_sweep1<R>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((huge_sweep2<R>), grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

} // namespace

// one batch: p.count series (solo: one per transform; else two), stages selected by `stages` (HUGE_STAGE_* bits)
hipError_t launch_huge(const HugeParams &p, const unsigned stages, hipStream_t stream)
{
    if (stages == HUGE_STAGE_STATS_ONLY) {
        if (!p.rows || !p.part || !p.snorm || !p.sfin || p.count < 1 || p.N < 2 || p.R1 != p.n / 4096)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(huge_stats, dim3((unsigned)p.R1, (unsigned)p.count), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(huge_norm, dim3((unsigned)p.count), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if (p.logn < 17 || p.logn > HUGE_MAX_LOGN || p.n != (1 << p.logn) || p.R1 != p.n / 4096 || p.count < 1 || p.N < 2 || p.N > p.n ||
        !p.rows || !p.Y || !p.thi || !p.tlo || !p.part || !p.sfin)
        return hipErrorInvalidValue;
    const int pairs = p.solo ? p.count : (p.count + 1) / 2;
    const dim3 gp((unsigned)p.R1, (unsigned)pairs);
    hipError_t e = hipSuccess;
    if ((stages & HUGE_STAGE_STATS) && p.normalize) {
        if (!p.snorm)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(huge_stats, dim3((unsigned)p.R1, (unsigned)p.count), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(huge_norm, dim3((unsigned)p.count), dim3(256), 0, stream, p);
        e = hipGetLastError();
    } else if ((stages & HUGE_STAGE_STATS) && p.sfin) { // raw samples: every flag "ok" (huge_final reads none of them, a later batch's x flags might)
        e = hipMemsetAsync(p.sfin, 0, (size_t)p.count * sizeof(double), stream);
    }
    for (int which = 1; which <= 2 && e == hipSuccess; which++) {
        if (which == 1 ? !(stages & HUGE_STAGE_SWEEP1) : !(stages & HUGE_STAGE_SWEEP2))
            goto rows;
        if (which == 2 && !p.amax)
            return hipErrorInvalidValue;
        switch (p.R1) {
        case 32: e = launch_sweeps<2>(p, which, gp, stream); break;
        case 64: e = launch_sweeps<4>(p, which, gp, stream); break;
        case 128: e = launch_sweeps<8>(p, which, gp, stream); break;
        case 256: e = launch_sweeps<16>(p, which, gp, stream); break;
        default: return hipErrorInvalidValue;
        }
    rows:
        if (which == 1 && e == hipSuccess) {
            if (stages & HUGE_STAGE_ROWS_FORWARD) {
                if (!p.table_out || !p.g2 || !p.g3a)
                    return hipErrorInvalidValue;
                hipLaunchKernelGGL((huge_rows<true>), gp, dim3(256), 0, stream, p);
                e = hipGetLastError();
            } else if (stages & HUGE_STAGE_ROWS) {
                if (!p.table || !p.g2 || !p.g3a || !p.g3b)
                    return hipErrorInvalidValue;
                hipLaunchKernelGGL((huge_rows<false>), gp, dim3(256), 0, stream, p);
                e = hipGetLastError();
            }
        }
    }
    if (e == hipSuccess && (stages & HUGE_STAGE_FINAL)) {
        if (!p.mv || !p.lag || !p.amax)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(huge_final, dim3((unsigned)pairs), dim3(256), 0, stream, p);
        e = hipGetLastError();
    }
    return e;
}

} // namespace muse
