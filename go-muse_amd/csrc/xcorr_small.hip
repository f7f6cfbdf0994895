// xcorr_small.hip -- fp64 kernels for the FFT lengths n = 512, 1024, 2048, 8192, 16384 (every config-5 length whose pair of
// series fits one workgroup's registers; n = 4096 has its own kernel, xcorr_r16_fold.hip), round 2.
//
// Mathematics: xCorrWithX, /root/reference/xcorr.go:160-197, two series per complex transform, radix-16 Stockham
// passes as in xcorr_stockham.hip (n = R1 * 16^(P-1): P = 3 with R1 = 2, 4, 8 for n <= 2048, P = 4 with R1 = 2, 4 for
// n = 8192, 16384; every thread owns the 16 points x[j + i S], S = n / 16; S threads per pair).  What is different from
// the round-1 kernels of these lengths:
//   * Occupancy.  Round 1 kept a full padded work buffer per pair (n complex = 69.6 KB per 256-thread workgroup):
//     two workgroups = 8 waves per CU, and every profile said the kernels were bound by that, not by arithmetic
//     (profiles/r01_sizes_*: 24-27 % of the HBM roofline).  Here every transpose runs in TWO HALF ROUNDS through a
//     buffer of n / 2 points (8.7 KB per wave: 16 waves per CU at 128 VGPRs for every length), and in all transposes
//     EVERY lane reads eight values per round (no idle half as in the n = 4096 kernel's wave-local transposes):
//       A (after the radix-R1 pass): writer j, output (m, r) -> position (j + m S) R1 + r; reader j reads j + i S.
//         Round h carries the positions [8 S h, 8 S (h + 1)): the outputs m in [h Q1/2, (h+1) Q1/2) of every lane, read
//         back as the inputs i in [8 h, 8 h + 8) of every lane.
//       B (between radix-16 passes, Ns -> 16 Ns): writer j = g Ns + m, output r -> position g 16 Ns + r Ns + m.
//         Round h: the lower / upper half of the columns write all sixteen outputs; every lane reads its inputs
//         i in [8 h, 8 h + 8).  Where both halves sit in one wave (n = 512, 1024) they first trade eight registers
//         (v_permlane16/32_swap) so that every lane stores eight values in each round (level()).
//   * No workgroup barrier for n <= 1024: a pair lives inside one wave (n = 512: two pairs per wave) and LDS
//     operations of one wave execute in order.  n >= 2048: a pair spans 2, 8 or 16 waves = the workgroup, the half rounds
//     are separated by workgroup barriers; its reductions share one exchange and one barrier per kind (pair_sum4 ...).
//   * Arithmetic: every radix-16 pass is a generalised 16-point transform with the twiddles folded into the
//     butterflies (fold_device.h: 192 instructions and eight table entries per pass instead of 264 and four), and the
//     second transform is the forward algorithm again (xcorr_stockham.hip, lds_transforms).
//   * Tables: pass 2's 8 x R1 factors from an LDS copy (a broadcast read; gathered per lane out of the W_65536 table they
//     were the longest stall of the kernel), the later passes' from lane-ordered per-length tables (coalesced).
//   * Lanes are relabelled to columns so that LDS read groups and write groups meet no bank conflicts (column_of_lane()).
//   * MULTI: R references in one pass (muse_batch_score_many): the pair's spectrum stays in registers (n <= 8192: 256 VGPRs, half
//     the resident waves; round 2 parked it in global scratch: x 1.4 per reference at R = 8, now x 1.5 - 1.8) or is parked per
//     workgroup (n = 16384), and every reference takes product, second transform and argmax from there.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "fold_device.h"
#include "r16_device.h"
#include "two_device.h"

namespace muse {

namespace small {

using namespace occ4;
using namespace fold;

constexpr int padk(int x) { return x + (x >> 4); }
// the values of the two 16-lane rows of a 32-lane pair side by side (lane i of row 0 with lane i of row 1), in both rows:
// v_permlane16_swap(v, v) leaves [row 0, row 0, row 2, row 2] and [row 1, row 1, row 3, row 3] -- one VALU instruction per
// dword where __shfl_xor(v, 16) is a ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ void rows_side_by_side(const int v, int &even, int &odd)
{
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    even = (int)r[0];
    odd = (int)r[1];
}
__device__ __forceinline__ void rows_side_by_side(const double v, double &even, double &odd)
{
    int el, eh, ol, oh;
    rows_side_by_side(__double2loint(v), el, ol);
    rows_side_by_side(__double2hiint(v), eh, oh);
    even = __hiloint2double(eh, el);
    odd = __hiloint2double(oh, ol);
}

// ---- reductions over the S lanes of a pair; every lane of the pair gets the result.  S > 64 (one pair per workgroup,
// S / 64 waves): through `red`, S / 64 doubles of LDS, two workgroup barriers.
template <int S>
__device__ __forceinline__ double pair_sum(double v, double *red, const int wave)
{
    if (S == 32) {
        v += dpp_f64<0xB1>(v);
        v += dpp_f64<0x4E>(v);
        v += dpp_f64<0x141>(v);
        v += dpp_f64<0x140>(v); // the lane's 16-lane row
        double e, o;
        rows_side_by_side(v, e, o);
        return e + o;
    }
    v = wave_sum_dpp(v);
    if (S > 64) {
        lds_barrier();
        red[wave] = v;
        lds_barrier();
        v = red[0];
#pragma unroll
        for (int w = 1; w < S / 64; w++)
            v += red[w];
    }
    return v;
}
template <int S>
__device__ __forceinline__ double pair_max(double v, double *red, const int wave)
{
    if (S == 32) {
        v = fmax(v, dpp_f64<0xB1>(v));
        v = fmax(v, dpp_f64<0x4E>(v));
        v = fmax(v, dpp_f64<0x141>(v));
        v = fmax(v, dpp_f64<0x140>(v));
        double e, o;
        rows_side_by_side(v, e, o);
        return fmax(e, o);
    }
    v = wave_max_nonneg(v); // (|cc| maxima: never negative)
    if (S > 64) {
        lds_barrier();
        red[wave] = v;
        lds_barrier();
        v = red[0];
#pragma unroll
        for (int w = 1; w < S / 64; w++)
            v = fmax(v, red[w]);
    }
    return v;
}
template <int S>
__device__ __forceinline__ int pair_min_i(int v, double *red, const int wave)
{
    if (S == 32) {
        v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
        v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
        v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
        v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
        int e, o;
        rows_side_by_side(v, e, o);
        return min(e, o);
    }
    v = wave_min_i_dpp(v);
    if (S > 64) {
        int *ri = (int *)red;
        lds_barrier();
        ri[wave] = v;
        lds_barrier();
        v = ri[0];
#pragma unroll
        for (int w = 1; w < S / 64; w++)
            v = min(v, ri[w]);
    }
    return v;
}

// S > 64 (the pair's S / 64 waves are the workgroup): several reductions through ONE exchange and ONE workgroup barrier each.
// `red` holds three regions of 4 x 16 doubles (sums, maxima, indices) that are never reused before the many barriers of the
// transforms in between have passed.  (Separate pair_sum / pair_max / pair_min_i calls cost two barriers apiece: 16 per pair.)
template <int S>
__device__ __forceinline__ void pair_sum4(double &q0, double &q1, double &q2, double &q3, double *red, const int wave)
{
    if (S <= 64) {
        q0 = pair_sum<S>(q0, red, wave);
        q1 = pair_sum<S>(q1, red, wave);
        q2 = pair_sum<S>(q2, red, wave);
        q3 = pair_sum<S>(q3, red, wave);
        return;
    }
    constexpr int NW = S / 64;
    const double w0 = wave_sum_dpp(q0), w1 = wave_sum_dpp(q1), w2 = wave_sum_dpp(q2), w3 = wave_sum_dpp(q3);
    red[wave] = w0; // (every lane of the wave stores the same value)
    red[16 + wave] = w1;
    red[32 + wave] = w2;
    red[48 + wave] = w3;
    lds_barrier();
    q0 = red[0];
    q1 = red[16];
    q2 = red[32];
    q3 = red[48];
    // the partials in batches of four waves, each batch read and added before the next is requested: left to the compiler the
    // 4 x NW partials stay live far into the statistics -- at n = 16384 (NW = 16) up to 128 registers beside the 64 the pair's
    // samples occupy, and the kernel parked 28-49 registers per lane in scratch around this barrier (1.5 GB of scratch writes
    // per launch)
#pragma unroll
    for (int w0 = 1; w0 < NW; w0 += 4) {
        fence();
#pragma unroll
        for (int w = w0; w < w0 + 4 && w < NW; w++) {
            q0 += red[w];
            q1 += red[16 + w];
            q2 += red[32 + w];
            q3 += red[48 + w];
        }
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)); // (the sums are formed HERE: left alone, the adds of the second
                                                                   // series sink below the first one's variance, their partials live)
        fence();
    }
}
template <int S>
__device__ __forceinline__ void pair_max2(double &a, double &b, double *red, const int wave)
{
    if (S <= 64) {
        a = pair_max<S>(a, red, wave);
        b = pair_max<S>(b, red, wave);
        return;
    }
    constexpr int NW = S / 64;
    const double wa = wave_max_nonneg(a), wb = wave_max_nonneg(b);
    red[64 + wave] = wa;
    red[80 + wave] = wb;
    lds_barrier();
    a = red[64];
    b = red[80];
#pragma unroll
    for (int w = 1; w < NW; w++) {
        a = fmax(a, red[64 + w]);
        b = fmax(b, red[80 + w]);
    }
}
template <int S>
__device__ __forceinline__ void pair_min_i2(int &a, int &b, double *red, const int wave)
{
    if (S <= 64) {
        a = pair_min_i<S>(a, red, wave);
        b = pair_min_i<S>(b, red, wave);
        return;
    }
    constexpr int NW = S / 64;
    int *ri = (int *)(red + 96);
    const int wa = wave_min_i_dpp(a), wb = wave_min_i_dpp(b);
    ri[wave] = wa;
    ri[16 + wave] = wb;
    lds_barrier();
    a = ri[0];
    b = ri[16];
#pragma unroll
    for (int w = 1; w < NW; w++) {
        a = min(a, ri[w]);
        b = min(b, ri[16 + w]);
    }
}

// plain radix-R DFTs on the registers m + s (16 / R), natural order in place (as xcorr_stockham.hip, dft_small)
__device__ __forceinline__ void r_dft2(double2 &a, double2 &b) { bf_one(a, b); }
__device__ __forceinline__ void r_dft4(double2 &a, double2 &b, double2 &c, double2 &d)
{
    bf_one(a, c); // (a + c, a - c)
    bf_one(b, d); // (b + d, b - d)
    bf_one(a, b); // X0 = a, X2 = b
    bf_mi(c, d);  // X1 = c = (a-c) - i (b-d), X3 = d
    const double2 t = b;
    b = c;
    c = t; // natural order: a = X0, b = X1, c = X2, d = X3
}
__device__ __forceinline__ void r_dft8(double2 &x0, double2 &x1, double2 &x2, double2 &x3, double2 &x4, double2 &x5,
                                       double2 &x6, double2 &x7)
{
    bf_one(x0, x4);
    bf_one(x1, x5);
    bf_one(x2, x6);
    bf_one(x3, x7); // x0..x3 = sums (even outputs), x4..x7 = differences (odd outputs, to be twiddled by W8^k)
    // even half: DFT4 of (x0, x1, x2, x3) -> X0, X2, X4, X6
    bf_one(x0, x2);
    bf_one(x1, x3);
    bf_one(x0, x1); // x0 = X0, x1 = X4
    bf_mi(x2, x3);  // x2 = X2, x3 = X6
    // odd half: DFT4 of (x4, W8 x5, -i x6, W8^3 x7) -> X1, X3, X5, X7, the twiddles folded into the butterflies
    bf_mi(x4, x6);    // x4 = d0 - i d2, x6 = d0 + i d2
    bf_mi(x5, x7);    // x5 = d1 - i d3, x7 = d1 + i d3       (W8 d1 + W8^3 d3 = W8 (d1 - i d3))
    bf_w8(x4, x5);    // x4 = X1 = (d0 - i d2) + W8 (d1 - i d3),  x5 = X5
    bf_w8_mi(x6, x7); // x6 = X3 = (d0 + i d2) + W8^3 (d1 + i d3), x7 = X7
    // natural order
    const double2 X0 = x0, X4 = x1, X2 = x2, X6 = x3, X1 = x4, X5 = x5, X3 = x6, X7 = x7;
    x0 = X0; x1 = X1; x2 = X2; x3 = X3; x4 = X4; x5 = X5; x6 = X6; x7 = X7;
}
template <int R>
__device__ __forceinline__ void r_dft(double2 (&v)[16])
{
    constexpr int Q = 16 / R;
#pragma unroll
    for (int m = 0; m < Q; m++) {
        if (R == 2)
            r_dft2(v[m], v[m + Q]);
        else if (R == 4)
            r_dft4(v[m], v[m + Q], v[m + 2 * Q], v[m + 3 * Q]);
        else
            r_dft8(v[m], v[m + Q], v[m + 2 * Q], v[m + 3 * Q], v[m + 4 * Q], v[m + 5 * Q], v[m + 6 * Q], v[m + 7 * Q]);
    }
}

// generalised radix-16 pass with phase delta = m / NS, factors from the W_65536 half-period table (xcorr_stockham.hip, fwd16g)
template <int NS>
__device__ __forceinline__ double2 tw_factor(const double2 *__restrict__ twm, const int m, const int s)
{
    constexpr int U = 4096 / NS;
    const int idx = s == 0 ? 8 * U * m : s == 1 ? 4 * U * m : s == 2 ? 2 * U * m : s == 3 ? 2 * U * m + 8192
                                                                                           : U * m + 4096 * (s - 4);
    return ldg2u(scalar_ptr(twm), (unsigned)idx); // scalar base + UNSIGNED 32-bit lane offset: no 64-bit address arithmetic
}

// generalised pass whose first four factors were requested earlier (before the transpose that precedes the pass: their L2
// latency then runs under the transpose instead of in front of the first butterfly); the other four are requested behind
// the second stage
template <typename F>
__device__ __forceinline__ void gpass_pre(double2 (&v)[16], const double2 (&ga)[4], F fetch)
{
    double2 gb[4];
    gdft16_nr_s12(v, ga[0], ga[1]);
    fence();
#pragma unroll
    for (int s = 0; s < 4; s++)
        gb[s] = fetch(4 + s);
    fence();
    gdft16_nr_s3(v, ga[2], ga[3]);
    gdft16_nr_s4(v, gb[0], gb[1], gb[2], gb[3]);
}

// forward transform of the pair's n points: v[i] = x[j + i S] -> X[j + r S] at v[BR16(r)].  b: the pair's half buffer.
// gs: the last pass's eight factors per thread, lane-ordered [8][S] (FusedParams::gsmall): coalesced 16-byte loads --
// out of the generic W_65536 table the same factors are 64 different cache lines per wave instruction.
// One more level: transpose B (Ns -> 16 Ns) in two half rounds, then the generalised pass with phase (j mod 16 Ns) / (16 Ns)
// whose factors come lane-ordered from `tab` ([8][16 Ns], index j mod 16 Ns).  The lanes of the lower half of the pair
// (j < S / 2) write all sixteen outputs in round 0, the others in round 1; everybody reads 8 + 8.
template <int S, int NS, typename SYNC>
__device__ __forceinline__ void level(double2 (&v)[16], double2 *b, const double2 *__restrict__ tab, const int j, const int rbase,
                                      SYNC sync)
{
    constexpr int NS2 = 16 * NS;
    double2 w[16], ga[4];
    const unsigned m3 = (unsigned)(j & (NS2 - 1));
#pragma unroll
    for (int s = 0; s < 4; s++) // the pass's first factors travel during the transpose
        ga[s] = ldg2u(scalar_ptr_at(tab, s * NS2), m3);
    fence();
    const int g = j / NS, mm = j & (NS - 1);
    // position (g mod S/(2 NS)) 16 NS + r NS + m, padded
    const int gl = g & (S / (2 * NS) - 1);
    const int wb = gl * padk(16 * NS) + mm + (NS >= 16 ? (mm >> 4) : 0);
    const bool lower = j < S / 2;
    if constexpr (S <= 64 && NS < 16) {
        // Both halves of the writers sit in ONE wave (S = 64: lanes 0-31 / 32-63; S = 32: rows 0, 2 / 1, 3 of two pairs): masked,
        // each half round would cost sixteen ds_write_b128 whose price (the 13-cycle register transfer) does not shrink with the
        // mask.  Instead the halves trade registers (v_permlane32_swap / v_permlane16_swap, gfx950): outputs r + 8 of the lower
        // columns move to the upper lanes and outputs r of the upper columns to the lower lanes, so that EVERY lane writes
        // eight values per round -- 16 full stores instead of 32 half-empty ones on the one LDS pipe the CU's 16 waves share,
        // for 32 VALU swaps on a SIMD that has slack.  The slot of a value only depends on its column modulo S / 2 and on
        // its output index: both rounds use the same per-lane base, 8 outputs further for the upper lanes.  Measured on one box
        // (A/B builds): n = 512 34.0 -> 37.5 % of the roofline, N = 480 30.5 -> 33.5 %, n = 1024 unchanged (38.7 %).
        double2 lo8[8], hi8[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            unsigned a[4] = {(unsigned)__double2loint(v[BR16(r)].x), (unsigned)__double2hiint(v[BR16(r)].x),
                             (unsigned)__double2loint(v[BR16(r)].y), (unsigned)__double2hiint(v[BR16(r)].y)};
            unsigned c[4] = {(unsigned)__double2loint(v[BR16(r + 8)].x), (unsigned)__double2hiint(v[BR16(r + 8)].x),
                             (unsigned)__double2loint(v[BR16(r + 8)].y), (unsigned)__double2hiint(v[BR16(r + 8)].y)};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (S == 64) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(a[d], c[d], false, false);
                    a[d] = sw[0];
                    c[d] = sw[1];
                } else {
                    const auto sw = __builtin_amdgcn_permlane16_swap(a[d], c[d], false, false);
                    a[d] = sw[0];
                    c[d] = sw[1];
                }
            }
            lo8[r] = make_double2(__hiloint2double((int)a[1], (int)a[0]), __hiloint2double((int)a[3], (int)a[2]));
            hi8[r] = make_double2(__hiloint2double((int)c[1], (int)c[0]), __hiloint2double((int)c[3], (int)c[2]));
        }
        const int wbh = wb + (lower ? 0 : 8 * NS + ((8 * NS) >> 4)); // (8 NS is a multiple of 16: no carry into the pad)
        sync();
#pragma unroll
        for (int r = 0; r < 8; r++)
            b[wbh + r * NS + ((r * NS) >> 4)] = lo8[r];
        sync();
#pragma unroll
        for (int i = 0; i < 8; i++)
            w[i] = b[rbase + i * padk(S)];
        sync();
#pragma unroll
        for (int r = 0; r < 8; r++)
            b[wbh + r * NS + ((r * NS) >> 4)] = hi8[r];
        sync();
#pragma unroll
        for (int i = 0; i < 8; i++)
            w[8 + i] = b[rbase + i * padk(S)];
    } else {
    sync();
    if (lower) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            b[wb + (NS >= 16 ? r * padk(NS) : r * NS + ((r * NS) >> 4))] = v[BR16(r)];
    }
    sync();
#pragma unroll
    for (int i = 0; i < 8; i++)
        w[i] = b[rbase + i * padk(S)];
    sync();
    if (!lower) {
#pragma unroll
        for (int r = 0; r < 16; r++)
            b[wb + (NS >= 16 ? r * padk(NS) : r * NS + ((r * NS) >> 4))] = v[BR16(r)];
    }
    sync();
#pragma unroll
    for (int i = 0; i < 8; i++)
        w[8 + i] = b[rbase + i * padk(S)];
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
        v[i] = w[i];
    gpass_pre(v, ga, [&](int s) __attribute__((always_inline)) { return ldg2u(scalar_ptr_at(tab, s * NS2), m3); });
}

// forward transform of the pair's n points: v[i] = x[j + i S] -> X[j + r S] at v[BR16(r)].  b: the pair's half buffer.
// gs: the factors of the passes behind the second one, lane-ordered (FusedParams::gsmall): [8][16 R1], then (n = 8192)
// [8][256 R1] -- coalesced 16-byte loads; out of the generic W_65536 table the same factors are up to 64 different cache
// lines per wave instruction.
template <int LOGN>
__device__ __forceinline__ void forward(double2 (&v)[16], double2 *b, const double2 *g2l, const double2 *__restrict__ gs,
                                        const int j_)
{
    constexpr int n = 1 << LOGN, S = n / 16, NP = (LOGN + 3) / 4, R1 = n >> (4 * (NP - 1)), Q1 = 16 / R1, HQ = Q1 / 2;
    int j = j_;
    asm volatile("" : "+v"(j)); // addresses are derived here, per call (not hoisted out of the pair loop)
    j &= S - 1;                 // (range for the compiler: 32-bit table offsets)
    // pairs inside one wave need no hardware barrier (LDS operations of a wave execute in order), but the COMPILER must
    // not move a lane's reads above its writes: other lanes' data arrives through them
    const auto sync = [&]() __attribute__((always_inline)) {
        if (S > 64) {
            lds_barrier();
        } else {
            fence();
            asm volatile("" ::: "memory");
            fence();
        }
    };
    const int rbase = j + (j >> 4);              // padpos(j + i S) = rbase + i padk(S)
    const int w1 = j * R1 + ((j * R1) >> 4);     // padpos((j + m S) R1 + r) = w1 + r + m padk(S R1)
    r_dft<R1>(v);                                // pass 1: output (m, r) at v[m + r Q1]
    {
        double2 w[16], ga[4];
        const int m2 = j & (R1 - 1);
#pragma unroll
        for (int s = 0; s < 4; s++) // pass 2's factors: 8 x R1 values, from the workgroup's LDS copy (a broadcast read)
            ga[s] = g2l[s * R1 + m2];
        fence();
        // ---- transpose A, two half rounds: the lower / upper half of the positions
        sync(); // (previous readers of the buffer are done)
#pragma unroll
        for (int m = 0; m < HQ; m++)
#pragma unroll
            for (int r = 0; r < R1; r++)
                b[w1 + r + m * padk(S * R1)] = v[m + r * Q1];
        sync();
#pragma unroll
        for (int i = 0; i < 8; i++)
            w[i] = b[rbase + i * padk(S)];
        sync();
#pragma unroll
        for (int m = 0; m < HQ; m++)
#pragma unroll
            for (int r = 0; r < R1; r++)
                b[w1 + r + m * padk(S * R1)] = v[HQ + m + r * Q1];
        sync();
#pragma unroll
        for (int i = 0; i < 8; i++)
            w[8 + i] = b[rbase + i * padk(S)];
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = w[i];
        gpass_pre(v, ga, [&](int s) __attribute__((always_inline)) { return g2l[s * R1 + m2]; }); // pass 2: Ns = R1
    }
    level<S, R1>(v, b, gs, j, rbase, sync); // pass 3: Ns = 16 R1
    if constexpr (NP >= 4)
        level<S, 16 * R1>(v, b, gs + 8 * 16 * R1, j, rbase, sync); // pass 4 (n = 8192): Ns = 256 R1 = S
}

// Which column j (elements j + i S) a lane works on.  Any bijection inside a wave is correct -- j is only ever an index --
// and the choice decides the LDS bank conflicts of the transposes: a ds_read_b128 is served in four groups of sixteen lanes,
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS), i.e. by the parity of lane bits
// 4, 3, 2, a ds_write_b128 in groups of eight consecutive lanes over eight 16-byte slots.  With j = lane the sixteen lanes of a
// read group straddle the pad slot of the 17/16 layout (two cycles per group instead of one) and the stride-R1 writes of
// transpose A hit every second slot twice.  A GF(2)-linear relabelling makes each read group read ONE aligned block of
// sixteen columns (column bit 4 = that parity) and spreads the eight lanes of a write group over the eight slots
// (column bit 3 ^= lane bit 1 for R1 = 4, lane bit 0 for R1 = 2 / 8): reads 8 -> 4 LDS cycles, transpose-A writes
// 16 -> 8; n = 8192: 26.0 % -> 27.9 % of the HBM roofline, n = 1024 / 2048: + 1 point.  Global addresses are
// permuted inside aligned 32-lane groups only: every wave instruction touches the same cache lines as before.
template <int LOGN>
__device__ __forceinline__ int column_of_lane(const int l)
{
    const int rg = ((l >> 4) ^ (l >> 3) ^ (l >> 2)) & 1;
    if (LOGN == 9) // n = 512: the register trade of transpose B (level()) pairs lane l with lane l + 16 as columns c and c + 16, so
                   // column bit 4 must stay lane bit 4 and the read groups keep their pad-slot conflict;
                   // column bit 3 ^= lane bit 2 still spreads the stride-2 writes of transpose A over all eight slots
                   // (tools/lds_bank_sim.py: 16 -> 8 LDS cycles per store) and leaves transpose B's stores conflict-free
        return (l & ~8) | ((((l >> 3) ^ (l >> 2)) & 1) << 3);
    const int b3 = ((l >> 3) ^ ((LOGN == 10 || LOGN == 14) ? (l >> 1) : l)) & 1;
    return (l & ~0x18) | (b3 << 3) | (rg << 4);
}

} // namespace small

// PADDED: N < n (leading zero pad); N == n needs no per-sample validity masks
// Workgroup: 256 threads (n <= 1024: pairs never leave a wave, the workgroup is only a scheduling unit) or the 128 threads
// of ONE pair (n = 2048: the barriers of the half rounds then couple the pair's two waves and nobody else).
// MULTI: R references against the group in one pass (muse_batch_score_many): rows are loaded, reduced and transformed
// once; the pair's spectrum is parked lane-ordered in the workgroup's slice of p.zscratch (L2 / MALL resident, every
// thread re-reads only what it wrote) and every reference takes product, second transform and argmax from there.
// F32: float32-storage group (muse_group_create_f32, opt-in): the rows are float32 in HBM, widened exactly as they are
// consumed; the arithmetic is the float64 arithmetic of the float64 groups.
// ZREG (MULTI, n <= 8192): the pair's spectrum stays in REGISTERS over the references (256 VGPRs: half the resident waves)
// instead of being parked in the workgroup's slice of global scratch (n = 16384: 1024 threads per pair cap a lane at 128).
template <int LOGN, bool PADDED, bool MULTI, bool F32 = false>
__global__ __launch_bounds__((LOGN >= 11 ? (1 << LOGN) / 16 : 256), ((MULTI && LOGN <= 13) ? 2 : 4)) void xcorr_fused_small(const FusedParams p)
{
    using namespace occ4;
    using namespace fold;
    using namespace small;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;   // threads per pair: 32, 64, 128
    constexpr int TPB = LOGN >= 11 ? S : 256;
    constexpr int G = TPB / S;  // pairs per workgroup iteration: 8, 4, 1, 1, 1
    static_assert((LOGN >= 9 && LOGN <= 11) || LOGN == 13 || LOGN == 14, "n = 512, 1024, 2048, 8192, 16384");
    __shared__ double red[112]; // multi-wave pair reductions (n >= 2048): sums [4][16], maxima [2][16], indices [2][16] ints
    // pass 2's eight factors per phase m2 / R1: 8 x R1 distinct values for the whole workgroup.  Gathered per lane from the
    // W_65536 table they were the kernel's longest stall (scattered L2 lines in front of the first butterfly of a pass:
    // tools/ablate/small_exp.sh, + 15-20 % with them out of the way); from LDS they are one broadcast read each.
    constexpr int NP_ = (LOGN + 3) / 4, R1_ = n >> (4 * (NP_ - 1));
    __shared__ double2 g2l[8 * R1_];
    __shared__ double2 xbuf[(TPB / 64) * 544]; // 8.7 KB per wave: half-round buffers of the pairs (8 S 17/16 double2 per pair)
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // (S >= 64: a wave works on one pair -- the pair slot is wave-uniform and everything derived from it stays scalar)
    const int g = S >= 64 ? __builtin_amdgcn_readfirstlane(t / S) : t / S;
    const int j = column_of_lane<LOGN>(t % S);
    double2 *const b = xbuf + g * (8 * S + S / 2);
    const int N = PADDED ? p.N : n, pad = PADDED ? n - N : 0;
    const double invN = PADDED ? p.invN : 1.0 / (double)n, invNm1 = PADDED ? p.invNm1 : 1.0 / (double)(n - 1); // (the launcher's quotients: scalar registers)
    const double2 *__restrict__ twm = p.twm;
    const double2 *__restrict__ gs = p.gsmall;
    if (t < 8 * R1_)
        g2l[t] = tw_factor<R1_>(twm, t % R1_, t / R1_);
    __syncthreads();
    // optional indirection (filter-and-refine Run): process pair_list[0 .. *pair_count) instead of every pair
    const long long total = p.pair_list ? (long long)*p.pair_count : p.npairs;
    const long long ngroups = (total + G - 1) / G;

    // The rows of the NEXT iteration are requested behind the per-lane argmax of the current one (the transform registers
    // are free then) and consumed at the top of the loop: element j + i S of the padded series is sample j + i S - pad; a pad
    // position reads up to `pad` samples IN FRONT of the row (the end of the previous row, or the guard the group
    // allocation keeps in front of row 0: capi_group.hip, GROUP_GUARD) and is masked -- no clamp, so every load is one
    // base plus a compile-time offset.
    double xa[16], xb[16], KA, KB;
    const auto request = [&](long long it2) __attribute__((always_inline)) {
        if (it2 >= ngroups)
            it2 = ngroups - 1; // (nothing left: an L2-hot dummy)
        const long long slot = it2 * G + g;
        const long long sl = slot < total ? slot : total - 1;
        const long long pair = p.pair_list ? p.pair_list[sl] : sl;
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        int jr = j;
        asm volatile("" : "+v"(jr)); // (offsets derived per request, not hoisted)
        jr &= S - 1;                 // (the range the compiler no longer sees: keeps global offsets 32-bit, saddr + voffset loads)
        if (F32) { // the same requests on float32 rows (half the bytes per request)
            const float *ra = p.rows32 + rA * p.stride, *rb = p.rows32 + (hasB ? rA + 1 : rA) * p.stride;
            const auto all_pad = [&](int i) __attribute__((always_inline)) { return PADDED && i < 8 && (i + 1) * S <= pad; };
            if (S >= 64) {
                KA = (double)scalar_ptr(ra)[0];
                KB = (double)scalar_ptr(rb)[0];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const long long off = all_pad(i) ? 0ll : (long long)i * S - pad;
                    xa[i] = (double)__builtin_nontemporal_load(scalar_ptr_at(ra, off) + (unsigned)jr);
                    xb[i] = (double)__builtin_nontemporal_load(scalar_ptr_at(rb, off) + (unsigned)jr);
                }
            } else {
                KA = (double)ra[0];
                KB = (double)rb[0];
                const float *la = ra - pad + jr, *lb = rb - pad + jr;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    xa[i] = (double)__builtin_nontemporal_load(all_pad(i) ? ra + jr : la + i * S);
                    xb[i] = (double)__builtin_nontemporal_load(all_pad(i) ? rb + jr : lb + i * S);
                }
            }
            return;
        }
        const double *ra = p.rows + rA * p.stride, *rb = p.rows + (hasB ? rA + 1 : rA) * p.stride;
        // PADDED: the elements i S .. (i + 1) S - 1 of the padded series are all pad when (i + 1) S <= pad (a wave-uniform
        // test; pad < n / 2, so only i < 8 can be).  Such a request would fetch the end of the previous row from HBM only to be
        // masked (N = 5000 -> n = 8192: six of the sixteen requests, N = 480 -> 512: one): it is pointed at the row's own
        // first S samples instead -- the same unconditional load instruction (a branch around it parks the row registers in
        // scratch), an L2 hit instead of HBM bytes.
        const auto all_pad = [&](int i) __attribute__((always_inline)) { return PADDED && i < 8 && (i + 1) * S <= pad; };
        if (S >= 64 && !PADDED) { // the wave works on one pair: scalar bases + the shared VGPR offset 8 j; ONE base per row serves
                                  // every request whose immediate offset (-4096 ... 4095 bytes) reaches it
            KA = scalar_ptr(ra)[0];
            KB = scalar_ptr(rb)[0];
            constexpr int PER = S <= 64 ? 8 : S <= 128 ? 4 : S <= 512 ? 2 : 1;
#pragma unroll
            for (int gq = 0; gq < 16 / PER; gq++) {
                constexpr int HALF = PER / 2;
                const int c = (gq * PER + HALF) * S;
                const gptr<double> ba = scalar_ptr_at(ra, c), bb = scalar_ptr_at(rb, c);
#pragma unroll
                for (int k = 0; k < PER; k++) {
                    const int i = gq * PER + k;
                    xa[i] = __builtin_nontemporal_load(ba + (i * S - c) + (unsigned)jr);
                    xb[i] = __builtin_nontemporal_load(bb + (i * S - c) + (unsigned)jr);
                }
            }
        } else if (S >= 64) {
            KA = scalar_ptr(ra)[0];
            KB = scalar_ptr(rb)[0];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const long long off = all_pad(i) ? 0ll : (long long)i * S - pad;
                xa[i] = __builtin_nontemporal_load(scalar_ptr_at(ra, off) + (unsigned)jr);
                xb[i] = __builtin_nontemporal_load(scalar_ptr_at(rb, off) + (unsigned)jr);
            }
        } else { // two pairs per wave: one 64-bit base per lane and row, immediate offsets 256 i bytes
            KA = ra[0];
            KB = rb[0];
            const double *la = ra - pad + jr, *lb = rb - pad + jr;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                xa[i] = __builtin_nontemporal_load(all_pad(i) ? ra + jr : la + i * S);
                xb[i] = __builtin_nontemporal_load(all_pad(i) ? rb + jr : lb + i * S);
            }
        }
    };
    // n <= 1024 (a wave holds one or two whole pairs, no workgroup barrier anywhere): the rows are requested where they are
    // consumed -- the sixteen waves of a CU hide the latency, and the 64 registers a prefetch would hold across the argmax
    // and the write-out are worth more (n = 512: 0.543 -> 0.453 ms per 400 000 series, 37.9 -> 45.4 % of the roofline;
    // n = 1024: +2 %; from n = 2048 the prefetch wins, at n = 16384 by 7 %).
    constexpr bool PREFETCH = MULTI || LOGN >= 11;
    if (PREFETCH && blockIdx.x < ngroups)
        request(blockIdx.x);
    for (long long it = blockIdx.x; it < ngroups; it += gridDim.x) {
        if (!PREFETCH)
            request(it);
        const long long slot = it * G + g;
        const bool live = slot < total;
        const long long sl = live ? slot : total - 1; // idle sub-groups shadow the last pair
        const long long pair = p.pair_list ? p.pair_list[sl] : sl;
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        // ---- d = x - K with K the first sample, shifted statistics
        double2 v[16];
        int js = j;
        asm volatile("" : "+v"(js)); // (per-sample validity derived per iteration, not hoisted: 16 masks)
        js &= S - 1;
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool valid = !PADDED || i >= 8 || js + i * S - pad >= 0; // (pad < n / 2: the upper half is always data)
            const double da = valid ? xa[i] - KA : 0.0, db = valid ? xb[i] - KB : 0.0;
            v[i] = make_double2(da, db);
            q0 += da;
            q1 = fma(da, da, q1);
            q2 += db;
            q3 = fma(db, db, q3);
        }
        pair_sum4<S>(q0, q1, q2, q3, red, wave);
        const Stat stA{q0, q1}, stB{q2, q3};
        bool zeroA, nanA, zeroB, nanB;
        const double varA0 = variance(stA, invN, invNm1, zeroA, nanA);
        const double varB0 = variance(stB, invN, invNm1, zeroB, nanB);
        const bool deadA = zeroA || nanA, deadB = zeroB || nanB || !hasB;
        // both series go into the shared transform at O(1): exact power-of-two scales close to 1/sigma
        // (fft_device.h, pow2_inv_sigma), folded into the mean removal; the variances scale along exactly
        const double sA = deadA ? 1.0 : pow2_inv_sigma(varA0), sB = deadB ? 1.0 : pow2_inv_sigma(varB0);
        // (a pair that fills its wave(s): the scaled variances wait for the result write-out in SGPRs, not in four registers the
        // allocator parks in scratch across the transforms)
        const double varA = S >= 64 ? uniform(varA0 * sA * sA) : varA0 * sA * sA, varB = S >= 64 ? uniform(varB0 * sB * sB) : varB0 * sB * sB;
        const double mA = q0 * invN * sA, mB = q2 * invN * sB;
        asm volatile("" : "+v"(js));
        js &= S - 1;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool valid = !PADDED || i >= 8 || js + i * S - pad >= 0; // (pad < n / 2: the upper half is always data)
            v[i].x = valid ? fma(v[i].x, sA, -mA) : 0.0;
            v[i].y = valid ? fma(v[i].y, sB, -mB) : 0.0;
        }
        if (deadA || deadB) { // uniform over the pair, rare: a sigma == 0 / NaN series (or the missing partner of an odd
                              // last row) must contribute exact zeros to the shared complex transform
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v[i].x = deadA ? 0.0 : v[i].x;
                v[i].y = deadB ? 0.0 : v[i].y;
            }
        }
        // ---- Z = FFT(yA + i yB);  V = Z conj(X)/n;  ccA + i ccB = FFT(V)
        forward<LOGN>(v, b, g2l, gs, j);
        // the workgroup's slice of the spectrum scratch: element i of thread gt at [i][gt] (scalar base + 32-bit lane offset)
        const long long ZT = (long long)gridDim.x * TPB;
        const auto zslot = [&](int i) __attribute__((always_inline)) {
            int gt = t;
            asm volatile("" : "+v"(gt)); // (derived per use, not hoisted)
            gt &= TPB - 1;
            return (d2v __attribute__((address_space(1))) *)scalar_ptr_at(p.zscratch, i * ZT + (long long)blockIdx.x * TPB) + (unsigned)gt;
        };
        constexpr bool ZREG = MULTI && LOGN <= 13;
        double2 Z[16];
        if (MULTI) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (ZREG)
                    Z[i] = v[i];
                else
                    *zslot(i) = d2v{v[i].x, v[i].y};
            }
        }
        const int R = MULTI ? p.R : 1;
#pragma clang loop unroll(disable)
        for (int ref = 0; ref < R; ref++) {
        const double2 *__restrict__ xcr = MULTI ? uniform_ptr(p.xcp_many[ref]) : p.xc;
        if (MULTI) {
            fence();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (ZREG) {
                    v[i] = Z[i];
                } else {
                    const d2v z = *zslot(i);
                    v[i] = make_double2(z.x, z.y);
                }
            }
        }
        { // V = Z conj(X)/n in place, the factors in four batches of four (two in flight: 32 registers), then the
          // registers renamed to natural order (X[j + r S] sits at v[BR16(r)])
            int jx = j;
            asm volatile("" : "+v"(jx)); // (the table offsets are derived here, not hoisted out of the pair loop)
            jx &= S - 1;
            // one scalar base per PERX table rows (16 S bytes apart; immediate offsets -4096 ... 4095 bytes), formed once
            constexpr int PERX = S <= 32 ? 16 : S <= 64 ? 8 : S <= 128 ? 4 : S <= 256 ? 2 : 1;
            gptr<double2> xbase[16 / PERX];
#pragma unroll
            for (int gq = 0; gq < 16 / PERX; gq++)
                xbase[gq] = scalar_ptr_at(xcr, (gq * PERX + PERX / 2) * S);
            const auto xcl = [&](int r) __attribute__((always_inline)) {
                return ldg2u(xbase[r / PERX] + (r * S - ((r / PERX) * PERX + PERX / 2) * S), (unsigned)jx);
            };
            double2 xq[2][4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                xq[0][k] = xcl(k);
#pragma unroll
            for (int bt = 0; bt < 4; bt++) {
                fence();
                if (bt < 3) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        xq[(bt + 1) & 1][k] = xcl(4 * (bt + 1) + k);
                }
                fence();
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = 4 * bt + k;
                    v[BR16(r)] = cmul(v[BR16(r)], xq[bt & 1][k]);
                }
            }
            double2 w[16];
#pragma unroll
            for (int r = 0; r < 16; r++)
                w[r] = v[BR16(r)];
#pragma unroll
            for (int r = 0; r < 16; r++)
                v[r] = w[r];
        }
        forward<LOGN>(v, b, g2l, gs, j); // cc[j + r S] at v[BR16(r)]
        // ---- maxAbsIndex (xcorr.go:39-50) per series: ascending r = ascending index for this thread
        double sa = 0.0, sb = 0.0; // signed value of the lane's first maximum of |cc|, and its register index
        int ra_ = 0, rb_ = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const double xa = v[BR16(r)].x, xb = v[BR16(r)].y;
            const bool ga = fabs(xa) > fabs(sa), gb = fabs(xb) > fabs(sb);
            sa = ga ? xa : sa;
            ra_ = ga ? r : ra_;
            sb = gb ? xb : sb;
            rb_ = gb ? r : rb_;
        }
        const double ma = fabs(sa), mb = fabs(sb);
        const int ia = j + ra_ * S, ib = j + rb_ * S;
        const double cc0a = v[0].x, cc0b = v[0].y; // (lane 0: cc[0], reported when nothing is above 0)
        fence();
        if (!MULTI && PREFETCH)
            request(it + gridDim.x); // the next iteration's rows: in flight during the reductions and the result write-out
        fence();
        double pa = ma, pb = mb;
        pair_max2<S>(pa, pb, red, wave);
        int ca = (ma == pa && pa > 0.0) ? ia : 0x7fffffff, cb = (mb == pb && pb > 0.0) ? ib : 0x7fffffff;
        pair_min_i2<S>(ca, cb, red, wave);
        // the lane that owns the winning index writes the result (nothing above 0: lane 0 reports cc[0] at index 0)
        if (live) {
            const bool ownA = ca == 0x7fffffff ? j == 0 : (ia == ca && ma == pa);
            if (ownA) {
                double y = __builtin_amdgcn_rsq(varA);
                y = y * fma(-0.5 * varA * y, y, 1.5);
                y = y * fma(-0.5 * varA * y, y, 1.5);
                double mv = (ca == 0x7fffffff ? cc0a : sa) * y;
                const int idx = ca == 0x7fffffff ? 0 : ca;
                int lag = idx > n / 2 ? idx - n : idx;
                if (zeroA) { mv = 0.0; lag = 0; }               // xcorr.go:166-167
                if (nanA) { mv = __builtin_nan(""); lag = 0; }
                (MULTI ? p.mv_many[ref] : p.mv)[rA] = mv;
                (MULTI ? p.lag_many[ref] : p.lag)[rA] = lag;
            }
            const bool ownB = cb == 0x7fffffff ? j == 0 : (ib == cb && mb == pb);
            if (ownB && hasB) {
                double y = __builtin_amdgcn_rsq(varB);
                y = y * fma(-0.5 * varB * y, y, 1.5);
                y = y * fma(-0.5 * varB * y, y, 1.5);
                double mv = (cb == 0x7fffffff ? cc0b : sb) * y;
                const int idx = cb == 0x7fffffff ? 0 : cb;
                int lag = idx > n / 2 ? idx - n : idx;
                if (zeroB) { mv = 0.0; lag = 0; }
                if (nanB) { mv = __builtin_nan(""); lag = 0; }
                (MULTI ? p.mv_many[ref] : p.mv)[rA + 1] = mv;
                (MULTI ? p.lag_many[ref] : p.lag)[rA + 1] = lag;
            }
        }
        } // (references)
        if (MULTI) { // (requested inside the loop the 64 row registers would be live across all references)
            fence();
            request(it + gridDim.x);
        }
    }
}

// The batched two-sided xCorr (xcorr.go:102-153) on the same transforms: pair i = (x_i, y_i), each zero-padded
// in front on its own (any Nx, Ny <= n).  With xr[j] = x[-j mod n] (the row read backwards: an address pattern) and
// z = xr + i y:  conj(X) = FFT(xr), Z = FFT(z) = Xr + i Y, Z^2 = (Xr^2 - Y^2) + 2 i Xr Y, so
//     cc = FFT(conj(X) Y) / n = Im FFT(Z^2) / (2 n)
// -- two forward transforms per pair as above, the spectrum product a square of what the thread already holds: no
// table, no mirrored element Z[-f] (round 3's first version fetched it through the Stockham engine's natural-order LDS image).
// Statistics first (xcorr.go:108-128: either sigma == 0 -> nil), both series centred and scaled to O(1) by exact powers of
// two (two_device.h).  Rows are requested where they are used: the kernel is bound by its arithmetic.
template <int LOGN, bool PADDED>
__global__ __launch_bounds__((LOGN >= 11 ? (1 << LOGN) / 16 : 256), 4) void xcorr_two_sided_small(const FusedParams p, const two::PairInv iv)
{
    using namespace occ4;
    using namespace fold;
    using namespace small;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;
    constexpr int TPB = LOGN >= 11 ? S : 256;
    constexpr int G = TPB / S;
    static_assert((LOGN >= 9 && LOGN <= 11) || LOGN == 13 || LOGN == 14, "n = 512, 1024, 2048, 8192, 16384");
    __shared__ double red[112];
    constexpr int NP_ = (LOGN + 3) / 4, R1_ = n >> (4 * (NP_ - 1));
    __shared__ double2 g2l[8 * R1_];
    __shared__ double2 xbuf[(TPB / 64) * 544];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = S >= 64 ? __builtin_amdgcn_readfirstlane(t / S) : t / S;
    const int j = column_of_lane<LOGN>(t % S);
    double2 *const b = xbuf + g * (8 * S + S / 2);
    const int padx = PADDED ? n - p.Nx : 0, pady = PADDED ? n - p.N : 0;
    const bool normalize = p.normalize_y != 0;
    const double2 *__restrict__ twm = p.twm;
    const double2 *__restrict__ gs = p.gsmall;
    if (t < 8 * R1_)
        g2l[t] = tw_factor<R1_>(twm, t % R1_, t / R1_);
    __syncthreads();
    const long long total = p.npairs;
    const long long ngroups = (total + G - 1) / G;
    for (long long it = blockIdx.x; it < ngroups; it += gridDim.x) {
        const long long slot = it * G + g;
        const bool live = slot < total;
        const long long pair = live ? slot : total - 1; // idle sub-groups shadow the last pair
        const double *const rx = p.xrows + pair * p.xstride, *const ry = p.rows + pair * p.stride;
        double2 v[16];
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        {
            const double KA = normalize ? rx[0] : 0.0, KB = normalize ? ry[0] : 0.0;
            // position e = j + i S of the padded arrays holds y[e - pady] and x[(-e mod n) - padx]; four batches of four
#pragma unroll
            for (int h = 0; h < 4; h++) {
                double xa[4], yb[4];
                int jb = j;
                asm volatile("" : "+v"(jb)); // (a batch's offsets and masks are formed in the batch)
                jb &= S - 1;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e = jb + (4 * h + k) * S;
                    const int ex = ((n - e) & (n - 1)) - padx, ey = e - pady;
                    if (S >= 64) { // the wave works on one pair: scalar bases + 32-bit lane offsets
                        xa[k] = __builtin_nontemporal_load(scalar_ptr(rx) + (unsigned)(PADDED && ex < 0 ? 0 : ex));
                        yb[k] = __builtin_nontemporal_load(scalar_ptr(ry) + (unsigned)(PADDED && ey < 0 ? 0 : ey));
                    } else {
                        xa[k] = __builtin_nontemporal_load(rx + (PADDED && ex < 0 ? 0 : ex));
                        yb[k] = __builtin_nontemporal_load(ry + (PADDED && ey < 0 ? 0 : ey));
                    }
                }
                fence();
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e = jb + (4 * h + k) * S;
                    double da = xa[k] - KA, db = yb[k] - KB;
                    if (PADDED) {
                        da = ((n - e) & (n - 1)) - padx >= 0 ? da : 0.0;
                        db = e - pady >= 0 ? db : 0.0;
                    }
                    v[4 * h + k] = make_double2(da, db);
                    q[0] += da;
                    q[1] = fma(da, da, q[1]);
                    q[2] += db;
                    q[3] = fma(db, db, q[3]);
                }
            }
        }
        pair_sum4<S>(q[0], q[1], q[2], q[3], red, wave);
        const two::PairScale ps = two::pair_scale(q, iv, normalize);
        const bool dead = ps.nil || ps.nan;
        const double fac = ps.fac * (1.0 / (2.0 * n)); // (pair_scale's factor assumes a spectrum already divided by n; 1 / 2n is exact)
        {
            const double sA = dead ? 0.0 : ps.sA, sB = dead ? 0.0 : ps.sB, mA = dead ? 0.0 : ps.mA, mB = dead ? 0.0 : ps.mB;
            int jb = j;
            asm volatile("" : "+v"(jb)); // (the validity masks are recomputed, not kept across the statistics)
            jb &= S - 1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = jb + i * S;
                const bool vx = !PADDED || ((n - e) & (n - 1)) - padx >= 0, vy = !PADDED || e - pady >= 0;
                v[i].x = vx ? fma(v[i].x, sA, -mA) : 0.0;
                v[i].y = vy ? fma(v[i].y, sB, -mB) : 0.0;
            }
        }
        forward<LOGN>(v, b, g2l, gs, j); // Z[j + r S] at v[BR16(r)]
        {
            double2 w[16];
#pragma unroll
            for (int r = 0; r < 16; r++) { // the square, renamed to natural order
                const double2 z = v[BR16(r)];
                w[r] = make_double2(fma(z.x, z.x, -(z.y * z.y)), (z.x + z.x) * z.y);
            }
#pragma unroll
            for (int r = 0; r < 16; r++)
                v[r] = w[r];
        }
        forward<LOGN>(v, b, g2l, gs, j); // 2 n cc[j + r S] at v[BR16(r)].y
        if (p.cc_out && live && !dead) {
            double *const cc = p.cc_out + pair * (long long)n;
#pragma unroll
            for (int r = 0; r < 16; r++)
                cc[j + r * S] = v[BR16(r)].y * fac;
        }
        // ---- maxAbsIndex (xcorr.go:39-50): ascending r = ascending index for this thread
        double sa = 0.0;
        int ra_ = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const double x = v[BR16(r)].y;
            const bool ga = fabs(x) > fabs(sa);
            sa = ga ? x : sa;
            ra_ = ga ? r : ra_;
        }
        const double ma = fabs(sa);
        const int ia = j + ra_ * S;
        const double cc0 = v[0].y; // (lane with j == 0: cc[0], reported when nothing is above 0)
        const double pa = pair_max<S>(ma, red, wave);
        const int ca = pair_min_i<S>((ma == pa && pa > 0.0) ? ia : 0x7fffffff, red + 96, wave);
        if (live) {
            const bool own = ca == 0x7fffffff ? j == 0 : (ia == ca && ma == pa);
            if (own) {
                const int idx = ca == 0x7fffffff ? 0 : ca;
                double mv = (ca == 0x7fffffff ? cc0 : sa) * fac;
                int lag = idx > n / 2 ? idx - n : idx;
                if (ps.nil) { mv = 0.0; lag = 0; }               // xcorr.go:110-127
                if (ps.nan) { mv = __builtin_nan(""); lag = 0; } // every cc is NaN: maxAbsIndex keeps index 0
                p.mv[pair] = mv;
                p.lag[pair] = lag;
                if (p.nil_out)
                    p.nil_out[pair] = ps.nil ? 1 : 0;
            }
        }
        if (S > 64)
            lds_barrier(); // (red is reused by the next pair's statistics)
    }
}

template <int LOGN>
static hipError_t launch_two_small_n(const FusedParams &p, int num_cus, hipStream_t stream)
{
    constexpr int TPB = LOGN >= 11 ? (1 << LOGN) / 16 : 256;
    constexpr int G = TPB / ((1 << LOGN) / 16);
    const long long ngroups = (p.npairs + G - 1) / G;
    const long long grid = std::min<long long>(ngroups, (long long)num_cus * (1024 / TPB) * 8);
    const two::PairInv iv = two::pair_inv(p.Nx, p.N, 1 << LOGN);
    if (p.Nx < (1 << LOGN) || p.N < (1 << LOGN))
        hipLaunchKernelGGL((xcorr_two_sided_small<LOGN, true>), dim3((unsigned)grid), dim3(TPB), 0, stream, p, iv);
    else
        hipLaunchKernelGGL((xcorr_two_sided_small<LOGN, false>), dim3((unsigned)grid), dim3(TPB), 0, stream, p, iv);
    return hipGetLastError();
}
// two-sided xCorr, n = 512, 1024, 2048, 8192, 16384 (launch_two_sided's argument checks apply)
hipError_t launch_two_sided_small(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (!p.xrows || !p.rows || !p.twm || !p.gsmall || !p.mv || !p.lag)
        return hipErrorInvalidValue;
    switch (p.logn) {
    case 9: return launch_two_small_n<9>(p, num_cus, stream);
    case 10: return launch_two_small_n<10>(p, num_cus, stream);
    case 11: return launch_two_small_n<11>(p, num_cus, stream);
    case 13: return launch_two_small_n<13>(p, num_cus, stream);
    case 14: return launch_two_small_n<14>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int LOGN>
static hipError_t launch_small_n(const FusedParams &p, int num_cus, hipStream_t stream)
{
    constexpr int TPB = LOGN >= 11 ? (1 << LOGN) / 16 : 256;
    constexpr int G = TPB / ((1 << LOGN) / 16);
    const long long ngroups = (p.npairs + G - 1) / G;
    if (p.R > 1) { // one pass for R references: exactly the resident workgroups (n = 16384: each with its slice of the spectrum scratch)
        constexpr bool ZREG = LOGN <= 13;
        if (!p.xcp_many || !p.mv_many || !p.lag_many || (!ZREG && !p.zscratch))
            return hipErrorInvalidValue;
        const long long grid = std::min<long long>(ngroups, (long long)num_cus * (ZREG ? std::max(1, 512 / TPB) : 1024 / TPB));
        if (!ZREG && (size_t)grid * TPB * 16 > (size_t)p.zslots * 4096)
            return hipErrorInvalidValue;
        if (p.N < (1 << LOGN))
            hipLaunchKernelGGL((xcorr_fused_small<LOGN, true, true>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_fused_small<LOGN, false, true>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
        return hipGetLastError();
    }
    const long long grid = std::min<long long>(ngroups, (long long)num_cus * (1024 / TPB) * 8);
    if (p.rows32) {
        if (p.N < (1 << LOGN))
            hipLaunchKernelGGL((xcorr_fused_small<LOGN, true, false, true>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_fused_small<LOGN, false, false, true>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
    } else if (p.N < (1 << LOGN))
        hipLaunchKernelGGL((xcorr_fused_small<LOGN, true, false>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
    else
        hipLaunchKernelGGL((xcorr_fused_small<LOGN, false, false>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
    return hipGetLastError();
}

// n = 512, 1024, 2048, 8192, 16384 (float64 rows); any N in (n/2, n]
// p.rows must carry n - N < n / 2 readable elements in front of row 0 (zero-padded rows are read unclamped and masked;
// capi_group.hip allocates every group with GROUP_GUARD >= SMALL_MAX_N / 2 such elements)
hipError_t launch_fused_small(const FusedParams &p_in, int num_cus, hipStream_t stream)
{
    const FusedParams p = with_reciprocals(p_in);
    static_assert((1 << 14) <= SMALL_MAX_N, "the largest length built below");
    if ((!p.rows && !p.rows32) || !p.twm || (!p.xc && p.R <= 1) || !p.gsmall || (p.rows32 && p.R > 1))
        return hipErrorInvalidValue;
    switch (p.logn) {
    case 9: return launch_small_n<9>(p, num_cus, stream);
    case 10: return launch_small_n<10>(p, num_cus, stream);
    case 11: return launch_small_n<11>(p, num_cus, stream);
    case 13: return launch_small_n<13>(p, num_cus, stream);
    case 14: return launch_small_n<14>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace muse
