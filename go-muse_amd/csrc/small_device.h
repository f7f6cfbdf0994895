// small_device.h -- the transforms of xcorr_small.hip (n = 512, 1024, 2048, 8192, 16384: radix-16 Stockham passes with the twiddles folded
// into the butterflies, every transpose in two half rounds through a buffer of n / 2 points, lanes relabelled to columns so that
// the LDS read and write groups meet no bank conflicts) and the reductions over a pair's lanes, shared with xcorr_real.hip
// (n = 32768 as ONE real series on the 16384-point complex transform).  The design notes are in xcorr_small.hip's header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fold_device.h"
#include "r16_device.h"

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

// ---- The 16384-point transform as 16 x 1024 (xcorr_real.hip): sixteen waves, each with a 1024-point transform of its OWN
// (forward<10>: its two transposes stay inside the wave -- no workgroup barrier, the waves drift apart and one wave's LDS phase
// runs under another's arithmetic) and ONE transpose across the workgroup per transform, where forward<14> has three with four
// barriers each and every wave of the CU in lockstep (measured on the lockstep kernels: the vector unit busy 0.54 - 0.59 of the
// time, on the kernels whose transposes are wave-local 0.79 - 0.86: profiles/r05_counters.json).  M = 16384, thread j = 64 w + c
// (w = wave, c = the lane's column of the 1024-point transform), b = the workgroup's buffer (16 x 544 points: the global transpose
// uses 16 x 512 of them unpadded -- a wave's 64 lanes store / load 64 consecutive points --, the wave-local transforms 544 each).
//
// Decimation in frequency (the FIRST transform of a series: the rows arrive coalesced as z[j + 1024 i]):
//   Y[k1] = W_M^(j k1) sum_i W_16^(i k1) z[j + 1024 i]        (dft16_nr + fifteen products, wfetch(k1) = W_M^(j k1))
//   transpose: wave k1 collects Y[k1] of all 1024 threads;  Z[k1 + 16 k2] = FFT_1024 over j of Y[k1][j]
// out: Z[w + 16 (c + 64 r)] at v[BR16(r)].
template <typename TW>
__device__ __forceinline__ void forward_split_dif(double2 (&v)[16], double2 *b, const double2 *g2l, const double2 *__restrict__ gs10,
                                                  const int j_, const int wave, TW wfetch)
{
    int j = j_;
    asm volatile("" : "+v"(j)); // (addresses derived here, per call)
    j &= 1023;
    {
        double2 ta[5], tb[5];
#pragma unroll
        for (int k = 0; k < 5; k++)
            ta[k] = wfetch(1 + k);
        fence();
        dft16_nr(v); // Y[k1] at v[BR16(k1)], before its twiddle
        fence();
#pragma unroll
        for (int k = 0; k < 5; k++)
            tb[k] = wfetch(6 + k);
#pragma unroll
        for (int k = 0; k < 5; k++)
            v[BR16(1 + k)] = cmul(v[BR16(1 + k)], ta[k]);
        fence();
#pragma unroll
        for (int k = 0; k < 5; k++)
            ta[k] = wfetch(11 + k);
#pragma unroll
        for (int k = 0; k < 5; k++)
            v[BR16(6 + k)] = cmul(v[BR16(6 + k)], tb[k]);
        fence();
#pragma unroll
        for (int k = 0; k < 5; k++)
            v[BR16(11 + k)] = cmul(v[BR16(11 + k)], ta[k]);
    }
    // the transpose across the workgroup, two half rounds: threads j < 512 (waves 0 - 7) store all sixteen values, everybody
    // loads eight; then the other half.  Position k1 512 + (j mod 512).
    const int c = j & 63, wpos = j & 511, rpos = wave * 512 + c;
    double2 w[16];
    lds_barrier(); // (the buffer's previous users are done)
    if (wave < 8) {
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++)
            lds_st2(b + k1 * 512 + wpos, v[BR16(k1)]);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 8; i++)
        w[i] = lds_ld2(b + rpos + 64 * i);
    lds_barrier();
    if (wave >= 8) {
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++)
            lds_st2(b + k1 * 512 + wpos, v[BR16(k1)]);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 8; i++)
        w[8 + i] = lds_ld2(b + rpos + 64 * i);
#pragma unroll
    for (int i = 0; i < 16; i++)
        v[i] = w[i];
    lds_barrier(); // (every wave has its values: the wave-local buffers lie over the transpose's image)
    forward<10>(v, b + wave * 544, g2l, gs10, c);
}
// Decimation in time (the SECOND transform: its input sits where the first one's output does, v[i] = C[w + 16 (c + 64 i)]):
//   y[m1][k2] = FFT_1024 over b of C[m1 + 16 b]            (wave m1, forward<10>)
//   transpose: thread j = k2 collects y[m1][k2], m1 = 0 .. 15
//   X[1024 k1 + j] = sum_m1 W_16^(m1 k1) W_M^(m1 j) y[m1][j]   (the generalised pass that ends forward<14>: tab = its table, [8][1024])
// out: X[j + 1024 r] at v[BR16(r)] -- forward<14>'s order.
__device__ __forceinline__ void forward_split_dit(double2 (&v)[16], double2 *b, const double2 *g2l, const double2 *__restrict__ gs10,
                                                  const double2 *__restrict__ tab, const int j_, const int wave)
{
    int j = j_;
    asm volatile("" : "+v"(j));
    j &= 1023;
    const int c = j & 63, wpos = wave * 512 + c, rpos = j & 511;
    lds_barrier(); // (the buffer's previous users are done)
    forward<10>(v, b + wave * 544, g2l, gs10, c); // y[w][c + 64 r] at v[BR16(r)]
    double2 ga[4];
#pragma unroll
    for (int s = 0; s < 4; s++) // the last pass's first factors travel during the transpose
        ga[s] = ldg2u(scalar_ptr_at(tab, s * 1024), (unsigned)j);
    fence();
    // two half rounds: every wave stores its registers r < 8 (k2 < 512), the threads j < 512 load their sixteen; then r >= 8
    lds_barrier(); // (every wave is through with its own buffer)
#pragma unroll
    for (int r = 0; r < 8; r++)
        lds_st2(b + wpos + 64 * r, v[BR16(r)]);
    lds_barrier();
    if (wave < 8) {
        double2 w[16];
#pragma unroll
        for (int m = 0; m < 16; m++)
            w[m] = lds_ld2(b + m * 512 + rpos);
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 8; r++)
            lds_st2(b + wpos + 64 * r, v[BR16(8 + r)]);
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 16; m++)
            v[m] = w[m];
    } else {
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 8; r++)
            lds_st2(b + wpos + 64 * r, v[BR16(8 + r)]);
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 16; m++)
            v[m] = lds_ld2(b + m * 512 + rpos);
    }
    gpass_pre(v, ga, [&](int s) __attribute__((always_inline)) { return ldg2u(scalar_ptr_at(tab, s * 1024), (unsigned)j); });
}

// Which column j (elements j + i S) a lane works on.  Any bijection inside a wave is correct -- j is only ever an index --
// and the choice decides the LDS bank conflicts of the transposes: a ds_read_b128 is served in four groups of sixteen lanes,
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS), i.e. by the parity of lane bits
// 4, 3, 2, a ds_write_b128 in groups of eight consecutive lanes over eight 16-byte slots.  With j = lane the sixteen lanes of a
// read group straddle the pad slot of the 17/16 layout (two cycles per group instead of one) and the stride-R1 writes of
// transpose A hit every second slot twice.  A GF(2)-linear relabelling makes each read group read ONE aligned block of
// sixteen columns (column bit 4 = that parity) and spreads the eight lanes of a write group over the eight slots
// (column bit 3 ^= lane bit 1 for R1 = 4, lane bit 0 for R1 = 2 / 8): reads 8 -> 4 LDS cycles, transpose-A writes
// 16 -> 8 (docs/HISTORY.md section 4.2); n = 8192: 26.0 % -> 27.9 % of the HBM roofline, n = 1024 / 2048: + 1 point.  Global addresses are
// permuted inside aligned 32-lane groups only: every wave instruction touches the same cache lines as before.
template <int LOGN>
__device__ __forceinline__ int column_of_lane(const int l)
{
    const int rg = ((l >> 4) ^ (l >> 3) ^ (l >> 2)) & 1;
    if (LOGN == 9) // n = 512: the register trade of transpose B (level()) pairs lane l with lane l + 16 as columns c and c + 16, so
                   // column bit 4 must stay lane bit 4 and the read groups keep their pad-slot conflict (docs/HISTORY.md section 4.2);
                   // column bit 3 ^= lane bit 2 still spreads the stride-2 writes of transpose A over all eight slots
                   // (tools/lds_bank_sim.py: 16 -> 8 LDS cycles per store) and leaves transpose B's stores conflict-free
        return (l & ~8) | ((((l >> 3) ^ (l >> 2)) & 1) << 3);
    const int b3 = ((l >> 3) ^ ((LOGN == 10 || LOGN == 14) ? (l >> 1) : l)) & 1;
    return (l & ~0x18) | (b3 << 3) | (rg << 4);
}

} // namespace small

} // namespace muse
