// xcorr_stockham.hip -- radix-16 Stockham kernels for the FFT lengths the tuned
// n = 4096 kernels do not cover (BASELINE config 5: N in {512 ... 65536}).
//
// Mathematics: as xcorr_fused_n4096 (xcorr_kernels.hip header); reference path
// xCorrWithX, /root/reference/xcorr.go:160-197.  Two real series per complex
// transform, two FORWARD transforms per pair.
//
// Transform structure.  n = R1 * 16^(NP-1), R1 = 2, 4, 8 or 16.  Every thread owns 16
// points x[j + i*S], S = n/16.  Forward transform = NP Stockham (autosort, decimation in
// time) passes; pass p has radix R (R1 for the first, 16 after), Ns = product of the
// earlier radices, and for butterfly q computes
//     y[(q / Ns) * Ns * R + (q % Ns) + r * Ns] = sum_s W_R^(r s) W_(Ns R)^(s (q % Ns)) x[q + s * n / R]
// (Ns = 1: no twiddles; R < 16: a thread runs 16/R butterflies q = j + m*S on the registers
// m + s*(16/R)).  The last pass leaves X[j + r*S] in the thread that owns j: the natural
// layout again, so the spectrum multiply needs no exchange.
// The second transform runs the TRANSPOSED passes in reverse order (the DFT matrix is
// symmetric: F = M_NP ... M_1 = M_1^T ... M_NP^T): pass p^T reads the positions pass p
// wrote, applies the same radix-R DFT, multiplies by the twiddle AFTER the butterfly and
// writes the positions pass p read.  It therefore starts from the layout the first
// transform ended in and ends with cc[j + i*S] in register i of thread j.
// Twiddles W^(s m), s = 1..15: W^m, W^2m, W^4m, W^8m come from the context's W_65536 table
// (8 m * 65536/(16 Ns) < 32768: always inside the half-period table), the other eleven are
// products of two of them (2 ulp; the parity bar is 1e-6).
//
// xcorr_fused_stk_lds<LOGN> (n = 512, 1024, 2048): n/16 threads per pair, 4096/n pairs per
// 256-thread workgroup, one padded LDS work buffer per pair (pos + pos/16: the Ns = 1 pass
// writes with a lane stride of R1 slots), three passes per transform, 11 barriers per
// workgroup iteration; <13> (n = 8192): one pair per 512-thread workgroup, four passes.
// xcorr_fused_stk_4step<LOGN> (n = 16384 ... 65536): the long series as a four-step transform
// (radix-R1 sweeps in place in a global scratch slice, 4096-point rows on chip) -- what
// automatic selection uses.
// All of them bring both series of a pair to O(1) with exact powers of two before the shared
// transform (fft_device.h, pow2_inv_sigma) and apply 1/sigma to the winning value only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "stk_device.h"

namespace muse {


template <int LOGN>
__global__ __launch_bounds__(((1 << LOGN) / 16 > 256 ? (1 << LOGN) / 16 : 256), ((1 << LOGN) / 16 > 256 ? 1 : 2))
void xcorr_fused_stk_lds(const FusedParams p)
{
    using namespace occ4;
    using namespace stk;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;                 // threads per pair = stride between a thread's points
    constexpr int G = S >= 256 ? 1 : 256 / S; // pairs per workgroup iteration
    constexpr int TPB = S * G;                // 256 (n <= 4096) or 512 (n = 8192)
    constexpr int NP = (LOGN + 3) / 4;        // passes per transform: radix R1, then radix 16
    constexpr int R1 = n >> (4 * (NP - 1));   // first radix (2, 4, 8 or 16)
    constexpr int Q1 = 16 / R1;               // butterflies per thread in the radix-R1 pass
    constexpr int ROWS = S / 16;              // 16-lane rows per pair
    constexpr int BUF = n + n / 16;
    static_assert((LOGN >= 9 && LOGN <= 11) || LOGN == 13, "LDS Stockham kernel: n = 512, 1024, 2048, 8192");
    __shared__ double2 buf[G * BUF];
    __shared__ double red[(TPB / 16) * 4];      // per 16-lane row: {sum dA, sum dA^2, sum dB, sum dB^2}
    __shared__ double arg[(TPB / 16) * 2 * 3];  // per row and series: {max |cc|, signed value, index}
    const int t = threadIdx.x;
    const int g = t / S, j = t % S;
    const int row = t >> 4;             // global 16-lane row id; pair g owns rows g*ROWS .. +ROWS
    double2 *const b = buf + g * BUF;
    const int N = p.N, pad = n - N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);
    const double2 *__restrict__ twm = p.twm;
    // optional indirection (filter-and-refine Run): process pair_list[0 .. *pair_count) instead of every pair
    const long long total = p.pair_list ? (long long)*p.pair_count : p.npairs;
    const long long ngroups = (total + G - 1) / G;

    for (long long it = blockIdx.x; it < ngroups; it += gridDim.x) {
        const long long slot = it * G + g;
        const bool live = slot < total;
        const long long sl = live ? slot : total - 1; // idle sub-groups shadow the last pair
        const long long pair = p.pair_list ? p.pair_list[sl] : sl;
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        const double *__restrict__ ra = p.rows + rA * p.stride;
        const double *__restrict__ rb = p.rows + (hasB ? rA + 1 : rA) * p.stride;
        // ---- rows (leading zero pad), d = x - K with K the first sample, shifted statistics
        double2 v[16];
        const double KA = ra[0], KB = rb[0];
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = j + i * S - pad;
            const int ec = e < 0 ? 0 : e;
            double da = __builtin_nontemporal_load(ra + ec) - KA, db = __builtin_nontemporal_load(rb + ec) - KB;
            da = e >= 0 ? da : 0.0;
            db = e >= 0 ? db : 0.0;
            v[i] = make_double2(da, db);
            q[0] += da;
            q[1] = fma(da, da, q[1]);
            q[2] += db;
            q[3] = fma(db, db, q[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            q[k] = row_sum_dpp(q[k]);
        if ((t & 15) == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                red[row * 4 + k] = q[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double s = 0.0;
            for (int r = 0; r < ROWS; r++)
                s += red[(g * ROWS + r) * 4 + k];
            q[k] = s;
        }
        const Stat stA{q[0], q[1]}, stB{q[2], q[3]};
        bool zeroA, nanA, zeroB, nanB;
        const double varA0 = variance(stA, invN, invNm1, zeroA, nanA);
        const double varB0 = variance(stB, invN, invNm1, zeroB, nanB);
        const bool deadA = zeroA || nanA, deadB = zeroB || nanB || !hasB;
        // both series go into the shared transform at O(1): exact power-of-two scales close to 1/sigma
        // (r16_device.h, pow2_inv_sigma), folded into the mean removal; the variances scale along exactly
        const double sA = deadA ? 1.0 : pow2_inv_sigma(varA0), sB = deadB ? 1.0 : pow2_inv_sigma(varB0);
        const double varA = varA0 * sA * sA, varB = varB0 * sB * sB;
        const double mA = q[0] * invN * sA, mB = q[2] * invN * sB;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool valid = j + i * S - pad >= 0;
            v[i].x = (valid && !deadA) ? fma(v[i].x, sA, -mA) : 0.0;
            v[i].y = (valid && !deadB) ? fma(v[i].y, sB, -mB) : 0.0;
        }
        // forward transform, V = Z conj(X)/n, transposed transform: register i ends with cc[j + i S]
        lds_transforms<LOGN>(v, b, twm, j, [&](int r) __attribute__((always_inline)) { return p.xc[j + r * S]; });
        // ================= maxAbsIndex (xcorr.go:39-50) per series =================
        double ma = 0.0, mb = 0.0, sa = 0.0, sb = 0.0;
        int ia = 0x7fffffff, ib = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 16; i++) { // ascending i = ascending index for this thread
            const double aa = fabs(v[i].x), ab = fabs(v[i].y);
            if (aa > ma) { ma = aa; sa = v[i].x; ia = j + i * S; }
            if (ab > mb) { mb = ab; sb = v[i].y; ib = j + i * S; }
        }
        {
            const double rma = row_max_dpp(ma), rmb = row_max_dpp(mb);
            const int ca = row_min_i_dpp((ma == rma && rma > 0.0) ? ia : 0x7fffffff);
            const int cb = row_min_i_dpp((mb == rmb && rmb > 0.0) ? ib : 0x7fffffff);
            if (ca == 0x7fffffff) {
                if ((t & 15) == 0) {
                    arg[row * 6 + 0] = 0.0;
                    arg[row * 6 + 1] = 0.0;
                    arg[row * 6 + 2] = (double)0x7fffffff;
                }
            } else if (ia == ca && ma == rma) {
                arg[row * 6 + 0] = rma;
                arg[row * 6 + 1] = sa;
                arg[row * 6 + 2] = (double)ca;
            }
            if (cb == 0x7fffffff) {
                if ((t & 15) == 0) {
                    arg[row * 6 + 3] = 0.0;
                    arg[row * 6 + 4] = 0.0;
                    arg[row * 6 + 5] = (double)0x7fffffff;
                }
            } else if (ib == cb && mb == rmb) {
                arg[row * 6 + 3] = rmb;
                arg[row * 6 + 4] = sb;
                arg[row * 6 + 5] = (double)cb;
            }
        }
        if (j == 0) // cc[0] of both series, for the "nothing above zero" case (index 0, mv = cc[0])
            red[g * ROWS * 4] = v[0].x, red[g * ROWS * 4 + 1] = v[0].y;
        __syncthreads();
        if (j < 2 && live && (j == 0 || hasB)) {
            double best = 0.0, bsv = 0.0, bidx = (double)0x7fffffff;
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const double *a = arg + (g * ROWS + r) * 6 + 3 * j;
                if (a[0] > best || (a[0] == best && a[2] < bidx)) {
                    best = a[0];
                    bsv = a[1];
                    bidx = a[2];
                }
            }
            const double var = j == 0 ? varA : varB;
            const bool zero = j == 0 ? zeroA : zeroB, nan = j == 0 ? nanA : nanB;
            const int idx = (best > 0.0) ? (int)bidx : 0;
            double y = __builtin_amdgcn_rsq(var);
            y = y * fma(-0.5 * var * y, y, 1.5);
            y = y * fma(-0.5 * var * y, y, 1.5);
            double mv = ((best > 0.0) ? bsv : red[g * ROWS * 4 + j]) * y;
            int lag = idx > n / 2 ? idx - n : idx;
            if (zero) { mv = 0.0; lag = 0; }
            if (nan) { mv = __builtin_nan(""); lag = 0; }
            p.mv[rA + j] = mv;
            p.lag[rA + j] = lag;
        }
        __syncthreads(); // red / arg / buf free for the next iteration
    }
}

// ---------------------------------------------------------------------------
// xcorr_fused_stk_4step<LOGN> (n = R1 * 4096, R1 = 4, 8, 16): the long-series path as a
// four-step transform, n1 = R1 (in-thread), n2 = 4096 (on chip through LDS):
//   X[k1 + R1 k2] = sum_m2 W_4096^(m2 k2) [ W_n^(m2 k1) sum_m1 W_R1^(m1 k1) x[m1 4096 + m2] ]
// sweep 0: rows -> d = x - K into the workgroup's scratch slice, statistics;
// sweep 1: radix-R1 over m1 + twiddle, IN PLACE (a thread reads and writes the same
//          positions q + s 4096);
// rows:    for each k1: the 4096 points of row k1 -> lds_transforms<12> (forward, multiply by
//          xc[k1 + R1 k2], transposed forward) -> back to the row, in place;
// sweep 2: twiddle + radix-R1 over k1 -> cc[m1 4096 + m2], argmax on the fly.
// Scratch traffic: 3 writes + 3 reads of the pair (the four-pass kernel above: 14), one
// n-element slice per workgroup, and the butterflies of the 4096-point rows run on chip.
template <int LOGN>
__global__ __launch_bounds__(256, 2) void xcorr_fused_stk_4step(const FusedParams p)
{
    using namespace occ4;
    using namespace stk;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;
    constexpr int CH = S / 256;
    constexpr int R1 = n / 4096;
    constexpr int Q1 = 16 / R1;
    static_assert(LOGN >= 13 && LOGN <= 16, "four-step kernel: n = 8192 ... 65536");
    __shared__ double2 buf[4096 + 256];
    __shared__ double red[64];
    __shared__ int redi[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double2 *const Y = p.gscratch + (size_t)blockIdx.x * (size_t)n; // the workgroup's slice: one pair
    const int N = p.N, pad = n - N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);
    const double2 *__restrict__ twm = p.twm;

    // optional indirection (filter-and-refine Run): process pair_list[0 .. *pair_count) instead of every pair
    const long long total = p.pair_list ? (long long)*p.pair_count : p.npairs;
    for (long long slot = blockIdx.x; slot < total; slot += gridDim.x) {
        const long long pair = p.pair_list ? p.pair_list[slot] : slot;
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        const double *__restrict__ ra = p.rows + rA * p.stride;
        const double *__restrict__ rb = p.rows + (hasB ? rA + 1 : rA) * p.stride;
        const double KA = ra[0], KB = rb[0];
        // twiddle of sweeps 1 and 2: v[m + r Q1] *= W_n^(m2 r), m2 = j + m S
        const auto twiddle_rows = [&](double2 (&v)[16], const int j) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < Q1; m++) {
                const int m2 = j + m * S;
#pragma unroll
                for (int r = 1; r < R1; r++) {
                    const int e = (m2 * r * (65536 / n)) & 65535;
                    const double2 w = twm[e & 32767];
                    const double2 ws = e >= 32768 ? make_double2(-w.x, -w.y) : w;
                    v[m + r * Q1] = cmul(v[m + r * Q1], ws);
                }
            }
        };
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        bool zeroA = false, nanA = false, zeroB = false, nanB = false;
        double varA = 0.0, varB = 0.0;
        bool zero_dc = false;
        if (pad == 0) {
            // N == n: the mean is never needed before the transform (the centred series' DC bin is
            // exactly 0: bin 0 is zeroed through the spectrum multiplier), so sweep 1 reads the rows
            // directly and the statistics ride along -- no sweep 0.  A NaN / Inf series cannot be
            // isolated from its partner this way: such pairs take the general path below.
#pragma clang loop unroll(disable)
            for (int ch = 0; ch < CH; ch++) {
                const int j = t + 256 * ch;
                double2 v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const double da = __builtin_nontemporal_load(ra + j + i * S) - KA;
                    const double db = __builtin_nontemporal_load(rb + j + i * S) - KB;
                    v[i] = make_double2(da, db);
                    q[0] += da;
                    q[1] = fma(da, da, q[1]);
                    q[2] += db;
                    q[3] = fma(db, db, q[3]);
                }
                dft_small<R1>(v);
                twiddle_rows(v, j);
#pragma unroll
                for (int i = 0; i < 16; i++)
                    Y[j + i * S] = v[i];
            }
            block_sum<4>(q, red);
            const Stat stA{q[0], q[1]}, stB{q[2], q[3]};
            varA = variance(stA, invN, invNm1, zeroA, nanA);
            varB = variance(stB, invN, invNm1, zeroB, nanB);
            // block-uniform; sigmas too far apart for the unscaled shared transform also take the general path
            zero_dc = !(nanA || nanB) && !(hasB && sigma_spread_too_wide(varA, varB));
        }
        if (!zero_dc) {
            // ---- sweep 0: rows -> d = x - K (leading zero pad) into the slice, shifted statistics
            q[0] = q[1] = q[2] = q[3] = 0.0;
            __syncthreads(); // (N == n fallback: every thread is done with its sweep-1 stores)
#pragma clang loop unroll(disable)
            for (int ch = 0; ch < CH; ch++) {
                const int j = t + 256 * ch;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int e = j + i * S - pad;
                    const int ec = e < 0 ? 0 : e;
                    double da = __builtin_nontemporal_load(ra + ec) - KA, db = __builtin_nontemporal_load(rb + ec) - KB;
                    da = e >= 0 ? da : 0.0;
                    db = e >= 0 ? db : 0.0;
                    Y[j + i * S] = make_double2(da, db);
                    q[0] += da;
                    q[1] = fma(da, da, q[1]);
                    q[2] += db;
                    q[3] = fma(db, db, q[3]);
                }
            }
            block_sum<4>(q, red);
            const Stat stA{q[0], q[1]}, stB{q[2], q[3]};
            const double varA0 = variance(stA, invN, invNm1, zeroA, nanA);
            const double varB0 = variance(stB, invN, invNm1, zeroB, nanB);
            const bool deadA = zeroA || nanA, deadB = zeroB || nanB || !hasB;
            const double sA = deadA ? 1.0 : pow2_inv_sigma(varA0), sB = deadB ? 1.0 : pow2_inv_sigma(varB0); // see the LDS kernel
            varA = varA0 * sA * sA;
            varB = varB0 * sB * sB;
            const double mA = q[0] * invN * sA, mB = q[2] * invN * sB;
            // ---- sweep 1: radix R1 over m1 (butterflies m2 = j + m S on registers m + s Q1), twiddle
            // W_n^(m2 k1), in place (positions m2 + s 4096 = j + (m + s Q1) S)
#pragma clang loop unroll(disable)
            for (int ch = 0; ch < CH; ch++) {
                const int j = t + 256 * ch;
                double2 v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const bool valid = j + i * S - pad >= 0;
                    const double2 d = Y[j + i * S];
                    v[i].x = (valid && !deadA) ? fma(d.x, sA, -mA) : 0.0;
                    v[i].y = (valid && !deadB) ? fma(d.y, sB, -mB) : 0.0;
                }
                dft_small<R1>(v);
                twiddle_rows(v, j);
#pragma unroll
                for (int i = 0; i < 16; i++)
                    Y[j + i * S] = v[i]; // register m + r Q1 <-> row k1 = r, column m2: position r 4096 + m2
            }
        }
        __syncthreads();
        // ---- rows: k1 = 0 .. R1-1, 4096 points each, on chip
#pragma clang loop unroll(disable)
        for (int k1 = 0; k1 < R1; k1++) {
            double2 *const row = Y + k1 * 4096;
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = row[t + 256 * i];
            const bool dc = zero_dc && k1 == 0 && t == 0; // this thread's output r = 0 is bin 0 of the pair
            lds_transforms<12>(v, buf, twm, t, [&](int r) __attribute__((always_inline)) {
                const double2 x = p.xc[k1 + R1 * (t + 256 * r)];
                return (dc && r == 0) ? make_double2(0.0, 0.0) : x;
            });
#pragma unroll
            for (int i = 0; i < 16; i++)
                row[t + 256 * i] = v[i];
        }
        __syncthreads();
        // ---- sweep 2: twiddle, radix R1 over k1 -> cc[m1 4096 + m2] at register m + m1 Q1; argmax
        double ma = 0.0, mb = 0.0, sa = 0.0, sb = 0.0, cc0a = 0.0, cc0b = 0.0;
        int ia = 0x7fffffff, ib = 0x7fffffff;
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = t + 256 * ch;
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = Y[j + i * S];
            twiddle_rows(v, j);
            dft_small<R1>(v);
            if (ch == 0) {
                cc0a = v[0].x;
                cc0b = v[0].y;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const double aa = fabs(v[i].x), ab = fabs(v[i].y);
                const int idx = j + i * S;
                if (aa > ma || (aa == ma && aa > 0.0 && idx < ia)) { ma = aa; sa = v[i].x; ia = idx; }
                if (ab > mb || (ab == mb && ab > 0.0 && idx < ib)) { mb = ab; sb = v[i].y; ib = idx; }
            }
        }
        // ---- block argmax (first index of the maximum), owner thread stores
        {
            const double wa = wave_max(ma), wb = wave_max(mb);
            if (lane == 0) {
                red[32 + wave] = wa;
                red[36 + wave] = wb;
            }
            if (t == 0) {
                red[40] = cc0a;
                red[41] = cc0b;
            }
            __syncthreads();
            const double MA = fmax(fmax(red[32], red[33]), fmax(red[34], red[35]));
            const double MB = fmax(fmax(red[36], red[37]), fmax(red[38], red[39]));
            int ca = (ma == MA && MA > 0.0) ? ia : 0x7fffffff;
            int cb = (mb == MB && MB > 0.0) ? ib : 0x7fffffff;
            ca = wave_min_i(ca);
            cb = wave_min_i(cb);
            if (lane == 0) {
                redi[wave] = ca;
                redi[4 + wave] = cb;
            }
            __syncthreads();
            const int IA = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
            const int IB = min(min(redi[4], redi[5]), min(redi[6], redi[7]));
            for (int sidx = 0; sidx < 2; sidx++) {
                if (sidx == 1 && !hasB)
                    break;
                const int I = sidx ? IB : IA;
                const bool none = I == 0x7fffffff;
                const bool owner = none ? (t == 0) : ((sidx ? ib : ia) == I && (sidx ? mb : ma) == (sidx ? MB : MA));
                if (owner) {
                    const double var = sidx ? varB : varA;
                    const bool zero = sidx ? zeroB : zeroA, nan = sidx ? nanB : nanA;
                    double y = __builtin_amdgcn_rsq(var);
                    y = y * fma(-0.5 * var * y, y, 1.5);
                    y = y * fma(-0.5 * var * y, y, 1.5);
                    const int idx = none ? 0 : I;
                    double mv = (none ? red[40 + sidx] : (sidx ? sb : sa)) * y;
                    int lag = idx > n / 2 ? idx - n : idx;
                    if (zero) { mv = 0.0; lag = 0; }
                    if (nan) { mv = __builtin_nan(""); lag = 0; }
                    p.mv[rA + sidx] = mv;
                    p.lag[rA + sidx] = lag;
                }
            }
            __syncthreads();
        }
    }
}

template <int LOGN>
static hipError_t launch_stk_4step(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (!p.gscratch)
        return hipErrorInvalidValue;
    // resident workgroups, each with an n-element scratch slice the pair crosses four times
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * STOCKHAM_GLOBAL_WGS_PER_CU);
    if (grid > p.gscratch_slices)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL((xcorr_fused_stk_4step<LOGN>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_stk_lds(const FusedParams &p, int num_cus, hipStream_t stream)
{
    constexpr int S = (1 << LOGN) / 16;
    constexpr int G = S >= 256 ? 1 : 256 / S;
    constexpr int TPB = S * G;
    constexpr int WPC = TPB > 256 ? 1 : 2; // workgroups per CU the LDS buffer allows
    const long long ngroups = (p.npairs + G - 1) / G;
    const long long grid = std::min<long long>(ngroups, (long long)num_cus * WPC * 8);
    hipLaunchKernelGGL((xcorr_fused_stk_lds<LOGN>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
    return hipGetLastError();
}

// n = 512, 1024, 2048 (LDS) and 8192 ... 65536 (global scratch: 2 n complex per workgroup); any N in (n/2, n]
hipError_t launch_fused_stockham(const FusedParams &p0, int num_cus, hipStream_t stream)
{
    const FusedParams &p = p0;
    switch (p.logn) {
    case 9: return launch_stk_lds<9>(p, num_cus, stream);
    case 10: return launch_stk_lds<10>(p, num_cus, stream);
    case 11: return launch_stk_lds<11>(p, num_cus, stream);
    case 13: return launch_stk_lds<13>(p, num_cus, stream);
    case 14: return launch_stk_4step<14>(p, num_cus, stream);
    case 15: return launch_stk_4step<15>(p, num_cus, stream);
    case 16: return launch_stk_4step<16>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace muse
