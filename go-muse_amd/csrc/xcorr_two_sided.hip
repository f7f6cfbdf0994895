// xcorr_two_sided.hip -- the batched two-sided xCorr.
//
// Mathematics: xCorr, /root/reference/xcorr.go:102-153 -- for each of M independent pairs (x, y): optional zNormalize of
// both (xcorr.go:108-128; sigma == 0 -> (nil, 0, 0)), leading zero pad of both to n (129-130), X = FFT(x), Y = FFT(y),
// cc = IFFT(X conj(Y)) (134-138), scale 1 / (n (n - 1)) when normalized, else 1 / n (139-143) -- n - 1 of the FFT length,
// not of the series length --, global first-strict argmax of |cc| and lag unwrap (145-150).
//
// n <= 16384: the squared-spectrum form (xcorr_two_sided_fold below for n = 4096, xcorr_small.hip for the other lengths).
// n >= 32768 (four-step kernel at the end of this file): one complex transform serves both series of a pair:
// z = x + i y, Z = FFT(z).  With Zm[f] = Z[-f mod n]
//     X[f] = (Z[f] + conj(Zm[f])) / 2,   Y[f] = (Z[f] - conj(Zm[f])) / 2i
//     P[f] = X[f] conj(Y[f]):   Re P = Im(Z[f] Zm[f]) / 2,   Im P = (|Z[f]|^2 - |Zm[f]|^2) / 4
// and cc = IFFT(P) is real, so FFT(conj(P) / n) = cc: a pair costs two FORWARD complex transforms of length n, exactly
// what a pair of series costs in the xCorrWithX kernels.  The transforms are the Stockham engine's (stk_device.h: natural
// order, folded arithmetic), whose output order makes the mirrored element Z[-f] one scratch read away.  Both series enter
// the shared transform at O(1): exact power-of-two scales near 1 / sigma (or near 1 / rms when not normalized), undone in
// the one factor the winning value is multiplied by (two_device.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "stk_device.h"
#include "foldk_device.h"
#include "two_device.h"

namespace muse {

namespace two {

using namespace occ4;
using namespace stk;

// V[f] = conj(P[f]) * s from Z[f] and Z[-f]
__device__ __forceinline__ double2 untangle(const double2 z, const double2 zm, const double s)
{
    const double re = 0.5 * fma(z.x, zm.y, z.y * zm.x);
    const double im = 0.25 * (fma(z.x, z.x, z.y * z.y) - fma(zm.x, zm.x, zm.y * zm.y));
    return make_double2(re * s, -im * s);
}

} // namespace two

// n = 4096 on the xCorrWithX kernel's machinery (foldk_device.h: three radix-16 passes per transform with the twiddles
// folded into the butterflies, half-round LDS transposes, 128 registers -> four workgroups per CU) WITHOUT the mirrored
// element: with xr[j] = x[-j mod n] (x read backwards -- an address pattern, not an instruction) conj(X) = FFT(xr), so
//     cc = FFT(conj(X) Y) / n = FFT(FFT(xr) FFT(y)) / n,      and with z = xr + i y,  Z = FFT(z) = Xr + i Y:
//     Z^2 = (Xr^2 - Y^2) + 2 i Xr Y        ->        cc = Im FFT(Z^2) / (2 n)
// (Xr^2, Y^2 and Xr Y are spectra of real sequences, so their forward transforms are real).  A pair costs the same two
// forward transforms as in the kernels above, but no spectrum table, no Z[-f] and no exchange to fetch it: the square
// rides in the first butterfly stage of the second transform (fold_device.h, bf_sq).  The real part, xr * xr - y * y,
// is computed and dropped.  Statistics first (one workgroup barrier), as in the kernels above: both series enter the shared
// transform centred and at O(1).  The previous pair's four wave records are combined behind the next pair's statistics
// barrier (no barrier of its own).
// PADDED = false: Nx == Ny == n (every position valid: shared lane offsets + immediates, no masks).
// iv: the launch's reciprocals, formed on the host (IEEE division both sides) so that they arrive in scalar registers.
template <bool PADDED>
__global__ __launch_bounds__(256, 4) void xcorr_two_sided_fold(const FusedParams p, const two::PairInv iv)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    using namespace two;
    constexpr int n = 4096;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[16];
    __shared__ double arg[2][16]; // per parity: four waves x {max, value, index}, [12] = cc[0], [13] = factor, [14] = nil, [15] = NaN
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + XW * wave;
    const int Nx = p.Nx, Ny = p.N, padx = n - Nx, pady = n - Ny;
    const bool normalize = p.normalize_y != 0;
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();
    int parity = 0;
    // the pair before `pair` (this workgroup's previous one): its four wave records, cc[0], factor and flags sit in LDS -- no
    // register lives across the transforms for it
    const auto finish_prev = [&](const long long prev_pair) __attribute__((always_inline)) {
        if (t == 0 && prev_pair >= 0) {
            const double *a = arg[parity ^ 1];
            double best = 0.0, bsv = 0.0, bidx = (double)0x7fffffff;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if (a[3 * w] > best || (a[3 * w] == best && a[3 * w + 2] < bidx)) {
                    best = a[3 * w];
                    bsv = a[3 * w + 1];
                    bidx = a[3 * w + 2];
                }
            }
            const int idx = (best > 0.0) ? (int)bidx : 0;
            double mv = ((best > 0.0) ? bsv : a[12]) * a[13];
            int lag = idx > n / 2 ? idx - n : idx;
            const bool nil = a[14] != 0.0, nan = a[15] != 0.0;
            if (nil) { mv = 0.0; lag = 0; }                 // xcorr.go:110-127
            if (nan) { mv = __builtin_nan(""); lag = 0; }   // every cc is NaN: maxAbsIndex keeps index 0
            p.mv[prev_pair] = mv;
            p.lag[prev_pair] = lag;
            if (p.nil_out)
                p.nil_out[prev_pair] = nil ? 1 : 0;
        }
    };

    long long pair = blockIdx.x;
    for (; pair < p.npairs; pair += gridDim.x) {
        double2 v[16];
        {
            const double *const rx = p.xrows + pair * p.xstride, *const ry = p.rows + pair * p.stride;
            typedef const double __attribute__((address_space(4))) *cptr; // the shift constants through the scalar cache
            const double KA = normalize ? *(cptr)(unsigned long long)rx : 0.0, KB = normalize ? *(cptr)(unsigned long long)ry : 0.0;
            double q[4] = {0.0, 0.0, 0.0, 0.0};
            // position e = t + 256 i of the padded arrays holds y[e - pady] and x[(-e mod n) - padx] (leading zero pads,
            // xcorr.go:129-130); four batches of four positions bound the registers in flight
#pragma unroll
            for (int h = 0; h < 4; h++) {
                double xa[4], yb[4];
                int tb = t;
                if (PADDED)
                    asm volatile("" : "+v"(tb)); // (a batch's offsets and masks are formed in the batch, not hoisted in front of all four)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = 4 * h + k;
                    if (PADDED) {
                        const int e = tb + 256 * i;
                        const int ex = ((n - e) & (n - 1)) - padx, ey = e - pady;
                        xa[k] = __builtin_nontemporal_load(scalar_ptr(rx) + (unsigned)(ex < 0 ? 0 : ex));
                        yb[k] = __builtin_nontemporal_load(scalar_ptr(ry) + (unsigned)(ey < 0 ? 0 : ey));
                    } else {
                        // y[t + 256 i]; x[4096 - 256 i - t] = x[256 (15 - i) + (256 - t)] for i >= 1; position 0 reads x[0]
                        if (i == 0)
                            xa[k] = __builtin_nontemporal_load(scalar_ptr(rx) + (t == 0 ? 0u : (unsigned)(4096 - t)));
                        else
                            xa[k] = __builtin_nontemporal_load(scalar_ptr_at(rx, 256 * (15 - i)) + (unsigned)(256 - t));
                        yb[k] = __builtin_nontemporal_load(scalar_ptr_at(ry, 256 * i) + (unsigned)t);
                    }
                }
                fence();
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = 4 * h + k;
                    double da = xa[k] - KA, db = yb[k] - KB;
                    if (PADDED) {
                        const int e = tb + 256 * i;
                        da = ((n - e) & (n - 1)) - padx >= 0 ? da : 0.0;
                        db = e - pady >= 0 ? db : 0.0;
                    }
                    v[i] = make_double2(da, db);
                    q[0] += da;
                    q[1] = fma(da, da, q[1]);
                    q[2] += db;
                    q[3] = fma(db, db, q[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                q[k] = wave_sum_dpp(q[k]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    red[4 * wave + k] = q[k];
            }
            lds_barrier();
            finish_prev(pair - gridDim.x); // (its records were complete before this barrier)
            {   // the four waves' partial sums: ONE LDS read (lane l takes red[l & 15]: sixteen different addresses per read group --
                // 64 lanes on one address are not a free broadcast: SQ_LDS_BANK_CONFLICT was 10 % of this kernel's LDS cycles),
                // lanes k + 4 w added up inside each 16-lane row by two DPP shifts, lanes 0 - 3 read into scalar registers
                double r = red[lane & 15];
                r += dpp_f64<0x108>(r); // row_shl:8 (zero fill): lane i += lane i + 8
                r += dpp_f64<0x104>(r); // row_shl:4:             lane i += lane i + 4
#pragma unroll
                for (int k = 0; k < 4; k++)
                    q[k] = readlane_f64(r, k);
            }
            const PairScale ps = pair_scale(q, iv, normalize);
            const bool dead = ps.nil || ps.nan;
            if (t == 0) {
                arg[parity][13] = ps.fac * (1.0 / (2.0 * n)); // (pair_scale's factor assumes a spectrum already divided by n; 1 / 2n is exact)
                arg[parity][14] = ps.nil ? 1.0 : 0.0;
                arg[parity][15] = ps.nan ? 1.0 : 0.0;
            }
            const double sA = dead ? 0.0 : ps.sA, sB = dead ? 0.0 : ps.sB, mA = dead ? 0.0 : ps.mA, mB = dead ? 0.0 : ps.mB;
            int tt = t;
            asm volatile("" : "+v"(tt)); // (the validity masks are recomputed here, not kept in 32 register pairs across the statistics)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = tt + 256 * i;
                const bool vx = !PADDED || ((n - e) & (n - 1)) - padx >= 0, vy = !PADDED || e - pady >= 0;
                v[i].x = vx ? fma(v[i].x, sA, -mA) : 0.0;
                v[i].y = vy ? fma(v[i].y, sB, -mB) : 0.0;
            }
        }
        // ---- Z = FFT(xr + i y): Z[hi + 16 lo + 256 k3] at v[BR16(k3)] (as xcorr_fused_n4096_fold)
        dft16_nr(v);
        exchange_cross<0, 1, true>(v, xbuf, wave, t);
        gdft16_nr(v, G2Fetch{g2s, t >> 4});
        exchange_local<1>(v, xw, t);
        gdft16_nr_l2(v, G3Derived(p.g3a, t));
        // ---- FFT(Z^2): the square in the first stage of the plain pass
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            bf_sq(v[r], v[r + 1]);
        dft16_rn_s234(v);
        exchange_local<0>(v, xw, t);
        gdft16_nr(v, G2Fetch{g2s, t & 15});
        exchange_cross<1, 1>(v, xbuf, wave, t); // (the factor and flags written before the transforms are visible behind these barriers)
        gdft16_nr_l2(v, G3Derived(p.g3b, t)); // 2 n cc[t + 256 m] = v[BR16(m)].y
        if (p.cc_out && arg[parity][14] == 0.0 && arg[parity][15] == 0.0) {
            const double fac = arg[parity][13];
            double *const cc = p.cc_out + pair * (long long)n;
#pragma unroll
            for (int m = 0; m < 16; m++)
                cc[t + 256 * m] = v[BR16(m)].y * fac;
        }
        wave_argmax_store_one<1>(v, wave, lane, arg[parity] + 3 * wave);
        if (t == 0)
            arg[parity][12] = v[0].y;
        parity ^= 1;
    }
    lds_barrier();
    finish_prev(pair - gridDim.x);
}

// n = 16384 ... 65536: four-step, n = R1 * 4096 (xcorr_fused_stk_4step's geometry), TWO n-element scratch slices per
// workgroup: Y takes the rows, the first sweep and the row spectra Z (natural order: Z[k1 + R1 k2] at Y[4096 k1 + k2]);
// the second stage reads Z[f] and Z[-f] (row R1 - k1, column 4095 - k2; row 0: column 4096 - k2), transforms the untangled
// product row by row into Y2, and the last sweep runs over Y2.
template <int LOGN>
__global__ __launch_bounds__(256, 2) void xcorr_two_sided_4step(const FusedParams p)
{
    using namespace occ4;
    using namespace stk;
    using namespace two;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;
    constexpr int CH = S / 256;
    constexpr int R1 = n / 4096;
    constexpr int Q1 = 16 / R1;
    static_assert(LOGN >= 15 && LOGN <= 16, "four-step kernel: n = 32768, 65536");
    __shared__ double2 buf[4096 + 256];
    __shared__ double red[64];
    __shared__ int redi[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double2 *const Y = p.gscratch + (size_t)(2 * blockIdx.x) * (size_t)n;
    double2 *const Y2 = Y + n;
    const int Nx = p.Nx, Ny = p.N, padx = n - Nx, pady = n - Ny;
    const bool normalize = p.normalize_y != 0;
    const double2 *__restrict__ twm = p.twm;

    for (long long pair = blockIdx.x; pair < p.npairs; pair += gridDim.x) {
        const double *__restrict__ rx = p.xrows + pair * p.xstride;
        const double *__restrict__ ry = p.rows + pair * p.stride;
        const double KA = normalize ? rx[0] : 0.0, KB = normalize ? ry[0] : 0.0;
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
        // ---- sweep 0: rows -> d (leading zero pads) into the slice, statistics
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = t + 256 * ch;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int ex = j + i * S - padx, ey = j + i * S - pady;
                double da = __builtin_nontemporal_load(rx + (ex < 0 ? 0 : ex)) - KA;
                double db = __builtin_nontemporal_load(ry + (ey < 0 ? 0 : ey)) - KB;
                da = ex >= 0 ? da : 0.0;
                db = ey >= 0 ? db : 0.0;
                Y[j + i * S] = make_double2(da, db);
                q[0] += da;
                q[1] = fma(da, da, q[1]);
                q[2] += db;
                q[3] = fma(db, db, q[3]);
            }
        }
        block_sum<4>(q, red);
        const PairScale ps = pair_scale(q, Nx, Ny, n, normalize);
        const bool dead = ps.nil || ps.nan;
        // ---- sweep 1: scale / centre, radix R1 over m1, twiddle W_n^(m2 k1), in place
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = t + 256 * ch;
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool vx = j + i * S - padx >= 0, vy = j + i * S - pady >= 0;
                const double2 d = Y[j + i * S];
                v[i].x = (vx && !dead) ? fma(d.x, ps.sA, -ps.mA) : 0.0;
                v[i].y = (vy && !dead) ? fma(d.y, ps.sB, -ps.mB) : 0.0;
            }
            dft_small<R1>(v);
            twiddle_rows(v, j);
#pragma unroll
            for (int i = 0; i < 16; i++)
                Y[j + i * S] = v[i];
        }
        __syncthreads();
        // ---- rows, first transform: row k1 -> Z[k1 + R1 k2] at Y[4096 k1 + k2]
#pragma clang loop unroll(disable)
        for (int k1 = 0; k1 < R1; k1++) {
            double2 *const row = Y + k1 * 4096;
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = row[t + 256 * i];
            lds_forward<12, true>(v, buf, twm, t);
#pragma unroll
            for (int r = 0; r < 16; r++)
                row[t + 256 * r] = v[BR16(r)];
        }
        __syncthreads(); // every row's spectrum is in the slice
        // ---- rows, second transform: V[f] = conj(X[f] conj(Y[f])) / n from Z[f] and Z[-f], row k1 -> Y2
#pragma clang loop unroll(disable)
        for (int k1 = 0; k1 < R1; k1++) {
            const double2 *const row = Y + k1 * 4096;
            const double2 *const mrow = Y + ((R1 - k1) & (R1 - 1)) * 4096;
            double2 v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k2 = t + 256 * r;
                const int m2 = k1 == 0 ? ((4096 - k2) & 4095) : 4095 - k2;
                v[r] = untangle(row[k2], mrow[m2], 1.0 / (double)n);
            }
            lds_forward<12, true>(v, buf, twm, t);
            double2 *const out = Y2 + k1 * 4096;
#pragma unroll
            for (int r = 0; r < 16; r++)
                out[t + 256 * r] = v[BR16(r)];
        }
        __syncthreads();
        // ---- sweep 2: twiddle, radix R1 over k1 -> cc[m1 4096 + m2] at register m + m1 Q1 (real part); argmax
        double ma = 0.0, sa = 0.0, cc0 = 0.0;
        int ia = 0x7fffffff;
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = t + 256 * ch;
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = Y2[j + i * S];
            twiddle_rows(v, j);
            dft_small<R1>(v);
            if (ch == 0)
                cc0 = v[0].x;
            if (p.cc_out && !dead) {
                double *const cc = p.cc_out + pair * (long long)n;
#pragma unroll
                for (int i = 0; i < 16; i++)
                    cc[j + i * S] = v[i].x * ps.fac;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const double aa = fabs(v[i].x);
                const int idx = j + i * S;
                if (aa > ma || (aa == ma && aa > 0.0 && idx < ia)) { ma = aa; sa = v[i].x; ia = idx; }
            }
        }
        {
            const double wa = wave_max(ma);
            if (lane == 0)
                red[32 + wave] = wa;
            if (t == 0)
                red[40] = cc0;
            __syncthreads();
            const double MA = fmax(fmax(red[32], red[33]), fmax(red[34], red[35]));
            int ca = (ma == MA && MA > 0.0) ? ia : 0x7fffffff;
            ca = wave_min_i(ca);
            if (lane == 0)
                redi[wave] = ca;
            __syncthreads();
            const int IA = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
            const bool none = IA == 0x7fffffff;
            const bool owner = none ? (t == 0) : (ia == IA && ma == MA);
            if (owner) {
                const int idx = none ? 0 : IA;
                double mv = (none ? red[40] : sa) * ps.fac;
                int lag = idx > n / 2 ? idx - n : idx;
                if (ps.nil) { mv = 0.0; lag = 0; }
                if (ps.nan) { mv = __builtin_nan(""); lag = 0; }
                p.mv[pair] = mv;
                p.lag[pair] = lag;
                if (p.nil_out)
                    p.nil_out[pair] = ps.nil ? 1 : 0;
            }
            __syncthreads();
        }
    }
}

template <int LOGN>
static hipError_t launch_two_4step(const FusedParams &p, int num_cus, hipStream_t stream)
{
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * STOCKHAM_GLOBAL_WGS_PER_CU);
    if (!p.gscratch || 2 * grid > p.gscratch_slices) // two n-element slices per workgroup
        return hipErrorInvalidValue;
    hipLaunchKernelGGL((xcorr_two_sided_4step<LOGN>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// M = p.npairs pairs (x_i = p.xrows + i p.xstride, length p.Nx; y_i = p.rows + i p.stride, length p.N), FFT length
// p.n = 2^p.logn in 512 ... 65536 (>= both lengths), p.normalize_y = the reference's `normalize`; results in p.mv / p.lag /
// p.nil_out (optional) / p.cc_out (optional, M x n)
hipError_t launch_two_sided(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (!p.xrows || !p.rows || !p.twm || !p.mv || !p.lag || p.npairs < 1 || p.Nx < 1 || p.N < 1 || p.Nx > p.n || p.N > p.n ||
        (p.normalize_y && (p.Nx < 2 || p.N < 2)))
        return hipErrorInvalidValue;
    switch (p.logn) {
    case 9:
    case 10:
    case 11:
    case 13:
    case 14:
        return launch_two_sided_small(p, num_cus, stream);
    case 12: {
        if (!p.g2 || !p.g3a || !p.g3b)
            return hipErrorInvalidValue;
        const long long grid = std::min<long long>(p.npairs, (long long)num_cus * 4);
        const two::PairInv iv = two::pair_inv(p.Nx, p.N, 4096);
        if (p.Nx == 4096 && p.N == 4096)
            hipLaunchKernelGGL(xcorr_two_sided_fold<false>, dim3((unsigned)grid), dim3(256), 0, stream, p, iv);
        else
            hipLaunchKernelGGL(xcorr_two_sided_fold<true>, dim3((unsigned)grid), dim3(256), 0, stream, p, iv);
        return hipGetLastError();
    }
    case 15: return launch_two_4step<15>(p, num_cus, stream);
    case 16: return launch_two_4step<16>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace muse
