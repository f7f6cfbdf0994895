// xcorr_two_sided.hip -- the batched two-sided xCorr.
//
// Mathematics: xCorr, /root/reference/xcorr.go:102-153 -- for each of M independent pairs (x, y): optional zNormalize of
// both (xcorr.go:108-128; sigma == 0 -> (nil, 0, 0)), leading zero pad of both to n (129-130), X = FFT(x), Y = FFT(y),
// cc = IFFT(X conj(Y)) (134-138), scale 1 / (n (n - 1)) when normalized, else 1 / n (139-143) -- n - 1 of the FFT length,
// not of the series length --, global first-strict argmax of |cc| and lag unwrap (145-150).
//
// Every batched length runs the squared-spectrum form: with xr[j] = x[-j mod n] (x read backwards: an address pattern) and
// z = xr + i y, Z = FFT(z) = Xr + i Y and cc = Im FFT(Z^2) / 2n -- two FORWARD complex transforms of length n per pair, exactly
// what a pair of series costs in the xCorrWithX kernels, no spectrum table and no mirrored element Z[-f]: n = 4096 on
// xcorr_two_sided_fold below, n = 512 ... 2048, 8192, 16384 on xcorr_small.hip's transforms, n = 32768, 65536 on the long-series
// kernel's four-step transform (xcorr_two_sided_long at the end of this file).  Both series enter the shared transform centred and
// at O(1): exact power-of-two scales near 1 / sigma (or near 1 / rms when not normalized), undone in the one factor the winning
// value is multiplied by (two_device.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "long_device.h"
#include "two_device.h"

namespace muse {

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

// n = 32768, 65536: the squared-spectrum form on the long-series kernel's four-step transform (xcorr_long.hip: n = R1 * 4096,
// sweeps of radix R1 in registers with their twiddles formed as powers of one table entry, 4096-point rows on the n = 4096
// kernel's folded transforms at 16 waves per CU, ONE n-element scratch slice per workgroup):
//   pass 0  the pair's statistics (both rows read once; xcorr.go:108-128) -> exact power-of-two scales, means, the result factor;
//   sweep 1 the rows again, x backwards: z[e] = (x[-e mod n] s_x - m_x) + i (y[e] s_y - m_y) (leading zero pads of either series
//           masked), radix R1 over m1, twiddle -> the slice;
//   rows    first transform, the element-wise SQUARE in the first stage of the second transform (bf_sq), second transform, in place;
//   sweep 2 twiddle, radix R1 over k1: 2 n cc = the IMAGINARY part; argmax.
// Round 3 ran these lengths on the Stockham four-step with the mirrored element Z[-f] (two slices per workgroup, nine crossings of a
// slice per pair, 8 waves per CU): 0.06 of the roofline; this form crosses the slice four times and reads the rows twice.
// PRE = true: as described (any pad geometry; also the kernel that redoes a listed pair).  PRE = false (Nx = Ny = n): the rows
// are read ONCE -- sweep 1 takes d = sample - first sample unscaled and the statistics beside it, the means go with bin 0 of the
// spectrum (Z[0] = sum dx + i sum dy: zeroed behind the first transform of row 0; N = n: no pad to keep at zero), the scale
// 1 / (sigma_x sigma_y) comes out of Z^2 as one factor -- and a pair whose two series differ too much in scale for one unscaled square
// (or whose magnitudes are extreme) is LISTED and redone by the PRE kernel in a second launch bounded by the on-device count, as the
// xCorrWithX kernels do with their sigma-spread pairs: 5 instead of 6 crossings of 16 n bytes per pair.
template <int LOGN, bool PADDED, bool PRE>
__global__ __launch_bounds__(256, 4) void xcorr_two_sided_long(const FusedParams p, const two::PairInv iv)
{
    static_assert(PRE || !PADDED, "the single-read form needs N = n");
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    using namespace lng;
    using namespace two;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;
    constexpr int CH = S / 256;
    constexpr int R1 = n / 4096;
    constexpr int Q1 = 16 / R1;
    constexpr int NW = 4;
    static_assert(LOGN >= 15 && LOGN <= 16, "n = 65536 (n = 32768 is built too, but runs on xcorr_real.hip since round 5)");
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[4 * NW + NW + 2];
    __shared__ int redi[NW];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + XW * wave;
    double2 *const Y = p.gscratch + (size_t)blockIdx.x * (size_t)n;
    const int Nx = PADDED ? p.Nx : n, Ny = PADDED ? p.N : n, padx = n - Nx, pady = n - Ny;
    const bool normalize = p.normalize_y != 0;
    const double2 *__restrict__ twl = p.twl; // [4096] W_n^(m2)
    typedef d2v __attribute__((address_space(1))) *gd2;
    const auto yat = [&](long long off) __attribute__((always_inline)) { return (gd2)scalar_ptr_at(Y, off); };
    const auto opaque = [](int x) __attribute__((always_inline)) {
        asm volatile("" : "+v"(x));
        return x;
    };
    const auto tw_base = [&](int m, unsigned jj) __attribute__((always_inline)) { return ldg2u(scalar_ptr_at(twl, m * S), jj); };
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();

    // optional indirection (PRE): the pairs the single-read launch listed
    const long long total = p.pair_list ? (long long)*p.pair_count : p.npairs;
    for (long long it = blockIdx.x; it < total; it += gridDim.x) {
        const long long pair = p.pair_list ? p.pair_list[it] : it;
        const double *__restrict__ rx = p.xrows + pair * p.xstride;
        const double *__restrict__ ry = p.rows + pair * p.stride;
        const double KA = normalize ? rx[0] : 0.0, KB = normalize ? ry[0] : 0.0;
        // position e of the padded arrays holds y[e - pady] and x[(-e mod n) - padx] (leading zero pads, xcorr.go:129-130)
        const auto load_chunk = [&](const int j, double (&xa)[16], double (&yb)[16]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = j + i * S;
                const int ex = ((n - e) & (n - 1)) - padx, ey = e - pady;
                xa[i] = __builtin_nontemporal_load(scalar_ptr(rx) + (unsigned)(PADDED && ex < 0 ? 0 : ex));
                if (PADDED)
                    yb[i] = __builtin_nontemporal_load(scalar_ptr(ry) + (unsigned)(ey < 0 ? 0 : ey));
                else
                    yb[i] = __builtin_nontemporal_load(scalar_ptr_at(ry, i * S) + (unsigned)j);
            }
        };
        // ---------------- pass 0 (PRE): statistics (the wave's running sums in SGPRs)
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma clang loop unroll(disable)
        for (int ch = 0; PRE && ch < CH; ch++) {
            const int j = opaque(t + 256 * ch) & (S - 1);
            double xa[16], yb[16], c[4] = {0.0, 0.0, 0.0, 0.0};
            load_chunk(j, xa, yb);
            fence();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                double da = xa[i] - KA, db = yb[i] - KB;
                if (PADDED) {
                    const int e = j + i * S;
                    da = ((n - e) & (n - 1)) - padx >= 0 ? da : 0.0;
                    db = e - pady >= 0 ? db : 0.0;
                }
                c[0] += da;
                c[1] = fma(da, da, c[1]);
                c[2] += db;
                c[3] = fma(db, db, c[3]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                q[k] = uniform(q[k] + wave_sum_dpp(c[k]));
        }
        // the block's sums -> what the statistics decide
        const auto block_stats = [&]() __attribute__((always_inline)) {
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    red[4 * wave + k] = q[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; k++)
                q[k] = uniform((red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]));
        };
        PairScale ps{1.0, 1.0, 0.0, 0.0, 1.0, false, false};
        if (PRE) {
            block_stats();
            ps = pair_scale(q, iv, normalize);
        }
        bool dead = ps.nil || ps.nan, redo = false;
        const double sA = dead ? 0.0 : ps.sA, sB = dead ? 0.0 : ps.sB, mA = dead ? 0.0 : ps.mA, mB = dead ? 0.0 : ps.mB;
        double fac = ps.fac * (1.0 / (2.0 * n)); // (pair_scale's factor assumes a spectrum already divided by n; 1 / 2n is exact)
        // ---------------- sweep 1: the rows again, scaled and centred, radix R1 over m1, twiddle -> the slice
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = opaque(t + 256 * ch) & (S - 1);
            double2 v[16];
            double xa[16], yb[16];
            load_chunk(j, xa, yb);
            double2 wb[Q1];
            {
                const unsigned jw = (unsigned)(opaque(t + 256 * ch) & (S - 1));
#pragma unroll
                for (int m = 0; m < Q1; m++)
                    wb[m] = tw_base(m, jw);
            }
            fence();
            if (PRE) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int e = j + i * S;
                    const bool vx = !PADDED || ((n - e) & (n - 1)) - padx >= 0, vy = !PADDED || e - pady >= 0;
                    v[i].x = vx ? fma(xa[i] - KA, sA, -mA) : 0.0;
                    v[i].y = vy ? fma(yb[i] - KB, sB, -mB) : 0.0;
                }
            } else {
                double c[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const double da = xa[i] - KA, db = yb[i] - KB;
                    v[i] = make_double2(da, db);
                    c[0] += da;
                    c[1] = fma(da, da, c[1]);
                    c[2] += db;
                    c[3] = fma(db, db, c[3]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    q[k] = uniform(q[k] + wave_sum_dpp(c[k]));
            }
            sweep_dft<R1>(v);
            const unsigned js = (unsigned)(opaque(t + 256 * ch) & (S - 1));
#pragma unroll
            for (int m = 0; m < Q1; m++)
                *(yat((long long)m * S) + js) = d2v{v[m].x, v[m].y};
#pragma unroll
            for (int m = 0; m < Q1; m++) {
                twiddle_powers<R1>(wb[m], [&](const int k1, const double2 w) __attribute__((always_inline)) {
                    const double2 z = cmul(v[m + brev<R1>(k1) * Q1], w);
                    *(yat((long long)(m + k1 * Q1) * S) + js) = d2v{z.x, z.y};
                });
            }
        }
        if (!PRE) { // the statistics of the single read: nil / NaN, the one factor, or the pair is listed for the kernel that scales first
            block_stats();
            if (normalize) {
                bool zA, nA, zB, nB;
                const double vA = variance(Stat{q[0], q[1]}, iv.invNx, iv.invNxm1, zA, nA), vB = variance(Stat{q[2], q[3]}, iv.invNy, iv.invNym1, zB, nB);
                ps.nil = zA || zB; // xcorr.go:110-127
                ps.nan = !ps.nil && (nA || nB);
                dead = ps.nil || ps.nan;
                const int eA = var_exp(vA), eB = var_exp(vB);
                redo = !dead && (sigma_spread_too_wide(vA, vB) || eA > 400 || eA < -400 || eB > 400 || eB < -400);
                double ya = __builtin_amdgcn_rsq(vA), yb = __builtin_amdgcn_rsq(vB);
                ya = ya * fma(-0.5 * vA * ya, ya, 1.5);
                ya = ya * fma(-0.5 * vA * ya, ya, 1.5);
                yb = yb * fma(-0.5 * vB * yb, yb, 1.5);
                yb = yb * fma(-0.5 * vB * yb, yb, 1.5);
                fac = ya * yb * iv.invnm1 * (1.0 / (2.0 * n)); // xcorr.go:140: 1 / (n (n - 1)), the 1 / n with the 1 / 2 of Im Z^2
            } else {
                const double mA2 = q[1] * iv.invNx, mB2 = q[3] * iv.invNy; // mean squares: only their exponents matter
                ps.nil = false;
                // (a mean square of 0 -- an all-zero series, or squares that underflow -- gives no scale either: two_device.h, pair_scale)
                ps.nan = !__builtin_isfinite(mA2) || !__builtin_isfinite(mB2) || !(mA2 > 0.0) || !(mB2 > 0.0);
                dead = ps.nan;
                const int eA = var_exp(mA2), eB = var_exp(mB2);
                redo = !dead && (sigma_spread_too_wide(mA2, mB2) || eA > 400 || eA < -400 || eB > 400 || eB < -400);
                fac = 1.0 / (2.0 * n);
            }
            if (redo && t == 0) {
                const int slot = atomicAdd(p.ovf_count, 1);
                p.ovf_list[slot] = pair;
            }
        }
        __syncthreads(); // the slice is complete
        // ---------------- rows: Z = FFT(row), FFT(Z^2), in place
#pragma clang loop unroll(disable)
        for (int k1 = 0; k1 < R1; k1++) {
            double2 *const row = Y + k1 * 4096;
            double2 v[16];
            {
                const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const d2v z = __builtin_nontemporal_load((gd2)scalar_ptr_at(row, 256 * i) + tl);
                    v[i] = make_double2(z.x, z.y);
                }
            }
            dft16_nr(v);
            exchange_cross<0, 1, true>(v, xbuf, wave, t);
            gdft16_nr(v, G2Fetch{g2s, t >> 4});
            exchange_local<1>(v, xw, t);
            gdft16_nr_l2(v, G3Fetch{p.g3a, t});
            if (!PRE && normalize && k1 == 0) { // bin 0 = sum dx + i sum dy: both means leave with it
                v[0].x = (t == 0) ? 0.0 : v[0].x;
                v[0].y = (t == 0) ? 0.0 : v[0].y;
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2)
                bf_sq(v[r], v[r + 1]);
            dft16_rn_s234(v);
            exchange_local<0>(v, xw, t);
            gdft16_nr(v, G2Fetch{g2s, t & 15});
            exchange_cross<1, 1>(v, xbuf, wave, t);
            gdft16_nr_l2(v, G3Fetch{p.g3b, t});
            {
                const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int m = 0; m < 16; m++)
                    *((gd2)scalar_ptr_at(row, 256 * m) + tl) = d2v{v[BR16(m)].x, v[BR16(m)].y};
            }
        }
        __syncthreads();
        // ---------------- sweep 2: twiddle, radix R1 over k1; 2 n cc[j + i S] = Im; running argmax (maxAbsIndex, xcorr.go:39-50)
        double ma = 0.0, sa = 0.0, cc0 = 0.0;
        int ia = 0x7fffffff;
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = opaque(t + 256 * ch) & (S - 1);
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const d2v z = __builtin_nontemporal_load(yat((long long)i * S) + (unsigned)j);
                v[i] = make_double2(z.x, z.y);
            }
            {
                double2 wb[Q1];
#pragma unroll
                for (int m = 0; m < Q1; m++)
                    wb[m] = tw_base(m, (unsigned)j);
                fence();
#pragma unroll
                for (int m = 0; m < Q1; m++)
                    twiddle_powers<R1>(wb[m], [&](const int k1, const double2 w) __attribute__((always_inline)) {
                        v[m + k1 * Q1] = cmul(v[m + k1 * Q1], w);
                    });
            }
            sweep_dft<R1>(v);
            double cs = 0.0;
            int ci = 0;
            const int jc = opaque(t + 256 * ch) & (S - 1);
            if (p.cc_out && !dead && !redo) {
                double *const cc = p.cc_out + pair * (long long)n;
#pragma unroll
                for (int i = 0; i < 16; i++)
                    cc[jc + i * S] = v[i % Q1 + brev<R1>(i / Q1) * Q1].y * fac;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) { // i = m + l1 Q1: lag index j + i S
                const double c = v[i % Q1 + brev<R1>(i / Q1) * Q1].y;
                if (i == 0)
                    cc0 = ch == 0 ? c : cc0; // (lane 0 of chunk 0: cc[0], the value reported when nothing is above 0)
                const bool g = fabs(c) > fabs(cs);
                cs = g ? c : cs;
                ci = g ? i : ci;
            }
            {   // merged into the lane's running maximum (chunks are not in index order: ties go to the lower index)
                const int xi = jc + ci * S;
                const double ca = fabs(cs);
                const bool tk = (ca > ma) | ((ca == ma) & (ca > 0.0) & (xi < ia));
                ma = tk ? ca : ma;
                sa = tk ? cs : sa;
                ia = tk ? xi : ia;
            }
        }
        {
            constexpr int RM = 4 * NW;
            const double wa = wave_max(ma);
            if (lane == 0)
                red[RM + wave] = wa;
            if (t == 0)
                red[RM + NW] = cc0;
            __syncthreads();
            double MA = red[RM];
#pragma unroll
            for (int x = 1; x < NW; x++)
                MA = fmax(MA, red[RM + x]);
            int ca = (ma == MA && MA > 0.0) ? ia : 0x7fffffff;
            ca = wave_min_i(ca);
            if (lane == 0)
                redi[wave] = ca;
            __syncthreads();
            int IA = redi[0];
#pragma unroll
            for (int x = 1; x < NW; x++)
                IA = min(IA, redi[x]);
            const bool none = IA == 0x7fffffff;
            const bool owner = (none ? (t == 0) : (ia == IA && ma == MA)) && !redo; // (a listed pair is written by its second pass)
            if (owner) {
                const int idx = none ? 0 : IA;
                double mv = (none ? red[RM + NW] : sa) * fac;
                int lag = idx > n / 2 ? idx - n : idx;
                if (ps.nil) { mv = 0.0; lag = 0; }               // xcorr.go:110-127
                if (ps.nan) { mv = __builtin_nan(""); lag = 0; } // every cc is NaN: maxAbsIndex keeps index 0
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
static hipError_t launch_two_long(const FusedParams &p, int num_cus, hipStream_t stream)
{
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * LONG_WGS_PER_CU);
    if (!p.gscratch || grid > p.gscratch_slices || !p.twl || !p.g2 || !p.g3a || !p.g3b) // one n-element slice per workgroup
        return hipErrorInvalidValue;
    const two::PairInv iv = two::pair_inv(p.Nx, p.N, 1 << LOGN);
    if (p.Nx == (1 << LOGN) && p.N == (1 << LOGN)) {
        if (!p.ovf_list || !p.ovf_count || p.pair_list) // (the caller zeroes *ovf_count in front of this launch)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL((xcorr_two_sided_long<LOGN, false, false>), dim3((unsigned)grid), dim3(256), 0, stream, p, iv);
        FusedParams q = p; // the listed pairs again, statistics first (grid size only: the loop is bounded by the count on the device)
        q.pair_list = p.ovf_list;
        q.pair_count = p.ovf_count;
        const long long g2 = std::min<long long>(grid, (long long)num_cus);
        hipLaunchKernelGGL((xcorr_two_sided_long<LOGN, false, true>), dim3((unsigned)g2), dim3(256), 0, stream, q, iv);
    } else
        hipLaunchKernelGGL((xcorr_two_sided_long<LOGN, true, true>), dim3((unsigned)grid), dim3(256), 0, stream, p, iv);
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
    case 15: return launch_two_sided_real(p, num_cus, stream); // xcorr_real.hip: each series a real transform on the 16384-point machinery
    case 16: return launch_two_long<16>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace muse
