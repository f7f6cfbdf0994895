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
#include "small_device.h"
#include "two_device.h"

namespace muse {

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
