// xcorr_r16_occ4.hip -- high-occupancy tuned kernel for n = 4096.
//
// Mathematics: identical to xcorr_fused_n4096 (xcorr_kernels.hip header; the
// reference path is xCorrWithX, /root/reference/xcorr.go:160-197).
//
// Why this shape (round-1 ablations and phase stamps, tools/ablate +
// profiles/): with the full-size complex exchange buffer (69.6 KB) only two
// workgroups fit a CU; a workgroup's timeline per pair is ~28k cycles of which
// ~8k are VALU work, the rest dependent stalls (HBM, L2, LDS, barriers, and
// fp64 div/sqrt chains), and two waves per SIMD cannot cover that.  Here:
//   * each LDS transpose runs as TWO half rounds through a 34.8 KB buffer
//     (outputs 0..7, then 8..15; waves 0-1 read after the first round, waves
//     2-3 after the second), so 3-4 workgroups share a CU;
//   * the dependent chains are shortened: twiddle / spectrum loads are issued
//     one phase ahead of their use; the z-normalisation needs no division on
//     the critical path (1/N, 1/(N-1) are per-launch constants; sqrt and 1/sigma
//     are evaluated only by the thread that writes the result); the argmax is
//     resolved with wave ballots on the scalar unit and its cross-wave combine
//     is deferred behind the next pair's first barrier (no barrier of its own);
//   * all global pointers are re-materialised as scalars (saddr + shared VGPR
//     offset): no hoisted 64-bit VGPR addresses; loads are unconditional
//     (clamped index + select);
//   * DPP wave reductions; 1/sigma multiplies only the winning value; for
//     N == n the mean is removed from the DC bin after the first FFT instead of
//     from every sample.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "r16_device.h"

namespace muse {

namespace occ4 {

// forward FFT: v[a] = x[t + 256 a] -> v[a] = X[t + 256 a].  With MULXC the
// result is multiplied by xc[t + 256 a] (the batch's conj(X)/n table); those 16
// L2 loads are issued around the last butterflies.  For N == n the DC bin is
// corrected by dc (see caller) before the multiply.
// Without MULXC (the second FFT) the next pair's row loads are issued just before
// the last butterflies: they stay in flight during pass 3 and the argmax, the only
// stretch with no other global access and the registers to spare.
template <int P0, bool MULXC, bool PADDED, bool TIMING, bool F32>
__device__ __forceinline__ void fft4096(double2 (&v)[16], double2 *xbuf, const double2 *tw2s,
                                        const double2 *__restrict__ tw1g, const double2 *__restrict__ xcg,
                                        const double2 dc, const int t, const int wave, PhaseClock<TIMING> &clk,
                                        RawPair &raw, const FusedParams &p, long long next_pair, int pad)
{
    // pass 1: DFT over a, twiddle W_4096^(k1 t) (L2-resident table)
    dft16_twiddle(v, Tw1Fetch{tw1g, t});
    clk.template stamp<P0>();
    exchange<false>(v, xbuf, wave, t);
    clk.template stamp<P0 + 1>();
    // pass 2: DFT over b (k1 = hi, c = lo), twiddle W_256^(k2 c) from the LDS table
    dft16_twiddle(v, Tw2Fetch{tw2s, t & 15});
    clk.template stamp<P0 + 2>();
    exchange<true>(v, xbuf, wave, t);
    clk.template stamp<P0 + 3>();
    // pass 3: DFT over c (k1 = lo, k2 = hi): f = t + 256 k3
    if (MULXC) {
        double2 xa[8], xb[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
            xa[j] = ldg2(scalar_ptr_at(xcg, 256 * ((j + 1) & ~1)), t - 256 * (j & 1));
        fence();
        dft16(v);
        fence();
#pragma unroll
        for (int j = 0; j < 8; j++)
            xb[j] = ldg2(scalar_ptr_at(xcg, 256 * ((9 + j) & ~1)), t - 256 * (j & 1));
        double2 w[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            w[k] = v[P16(k)];
        if (t == 0) { // N == n: FFT(d - m)[0] = FFT(d)[0] - n m  (dc = 0 otherwise)
            w[0].x -= dc.x;
            w[0].y -= dc.y;
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            v[j] = cmul(w[j], xa[j]);
        fence();
#pragma unroll
        for (int j = 0; j < 8; j++)
            v[8 + j] = cmul(w[8 + j], xb[j]);
    } else {
        fence();
        issue_row_loads<PADDED, F32>(raw, p, next_pair, t, pad);
        fence();
        dft16(v);
        double2 w[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            w[k] = v[P16(k)];
#pragma unroll
        for (int k = 0; k < 16; k++)
            v[k] = w[k];
    }
    clk.template stamp<P0 + 4>();
}

// cross-wave combine of one series' argmax + result store (threads 0 / 1 only):
// r[6*w + {0,1,2}] = wave w's {max |cc|, signed value, first index}
__device__ __forceinline__ void finalize(const double *r, const Stat &st, double invN, double invNm1,
                                         double *mv_out, int *lag_out)
{
    double m[4], s[4], ix[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        m[w] = r[6 * w];
        s[w] = r[6 * w + 1];
        ix[w] = r[6 * w + 2];
    }
    double best = m[0], bsv = s[0], bidx = ix[0];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        if (m[w] > best || (m[w] == best && ix[w] < bidx)) {
            best = m[w];
            bsv = s[w];
            bidx = ix[w];
        }
    }
    bool zero, nan;
    const double var = variance(st, invN, invNm1, zero, nan);
    const int idx = (best > 0.0) ? (int)bidx : 0;           // nothing above 0: index 0, mv = cc[0]
    // 1/sigma for the winner only: v_rsq_f64 seed + two Newton steps (full double accuracy)
    // instead of the sqrt + division chains, which sat on wave 0's way to the next barrier
    double y = __builtin_amdgcn_rsq(var);
    y = y * fma(-0.5 * var * y, y, 1.5);
    y = y * fma(-0.5 * var * y, y, 1.5);
    double mv = ((best > 0.0) ? bsv : s[0]) * y;
    int lag = idx > 2048 ? idx - 4096 : idx;
    if (zero) { mv = 0.0; lag = 0; }                        // xcorr.go:166-167
    if (nan) { mv = __builtin_nan(""); lag = 0; }           // NaN sigma: every cc is NaN
    *mv_out = mv;
    *lag_out = lag;
}

} // namespace occ4

template <bool PADDED, int WPS, bool TIMING = false, bool F32 = false>
__global__ __launch_bounds__(OCC_THREADS, WPS) void xcorr_fused_n4096_occ4(const FusedParams p)
{
    using namespace occ4;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 tw2s[256];
    __shared__ double red[16 + 2 * 24]; // [0,16): z-norm partials; then 2 parities x 4 waves x 2 series x 3
    // the previous pair's state per parity and series: {sum d, sum d^2, row (-1: none)} -- in LDS, not in registers carried across
    // the transforms by threads 0 / 1 (those six registers were parked in scratch: stored and reloaded every pair)
    __shared__ double pst[2][2][3];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform by construction
    const int N = p.N;
    const int pad = 4096 - N;
    const double invN = p.invN, invNm1 = p.invNm1; // (the launcher's quotients: kernel arguments in scalar registers)

    tw2s[t] = p.tw2[t];
    PhaseClock<TIMING> clk;

    // (the previous pair's cross-wave argmax combine runs behind this pair's first barrier: threads 0 / 1, from pst)
    if (t < 4)
        pst[t >> 1][t & 1][2] = -1.0;
    __syncthreads();
    clk.start();
    int parity = 0;

    // optional indirection: process pair_list[0 .. *pair_count) (overflow pairs of the
    // fp32 screening kernel); otherwise pairs 0 .. npairs
    // A DENSE list (more than an eighth of the group's pairs listed: a group of mixed-unit series) is not followed: the
    // kernel redoes EVERY pair, so that all results of such a group come from this kernel in every pass -- the first one, which
    // found the list, and the later ones, which the host sends here directly (capi_batch.hip): Run(); Run() is bit-identical.
    long long total = p.pair_count ? (long long)*p.pair_count : p.npairs;
    const long long *__restrict__ plist = p.pair_list;
    if (p.pair_count && p.dense_total > 0 && total * 8 > p.dense_total) {
        total = p.dense_total;
        plist = nullptr;
    }
    RawPair raw;
    {
        long long first = 0;
        if (blockIdx.x < total)
            first = plist ? plist[blockIdx.x] : (long long)blockIdx.x;
        issue_row_loads<PADDED, F32>(raw, p, first, t, pad);
    }

    for (long long it = blockIdx.x; it < total; it += gridDim.x) {
        const long long pair = plist ? plist[it] : it;
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        // ---- consume the prefetched rows (element t + 256 i of the zero-padded rows)
        double2 v[16];
        const double KA = raw.ka, KB = raw.kb;
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = make_double2(raw.a[i], raw.b[i]);
        if (TIMING) // charge the load wait to phase 0
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        clk.template stamp<0>();
        // ---- d = x - K, K = first sample (pads -> 0); shifted one-pass statistics:
        // mean = K + S1/N, (N-1) var = S2 - S1^2/N.  K is a sample of the series, so
        // (mean-K)^2 <= (N-1) var and the cancellation is bounded by ~N ulp.
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            // (PADDED: a pad position was loaded from the clamped index 0 = the first sample K: d = 0 already)
            const double da = v[i].x - KA, db = v[i].y - KB;
            v[i] = make_double2(da, db);
            q[0] += da;
            q[1] = fma(da, da, q[1]);
            q[2] += db;
            q[3] = fma(db, db, q[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            q[k] = wave_sum_dpp(q[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                red[wave * 4 + k] = q[k];
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) // block totals: identical in every lane -> SGPRs
            q[k] = uniform((red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]));
        // the previous pair's argmax triples are visible now: finish that pair
        if (t < 2) {
            const double *const ps = pst[parity ^ 1][t];
            if (ps[2] >= 0.0) {
                const long long prev_row = (long long)ps[2];
                finalize(red + 16 + 24 * (parity ^ 1) + 3 * t, Stat{ps[0], ps[1]}, invN, invNm1, p.mv + prev_row, p.lag + prev_row);
            }
        }
        Stat stA{q[0], q[1]}, stB{q[2], q[3]};
        bool zeroA, nanA, zeroB, nanB;
        const double varA0 = variance(stA, invN, invNm1, zeroA, nanA);
        const double varB0 = variance(stB, invN, invNm1, zeroB, nanB);
        double mA = uniform(q[0] * invN), mB = uniform(q[2] * invN); // mean of d
        if (PADDED) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool valid = i >= 8 || t + 256 * i - pad >= 0;
                v[i].x = valid ? v[i].x - mA : 0.0;
                v[i].y = valid ? v[i].y - mB : 0.0;
            }
        }
        // a sigma == 0 / NaN series (or the missing partner of an odd last row) must
        // contribute exact zeros to the shared complex transform
        const bool deadA = zeroA || nanA, deadB = zeroB || nanB || !hasB;
        if (deadA || deadB) { // block-uniform, rare
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v[i].x = deadA ? 0.0 : v[i].x;
                v[i].y = deadB ? 0.0 : v[i].y;
            }
        }
        if (hasB && !nanA && !nanB && sigma_spread_too_wide(varA0, varB0)) { // block-uniform, rare
            // sigmas more than 2^16 apart: bring both series to O(1) with exact powers of two, or the
            // shared transform's rounding drowns the smaller one; the statistics scale along exactly
            const double sA = pow2_inv_sigma(varA0), sB = pow2_inv_sigma(varB0);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v[i].x *= sA;
                v[i].y *= sB;
            }
            stA.s1 *= sA;
            stA.s2 *= sA * sA;
            stB.s1 *= sB;
            stB.s2 *= sB * sB;
            mA *= sA;
            mB *= sB;
        }
        if (t == 0) { // this pair's state for the combine behind the NEXT pair's first barrier (the buffer's last reader was two pairs ago)
            pst[parity][0][0] = stA.s1;
            pst[parity][0][1] = stA.s2;
            pst[parity][0][2] = (double)rA;
            pst[parity][1][0] = stB.s1;
            pst[parity][1][1] = stB.s2;
            pst[parity][1][2] = hasB ? (double)rB : -1.0;
        }
        const double2 dc = PADDED ? make_double2(0.0, 0.0)
                                  : make_double2(uniform(deadA ? 0.0 : 4096.0 * mA), uniform(deadB ? 0.0 : 4096.0 * mB));
        clk.template stamp<1>();
        // ---- Z = FFT(yA + i yB);  V[f] = Z[f] * conj(X[f]) / n
        long long nxt = 0; // last iteration: an unconditional dummy prefetch of pair 0, whose 64 KB every
                           // workgroup re-reads (L2-resident) -- re-reading the own pair cost 2.4 % HBM traffic
        if (it + gridDim.x < total)
            nxt = plist ? plist[it + gridDim.x] : it + gridDim.x;
        fft4096<2, true, PADDED, TIMING, F32>(v, xbuf, tw2s, p.tw1, p.xc, dc, t, wave, clk, raw, p, nxt, pad);
        clk.template stamp<7>();
        // ---- ccA + i ccB = FFT(V)   (unscaled by 1/sigma)
        fft4096<8, false, PADDED, TIMING, F32>(v, xbuf, tw2s, p.tw1, p.xc, dc, t, wave, clk, raw, p, nxt, pad);

        // ---- maxAbsIndex (xcorr.go:39-50), index = t + 256 k.  Per thread only
        // max |cc| is tracked; the wave's first index attaining the wave maximum and
        // its sign come from ballots (scalar unit).  Lowest k first, then lowest lane
        // == lowest index, because t < 256.
        double ma = 0.0, mb = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            ma = fmax(ma, fabs(v[k].x));
            mb = fmax(mb, fabs(v[k].y));
        }
        const double wa = wave_max_dpp(ma), wb = wave_max_dpp(mb);
        // first index attaining the wave maximum: all 16 ballots are issued back to back
        // (VALU -> SGPR), then a branch-free scalar select chain picks the lowest k with a
        // hit; lowest k first, then lowest lane == lowest index because t < 256.  The sign
        // comes from the selected value's high word at that lane.
        int widxA = 0x7fffffff, widxB = 0x7fffffff;
        double svA = 0.0, svB = 0.0;
        {
            unsigned long long selA = 0ull, selB = 0ull;
            int kA = 0, kB = 0, hiA = 0, hiB = 0;
#pragma unroll
            for (int k = 15; k >= 0; k--) { // descending: the lowest k is selected last
                const unsigned long long mA_ = __ballot(fabs(v[k].x) == wa);
                const unsigned long long mB_ = __ballot(fabs(v[k].y) == wb);
                const bool hA = mA_ != 0ull, hB = mB_ != 0ull; // wave-uniform
                selA = hA ? mA_ : selA;
                kA = hA ? k : kA;
                hiA = hA ? __double2hiint(v[k].x) : hiA;
                selB = hB ? mB_ : selB;
                kB = hB ? k : kB;
                hiB = hB ? __double2hiint(v[k].y) : hiB;
            }
            if (wa > 0.0 && selA != 0ull) {
                const int l = __ffsll((long long)selA) - 1;
                widxA = wave * 64 + l + 256 * kA;
                svA = (__builtin_amdgcn_readlane(hiA, l) < 0) ? -wa : wa;
            }
            if (wb > 0.0 && selB != 0ull) {
                const int l = __ffsll((long long)selB) - 1;
                widxB = wave * 64 + l + 256 * kB;
                svB = (__builtin_amdgcn_readlane(hiB, l) < 0) ? -wb : wb;
            }
        }
        if (lane == 0) { // {max |cc|, signed value (cc[0] when nothing is above 0), index}
            double *ra_ = red + 16 + 24 * parity + 6 * wave;
            ra_[0] = widxA == 0x7fffffff ? 0.0 : wa;
            ra_[1] = widxA == 0x7fffffff ? v[0].x : svA; // wave 0 lane 0 holds cc[0]
            ra_[2] = (double)widxA;
            ra_[3] = widxB == 0x7fffffff ? 0.0 : wb;
            ra_[4] = widxB == 0x7fffffff ? v[0].y : svB;
            ra_[5] = (double)widxB;
        }
        // no barrier here: the triples are combined behind the next pair's first one
        parity ^= 1;
        clk.template stamp<13>();
    }
    lds_barrier();
    if (t < 2) {
        const double *const ps = pst[parity ^ 1][t];
        if (ps[2] >= 0.0) {
            const long long prev_row = (long long)ps[2];
            finalize(red + 16 + 24 * (parity ^ 1) + 3 * t, Stat{ps[0], ps[1]}, invN, invNm1, p.mv + prev_row, p.lag + prev_row);
        }
    }
    if (TIMING && p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < NPHASE; i++)
            p.dbg[((long long)blockIdx.x * 4 + wave) * NPHASE + i] = clk.acc[i];
    }
}

hipError_t launch_fused_occ4(const FusedParams &p_in, int num_cus, hipStream_t stream)
{
    const FusedParams p = with_reciprocals(p_in);
    long long grid = p.npairs;
    // 16x the resident set (each workgroup still loops over ~40 pairs): workgroups that
    // start as others retire keep the CUs' phases decorrelated and balance CU speed
    // differences; measured 12.6 ms (1x) -> 11.5 ms (16x) per 1 M series, flat beyond.
    constexpr int mult = 16;
    const long long cap = (long long)num_cus * 3 * mult;
    if (grid > cap)
        grid = cap;
    const dim3 g((unsigned)grid), b(OCC_THREADS);
    if (p.rows32) {
        if (p.N < 4096)
            hipLaunchKernelGGL((xcorr_fused_n4096_occ4<true, 3, false, true>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_fused_n4096_occ4<false, 3, false, true>), g, b, 0, stream, p);
    } else if (p.N < 4096)
        hipLaunchKernelGGL((xcorr_fused_n4096_occ4<true, 3>), g, b, 0, stream, p);
    else
        hipLaunchKernelGGL((xcorr_fused_n4096_occ4<false, 3>), g, b, 0, stream, p);
    return hipGetLastError();
}

} // namespace muse
