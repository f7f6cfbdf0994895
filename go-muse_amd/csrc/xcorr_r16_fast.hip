// xcorr_r16_fast.hip -- the default n = 4096 kernels (2048 < N <= 4096): nine workgroup
// barriers per pair of series (xcorr_r16_occ4.hip: nineteen), and the many-references
// variant that transforms each pair once for R references (second half of the file).
//
// Mathematics: identical to xcorr_fused_n4096 (xcorr_kernels.hip header; the
// reference path is xCorrWithX, /root/reference/xcorr.go:160-197).
//
// What changed against xcorr_r16_occ4.hip and why (phase stamps in profiles/:
// the LDS transposes with their barriers and the two reductions were 56 % of a
// workgroup's timeline, the butterflies 31 %):
//   * Thread numbering.  The first transform keeps (b, c) -> (k1, c) for its first
//     transpose (workgroup-wide, dictated by the coalesced row loads), but its second
//     transpose hands (k1, c) -> (k1, k2): the sixteen lanes that share k1 sit in one
//     wave, so it runs inside each wave's private quarter of the buffer with no
//     workgroup barrier.  The second transform starts from that numbering
//     (c' = k1 in the high half of the lane id, b' = k2 in the low half): its FIRST
//     transpose is wave-local for the same reason and only its second is
//     workgroup-wide.  The twiddle and spectrum tables are stored in lane order
//     (FusedParams::tw1p, xcp), so every table load stays coalesced.
//   * No statistics barrier.  For N == n the mean is never needed before the
//     transform: the DC bin of the centred series is exactly 0, so bin 0 is zeroed
//     after the first transform and sum(d) is read off it; sum(d^2) partials go to
//     LDS and are only read by the two lanes that write the results, behind the next
//     pair's first barrier, together with the argmax partials.
//   * A series with NaN/Inf samples cannot be isolated from its pair partner once
//     the statistics are deferred: such pairs are appended to FusedParams::ovf_list
//     and redone by the occ4 kernel (which zeroes the dead series before the
//     transform) in a second launch bounded by the on-device count.
//     The same list takes pairs whose sigmas are more than 2^16 apart (fft_device.h,
//     sigma_spread_too_wide): kernel 7 rescales both series before the shared transform.
//   * The state of the previous pair lives in LDS, not in registers that only two
//     lanes use.
//   * Pairs are handed out by an atomic counter to a grid of resident workgroups (DYN).
//   * N < 4096 (PADDED): see the comment at the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "r16_device.h"

namespace muse {

namespace fast {

using namespace occ4;

constexpr int XW = 544; // double2 per wave-private quarter of the 8 x 272 buffer (8 rows x 68)

// Workgroup-wide transpose in two half rounds (as occ4::exchange; layouts in
// double2 units, bank analysis there):
//   MODE 0 (first transform, transpose 1):
//      writer (b = hi, c = lo) output k1 -> 272*(k1&7) + t
//      reader (k1 = hi, c = lo) input b  <- 272*(hi&7) + 16*b + lo
//   MODE 1 (second transform, transpose 2):
//      writer (c' = hi, m1 = lo) output m2 -> 272*(m2&7) + 17*lo + hi
//      reader (m1 = lo, m2 = hi) input c'  <- 272*(hi&7) + 17*lo + c'
// Round 0 moves outputs 0..7 (read by waves 0-1), round 1 outputs 8..15 (waves 2-3).
// (Letting waves 2-3 read first in the second transpose of a pair, so that the late-reader role alternates,
// was measured: no difference.  So was s_setprio around the transposes.)
template <int MODE>
__device__ __forceinline__ void exchange_cross(double2 (&v)[16], double2 *xbuf, const int wave, const int t)
{
    constexpr int K0 = 0, K1 = 8;
    const int hi = t >> 4, lo = t & 15;
    const int wbase = MODE ? 17 * lo + hi : t;
    const int rbase = 272 * (hi & 7) + (MODE ? 17 * lo : lo);
    const bool early = wave < 2;
    if (MUSE_ABLATE & 1) { // no LDS: keep a register permutation so the data flow stays
        double2 w[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            w[e] = v[P16(e)];
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = w[e];
        return;
    }
    lds_barrier(); // buffer free: every wave is done with its previous (wave-local or shared) use
#pragma unroll
    for (int k = 0; k < 8; k++)
        xbuf[272 * k + wbase] = v[P16(K0 + k)];
    lds_barrier();
    if (early) {
        double2 w[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            w[e] = xbuf[rbase + (MODE ? e : 16 * e)];
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++)
            xbuf[272 * k + wbase] = v[P16(K1 + k)];
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = w[e];
    } else {
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++)
            xbuf[272 * k + wbase] = v[P16(K1 + k)];
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = xbuf[rbase + (MODE ? e : 16 * e)];
    }
}

// Wave-local transpose among the sixteen lanes that share hi: lane (hi, lo) holds
// outputs k = 0..15 (at v[P16(k)]) and receives input e = 0..15 of the lane-row's
// output lo, i.e. value lo of lane (hi, e).  Two half rounds through the wave's
// private quarter xw[0..544):
//   writer: output k (round k>>3) -> 68*(k&7) + 17*hl + lo        (hl = hi & 3)
//   reader: lane (hl, lo), lo>>3 == round, input e <- 68*(lo&7) + 17*hl + e
// ds_write_b128 groups are 8 contiguous lanes (fixed hl, consecutive lo): 8 consecutive
// 16-B slots.  ds_read_b128 groups hold, per round, 4 active lanes of one lane-row and 4
// of the next: slots 4*(lo&7) + hl + e (mod 16) are distinct across them.
// The half-lane reads are EXEC-masked inside one asm block: written as divergent C++
// branches the receiving registers become a scratch array (548 B/lane).  The block ends
// with s_waitcnt lgkmcnt(0), so its outputs are valid when it returns.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) char *lds_ptr;
#define MUSE_LDS_ADDR(p) ((unsigned)(unsigned long long)(lds_ptr)(p))
#else
#define MUSE_LDS_ADDR(p) 0u
#endif
__device__ __forceinline__ void exchange_local(double2 (&v)[16], double2 *xw, const int hl, const int lo)
{
    if (MUSE_ABLATE & 1) {
        double2 w[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            w[e] = v[P16(e)];
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = w[e];
        return;
    }
    const int wbase = 17 * hl + lo;
    const int rbase = 68 * (lo & 7) + 17 * hl;
#pragma unroll
    for (int k = 0; k < 8; k++)
        xw[68 * k + wbase] = v[P16(k)];
    const unsigned waddr = MUSE_LDS_ADDR(xw + wbase), raddr = MUSE_LDS_ADDR(xw + rbase);
    const unsigned long long first = __ballot(lo < 8); // lanes that read in round 0
    d2v w[16], d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        d[k].x = v[P16(8 + k)].x;
        d[k].y = v[P16(8 + k)].y;
    }
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 exec, %[sv], %[m]\n\t"
                 "ds_read_b128 %[w0], %[ra]\n\t"
                 "ds_read_b128 %[w1], %[ra] offset:16\n\t"
                 "ds_read_b128 %[w2], %[ra] offset:32\n\t"
                 "ds_read_b128 %[w3], %[ra] offset:48\n\t"
                 "ds_read_b128 %[w4], %[ra] offset:64\n\t"
                 "ds_read_b128 %[w5], %[ra] offset:80\n\t"
                 "ds_read_b128 %[w6], %[ra] offset:96\n\t"
                 "ds_read_b128 %[w7], %[ra] offset:112\n\t"
                 "ds_read_b128 %[w8], %[ra] offset:128\n\t"
                 "ds_read_b128 %[w9], %[ra] offset:144\n\t"
                 "ds_read_b128 %[w10], %[ra] offset:160\n\t"
                 "ds_read_b128 %[w11], %[ra] offset:176\n\t"
                 "ds_read_b128 %[w12], %[ra] offset:192\n\t"
                 "ds_read_b128 %[w13], %[ra] offset:208\n\t"
                 "ds_read_b128 %[w14], %[ra] offset:224\n\t"
                 "ds_read_b128 %[w15], %[ra] offset:240\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "ds_write_b128 %[wa], %[d0]\n\t"
                 "ds_write_b128 %[wa], %[d1] offset:1088\n\t"
                 "ds_write_b128 %[wa], %[d2] offset:2176\n\t"
                 "ds_write_b128 %[wa], %[d3] offset:3264\n\t"
                 "ds_write_b128 %[wa], %[d4] offset:4352\n\t"
                 "ds_write_b128 %[wa], %[d5] offset:5440\n\t"
                 "ds_write_b128 %[wa], %[d6] offset:6528\n\t"
                 "ds_write_b128 %[wa], %[d7] offset:7616\n\t"
                 "s_andn2_b64 exec, %[sv], %[m]\n\t"
                 "ds_read_b128 %[w0], %[ra]\n\t"
                 "ds_read_b128 %[w1], %[ra] offset:16\n\t"
                 "ds_read_b128 %[w2], %[ra] offset:32\n\t"
                 "ds_read_b128 %[w3], %[ra] offset:48\n\t"
                 "ds_read_b128 %[w4], %[ra] offset:64\n\t"
                 "ds_read_b128 %[w5], %[ra] offset:80\n\t"
                 "ds_read_b128 %[w6], %[ra] offset:96\n\t"
                 "ds_read_b128 %[w7], %[ra] offset:112\n\t"
                 "ds_read_b128 %[w8], %[ra] offset:128\n\t"
                 "ds_read_b128 %[w9], %[ra] offset:144\n\t"
                 "ds_read_b128 %[w10], %[ra] offset:160\n\t"
                 "ds_read_b128 %[w11], %[ra] offset:176\n\t"
                 "ds_read_b128 %[w12], %[ra] offset:192\n\t"
                 "ds_read_b128 %[w13], %[ra] offset:208\n\t"
                 "ds_read_b128 %[w14], %[ra] offset:224\n\t"
                 "ds_read_b128 %[w15], %[ra] offset:240\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [w0] "=&v"(w[0]), [w1] "=&v"(w[1]), [w2] "=&v"(w[2]), [w3] "=&v"(w[3]), [w4] "=&v"(w[4]),
                   [w5] "=&v"(w[5]), [w6] "=&v"(w[6]), [w7] "=&v"(w[7]), [w8] "=&v"(w[8]), [w9] "=&v"(w[9]),
                   [w10] "=&v"(w[10]), [w11] "=&v"(w[11]), [w12] "=&v"(w[12]), [w13] "=&v"(w[13]),
                   [w14] "=&v"(w[14]), [w15] "=&v"(w[15]), [sv] "=&s"(sv)
                 : [ra] "v"(raddr), [wa] "v"(waddr), [m] "s"(first), [d0] "v"(d[0]), [d1] "v"(d[1]), [d2] "v"(d[2]),
                   [d3] "v"(d[3]), [d4] "v"(d[4]), [d5] "v"(d[5]), [d6] "v"(d[6]), [d7] "v"(d[7])
                 : "memory", "scc");
#pragma unroll
    for (int e = 0; e < 16; e++)
        v[e] = make_double2(w[e].x, w[e].y);
}

// Twiddle passes.  With 168 registers (WPS 3) the first eight factors of a pass are requested
// one phase EARLY -- before the transpose (or the row consumption) that precedes the pass -- and
// the other seven right before the butterflies, so neither the L2 nor the LDS latency sits on
// the wave's dependent chain (phase stamps: a twiddled pass cost 1.8-2.5k ticks against 0.95k
// for the bare butterflies).  With 128 registers (WPS 4) there is no room to carry factors
// across a transpose: four batches of four, two in flight.
template <int WPS, typename F>
__device__ __forceinline__ void tw_early(double2 (&ta)[8], F fetch)
{
    if (WPS < 4) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            ta[j] = fetch(1 + j);
        fence();
    }
}
template <int WPS, typename F>
__device__ __forceinline__ void twiddle_pass(double2 (&v)[16], const double2 (&ta)[8], F fetch)
{
    if (WPS >= 4) {
        dft16_twiddle_small(v, fetch);
    } else {
        double2 tb[7];
#pragma unroll
        for (int j = 0; j < 7; j++)
            tb[j] = fetch(9 + j);
        fence();
        dft16(v);
        fence();
#pragma unroll
        for (int j = 0; j < 8; j++)
            v[P16(1 + j)] = cmul(v[P16(1 + j)], ta[j]);
        fence();
#pragma unroll
        for (int j = 0; j < 7; j++)
            v[P16(9 + j)] = cmul(v[P16(9 + j)], tb[j]);
    }
}

// LDS record of one pair (one per parity): what the two result-writing lanes need
// after the pair's transforms are over.
//   [0, 24)  argmax partials: 6*wave + 3*series + {max |cc|, signed value, index}
//   [24, 32) sum d^2 partials: 24 + 2*wave + series
//   [32, 34) sum d per series (from the DC bin)
//   [34]     first row of the pair as a double (exact below 2^53); < 0: nothing to write
//   [35]     1.0 when the pair has a second row
constexpr int REC = 36;

// cross-wave argmax combine + variance + store (lanes 0 / 1, one series each);
// returns true when the series' statistics are NaN/Inf (the pair must be redone)
__device__ __forceinline__ bool finalize(const double *r, const int series, const double invN, const double invNm1,
                                         double *mv_out, int *lag_out)
{
    double m[4], s[4], ix[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        m[w] = r[6 * w + 3 * series];
        s[w] = r[6 * w + 3 * series + 1];
        ix[w] = r[6 * w + 3 * series + 2];
    }
    const double s2 = (r[24 + series] + r[26 + series]) + (r[28 + series] + r[30 + series]);
    const Stat st{r[32 + series], s2};
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
    const int idx = (best > 0.0) ? (int)bidx : 0; // nothing above 0: index 0, mv = cc[0]
    double y = __builtin_amdgcn_rsq(var);
    y = y * fma(-0.5 * var * y, y, 1.5);
    y = y * fma(-0.5 * var * y, y, 1.5);
    double mv = ((best > 0.0) ? bsv : s[0]) * y;
    int lag = idx > 2048 ? idx - 4096 : idx;
    if (zero) { mv = 0.0; lag = 0; }              // xcorr.go:166-167
    if (nan) { mv = __builtin_nan(""); lag = 0; } // placeholder: the pair is redone
    *mv_out = mv;
    *lag_out = lag;
    bool redo = nan;
    if (series == 0 && r[35] != 0.0) { // the pair's other series: sigmas too far apart for one shared transform?
        const double s2b = (r[25] + r[27]) + (r[29] + r[31]);
        const Stat sb{r[33], s2b};
        bool zb, nb;
        const double varb = variance(sb, invN, invNm1, zb, nb);
        redo = redo || (!nb && sigma_spread_too_wide(var, varb));
    }
    return redo;
}

} // namespace fast

// WPS = waves per SIMD the register budget is set for: 3 (168 VGPRs: the next pair's rows are
// requested before the last butterflies) or 4 (128 VGPRs: no room for the 64 prefetch registers
// next to the butterflies, the rows are requested after the argmax and their latency is covered
// by the other three workgroups on the CU only).
// DYN: pairs are handed out by an atomic counter (FusedParams::work_counter, zeroed before the
// launch) to a grid of resident workgroups only, instead of a static stride over an oversubscribed
// grid: no tail while the slowest CUs finish their fixed share.
// PADDED (2048 < N < 4096, leading zero pad): the mean cannot be dropped with the DC bin -- the pad stays
// 0, only the valid samples are centred.  By linearity cc(d - m 1_valid) = cc(d) - m c1 with
// c1 = the batch's correlation of the valid-sample indicator with the reference (FusedParams::c1,
// built once per batch): the transforms run on d, sum d is read off the DC bin as before and the 16
// values a lane ends with are corrected by -m c1[index] before the argmax.
template <int WPS, bool TIMING = false, bool DYN = false, bool PADDED = false>
__global__ __launch_bounds__(OCC_THREADS, WPS) void xcorr_fused_n4096_fast(const FusedParams p)
{
    using namespace occ4;
    using namespace fast;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 tw2s[256];
    __shared__ double red[2 * REC];
    __shared__ int next_s[2];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform by construction
    const int hi = t >> 4, lo = t & 15;
    double2 *const xw = xbuf + XW * wave;
    const int pad = PADDED ? 4096 - p.N : 0;
    const double invN = PADDED ? 1.0 / (double)p.N : 1.0 / 4096.0, invNm1 = PADDED ? 1.0 / (double)(p.N - 1) : 1.0 / 4095.0;

    tw2s[t] = p.tw2[t];
    if (t < 2)
        red[REC * (t) + 34] = -1.0; // no previous pair yet (either parity)
    __syncthreads();
    PhaseClock<TIMING> clk;
    clk.start();

    int parity = 0;
    const long long total = p.npairs;
    RawPair raw;
    issue_row_loads<PADDED>(raw, p, blockIdx.x < total ? (long long)blockIdx.x : 0ll, t, pad);

    long long nextpair = 0;
    for (long long pair = blockIdx.x; pair < total; pair = nextpair) {
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        double *const rec = red + REC * parity;
        const double *const prec = red + REC * (parity ^ 1);
        if (DYN) { // the pair after this one: claimed now, read behind this pair's barriers
            if (t == 0)
                next_s[parity] = (int)gridDim.x + atomicAdd(p.work_counter, 1);
        } else {
            nextpair = pair + gridDim.x;
        }
        // ---- consume the prefetched rows: d = x - K (K = the row's first sample) bounds
        // the cancellation in sum d^2 - (sum d)^2 / N and keeps a large level out of the
        // transform's rounding; sum d itself is read off the DC bin below
        double2 v[16], ta[8];
        tw_early<WPS>(ta, Tw1Fetch{p.tw1, t}); // pass 1's first factors (L2) while the rows are consumed
        {
            const double KA = raw.ka, KB = raw.kb;
            double qa = 0.0, qb = 0.0;
            if (TIMING)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            clk.template stamp<0>();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                // (PADDED: a pad position was loaded from the clamped index 0, i.e. it holds the row's first
                // sample K itself, so d = K - K = 0 without any masking)
                const double da = raw.a[i] - KA, db = raw.b[i] - KB;
                v[i] = make_double2(da, db);
                qa = fma(da, da, qa);
                qb = fma(db, db, qb);
            }
            qa = wave_sum_dpp(qa);
            qb = wave_sum_dpp(qb);
            if (lane == 0) {
                rec[24 + 2 * wave] = qa;
                rec[24 + 2 * wave + 1] = qb;
            }
            if (t == 0) {
                rec[34] = (double)rA;
                rec[35] = hasB ? 1.0 : 0.0;
            }
        }
        clk.template stamp<1>();
        // ================= Z = FFT(dA + i dB) =================
        // pass 1: DFT over a, twiddle W_4096^(k1 t)
        twiddle_pass<WPS>(v, ta, Tw1Fetch{p.tw1, t});
        clk.template stamp<2>();
        tw_early<WPS>(ta, Tw2Fetch{tw2s, lo});
        exchange_cross<0>(v, xbuf, wave, t);
        // the previous pair's record is complete and visible: write its results
        if (t < 2 && prec[34] >= 0.0 && (t == 0 || prec[35] != 0.0)) {
            const long long row = (long long)prec[34] + t;
            if (finalize(prec, t, invN, invNm1, p.mv + row, p.lag + row)) {
                const int slot = atomicAdd(p.ovf_count, 1);
                p.ovf_list[slot] = row >> 1;
            }
        }
        clk.template stamp<3>();
        // pass 2: DFT over b (k1 = hi, c = lo), twiddle W_256^(k2 c)
        twiddle_pass<WPS>(v, ta, Tw2Fetch{tw2s, lo});
        clk.template stamp<4>();
        const auto xcl = [&](int j) { return ldg2(scalar_ptr_at(p.xcp, 256 * ((j + 1) & ~1)), t - 256 * (j & 1)); };
        constexpr int XB = WPS >= 4 ? 4 : 8; // spectrum factors per batch (two batches in flight)
        double2 xa[XB], xb[XB];
        if (WPS < 4) { // the first spectrum factors (L2) travel during the transpose
#pragma unroll
            for (int j = 0; j < XB; j++)
                xa[j] = xcl(j);
            fence();
        }
        lds_barrier(); // waves 2-3 may still be reading this wave's quarter (round 1 above)
        exchange_local(v, xw, hi & 3, lo);
        clk.template stamp<5>();
        // pass 3: DFT over c (k1 = hi, k2 = lo): f = hi + 16 lo + 256 k3; V = Z * conj(X)/n
        double s1a, s1b;
        {
            if (WPS >= 4) {
#pragma unroll
                for (int j = 0; j < XB; j++)
                    xa[j] = xcl(j);
            } else {
#pragma unroll
                for (int j = 0; j < XB; j++)
                    xb[j] = xcl(XB + j);
            }
            fence();
            dft16(v);
            fence();
            if (WPS >= 4) {
#pragma unroll
                for (int j = 0; j < XB; j++)
                    xb[j] = xcl(XB + j);
            }
            double2 w[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                w[k] = v[P16(k)];
            // bin 0 (lane 0 of wave 0) = (sum dA, sum dB): kept in SGPRs until the record is
            // written; the centred series' DC bin is exactly 0.  Branch-free on purpose: a
            // divergent block here splits the schedule and the table loads get spilled.
            s1a = readlane_f64(w[0].x, 0);
            s1b = readlane_f64(w[0].y, 0);
            if (!PADDED) {
                w[0].x = (t == 0) ? 0.0 : w[0].x;
                w[0].y = (t == 0) ? 0.0 : w[0].y;
            }
#pragma unroll
            for (int b = 0; b < 16 / XB; b += 2) {
#pragma unroll
                for (int j = 0; j < XB; j++)
                    v[b * XB + j] = cmul(w[b * XB + j], xa[j]);
                fence();
                if (b == 0)
                    tw_early<WPS>(ta, Tw1Fetch{p.tw1p, t}); // second transform, pass 1 (L2)
                if ((b + 2) * XB < 16) {
#pragma unroll
                    for (int j = 0; j < XB; j++)
                        xa[j] = xcl((b + 2) * XB + j);
                }
#pragma unroll
                for (int j = 0; j < XB; j++)
                    v[(b + 1) * XB + j] = cmul(w[(b + 1) * XB + j], xb[j]);
                fence();
                if ((b + 3) * XB < 16) {
#pragma unroll
                    for (int j = 0; j < XB; j++)
                        xb[j] = xcl((b + 3) * XB + j);
                }
            }
        }
        clk.template stamp<6>();
        // ================= ccA + i ccB = FFT(V) (unscaled by 1/sigma) =================
        // element f = 256 a' + 16 b' + c' with a' = k3 (register), b' = lo, c' = hi
        // pass 1: DFT over a', twiddle W_4096^(m1 (16 lo + hi)) (lane-ordered table)
        twiddle_pass<WPS>(v, ta, Tw1Fetch{p.tw1p, t});
        clk.template stamp<7>();
        tw_early<WPS>(ta, Tw2Fetch{tw2s, hi});
        exchange_local(v, xw, hi & 3, lo); // (c' = hi, b' = lo) -> (c' = hi, m1 = lo): same wave
        if (PADDED && wave == 0 && lane == 0) { // every lane needs the means before its argmax: visible behind
            rec[32] = s1a;                      // the four barriers of the transpose below
            rec[33] = s1b;
        }
        clk.template stamp<8>();
        // pass 2: DFT over b', twiddle W_256^(m2 c'), c' = hi
        twiddle_pass<WPS>(v, ta, Tw2Fetch{tw2s, hi});
        clk.template stamp<9>();
        exchange_cross<1>(v, xbuf, wave, t);
        clk.template stamp<10>();
        // pass 3: DFT over c' (m1 = lo, m2 = hi): index t + 256 m3.  The next pair's rows
        // are requested first: they stay in flight during the butterflies and the argmax.
        {
            if (DYN)
                nextpair = __builtin_amdgcn_readfirstlane(next_s[parity]);
            long long nxt = nextpair; // last iteration: pair 0 (L2-resident dummy)
            nxt = nxt < total ? nxt : 0;
            fence();
            if (WPS < 4)
                issue_row_loads<PADDED>(raw, p, nxt, t, pad);
            const auto c1l = [&](int k) __attribute__((always_inline)) {
                return scalar_ptr_at(p.c1, 256 * ((k + 1) & ~1))[t - 256 * (k & 1)];
            };
            double ca[8], cb[8];
            if (PADDED) { // first half of the correction table (L2): in flight during the butterflies
#pragma unroll
                for (int k = 0; k < 8; k++)
                    ca[k] = c1l(k);
            }
            fence();
            dft16(v);
            fence();
            double2 w[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                w[k] = v[P16(k)];
            if (PADDED) { // cc(d - m 1_valid) = cc(d) - m c1, m = sum d / N
#pragma unroll
                for (int k = 0; k < 8; k++)
                    cb[k] = c1l(8 + k);
                const double mA = rec[32] * invN, mB = rec[33] * invN;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    v[k] = make_double2(fma(-mA, ca[k], w[k].x), fma(-mB, ca[k], w[k].y));
                fence();
#pragma unroll
                for (int k = 0; k < 8; k++)
                    v[8 + k] = make_double2(fma(-mA, cb[k], w[8 + k].x), fma(-mB, cb[k], w[8 + k].y));
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    v[k] = w[k];
            }
        }
        clk.template stamp<11>();
        // ---- maxAbsIndex (xcorr.go:39-50), as in xcorr_r16_occ4.hip
        double ma = 0.0, mb = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            ma = fmax(ma, fabs(v[k].x));
            mb = fmax(mb, fabs(v[k].y));
        }
        const double wa = wave_max_dpp(ma), wb = wave_max_dpp(mb);
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
        const double cc0a = v[0].x, cc0b = v[0].y;
        if (WPS >= 4) {
            long long nxt = nextpair;
            nxt = nxt < total ? nxt : 0;
            fence();
            issue_row_loads<PADDED>(raw, p, nxt, t, pad);
            fence();
        }
        if (lane == 0) { // {max |cc|, signed value (cc[0] when nothing is above 0), index}
            double *ra_ = rec + 6 * wave;
            ra_[0] = widxA == 0x7fffffff ? 0.0 : wa;
            ra_[1] = widxA == 0x7fffffff ? cc0a : svA; // wave 0 lane 0 holds cc[0]
            ra_[2] = (double)widxA;
            ra_[3] = widxB == 0x7fffffff ? 0.0 : wb;
            ra_[4] = widxB == 0x7fffffff ? cc0b : svB;
            ra_[5] = (double)widxB;
            if (wave == 0) {
                rec[32] = s1a;
                rec[33] = s1b;
            }
        }
        parity ^= 1;
        clk.template stamp<12>();
    }
    lds_barrier();
    {
        const double *const prec = red + REC * (parity ^ 1);
        if (t < 2 && prec[34] >= 0.0 && (t == 0 || prec[35] != 0.0)) {
            const long long row = (long long)prec[34] + t;
            if (finalize(prec, t, invN, invNm1, p.mv + row, p.lag + row)) {
                const int slot = atomicAdd(p.ovf_count, 1);
                p.ovf_list[slot] = row >> 1;
            }
        }
    }
    if (TIMING && p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < NPHASE; i++)
            p.dbg[((long long)blockIdx.x * 4 + wave) * NPHASE + i] = clk.acc[i];
    }
}

// ============================================================================
// Many references against one resident group in ONE pass over the rows
// (the README use case iterates references over a fixed set of
// series).  Each pair of series is read from HBM and forward-transformed once; its
// spectrum Z (DC bin zeroed) is parked in a 64 KB per-workgroup slice of a global
// scratch buffer (L2 / Infinity-Cache resident: every lane re-reads exactly the
// addresses it wrote) and re-loaded for references 1..R-1, each of which costs one
// spectrum multiply, one transform and one argmax.  Per (series, reference) that is
// (1 + R) / (2 R) of the single-reference transform work and 1/R of the HBM bytes.
//
// One prefetch buffer serves both kinds of iteration: 16 x (re, im) of Z, or
// 16 x (row A sample, row B sample) -- the same packing the transform starts from --
// so every path through the loop defines all of it (no stale live ranges).
namespace fast {

constexpr int MSTAT = 12; // per pair: [0,8) sum d^2 partials (2*wave + series), [8,10) sum d, [10] first row, [11] has second row
constexpr int MTRIP = 26; // per iteration: [0,24) argmax partials, [24] reference index (< 0: nothing to write), [25] pair parity

__device__ __forceinline__ bool finalize_multi(const double *tr, const double *st, const int series, const double invN,
                                               const double invNm1, double *mv_out, int *lag_out)
{
    double m[4], s[4], ix[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        m[w] = tr[6 * w + 3 * series];
        s[w] = tr[6 * w + 3 * series + 1];
        ix[w] = tr[6 * w + 3 * series + 2];
    }
    const double s2 = (st[series] + st[2 + series]) + (st[4 + series] + st[6 + series]);
    const Stat stt{st[8 + series], s2};
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
    const double var = variance(stt, invN, invNm1, zero, nan);
    const int idx = (best > 0.0) ? (int)bidx : 0;
    double y = __builtin_amdgcn_rsq(var);
    y = y * fma(-0.5 * var * y, y, 1.5);
    y = y * fma(-0.5 * var * y, y, 1.5);
    double mv = ((best > 0.0) ? bsv : s[0]) * y;
    int lag = idx > 2048 ? idx - 4096 : idx;
    if (zero) { mv = 0.0; lag = 0; }
    if (nan) { mv = __builtin_nan(""); lag = 0; }
    *mv_out = mv;
    *lag_out = lag;
    bool redo = nan;
    if (series == 0 && st[11] != 0.0) { // sigmas too far apart for one shared transform?
        const Stat sb{st[9], (st[1] + st[3]) + (st[5] + st[7])};
        bool zb, nb;
        const double varb = variance(sb, invN, invNm1, zb, nb);
        redo = redo || (!nb && sigma_spread_too_wide(var, varb));
    }
    return redo;
}

// 16-byte global store through a scalar base (+ element offset `off`) and a lane index
__device__ __forceinline__ void zstore(double2 *base, long long off, int idx, d2v val)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long u = (unsigned long long)base;
    asm volatile("" : "+s"(u));
    u += (unsigned long long)(off * 16);
    asm volatile("" : "+s"(u));
    ((d2v __attribute__((address_space(1))) *)u)[idx] = val;
#endif
}

// the previous iteration's results, written behind the first barrier of the current one
// (a free function on purpose: a by-reference lambda with two call sites is not inlined and
// drags the kernel argument struct into private memory)
__device__ __forceinline__ void finalize_prev_multi(const double *trip, const double *stats, const int cur_ip,
                                                    const int t, const double invN, const double invNm1,
                                                    double *const *mv_many, int *const *lag_many, int *ovf_count,
                                                    long long *ovf_list)
{
    const double *const tr = trip + MTRIP * (cur_ip ^ 1);
    if (t < 2 && tr[24] >= 0.0) {
        const double *const st = stats + MSTAT * (int)tr[25];
        if (t == 0 || st[11] != 0.0) {
            const int r = (int)tr[24];
            const long long row = (long long)st[10] + t;
            if (finalize_multi(tr, st, t, invN, invNm1, mv_many[r] + row, lag_many[r] + row) && r == 0) {
                const int slot = atomicAdd(ovf_count, 1);
                ovf_list[slot] = row >> 1;
            }
        }
    }
}

} // namespace fast

// PADDED (2048 < N < 4096): as in xcorr_fused_n4096_fast -- the parked spectrum keeps its DC bin and every
// reference's results are corrected by -m c1_r[index] (FusedParams::c1_many) before the argmax.
template <bool TIMING = false, bool PADDED = false>
__global__ __launch_bounds__(OCC_THREADS, 4) void xcorr_fused_n4096_multi(const FusedParams p)
{
    using namespace occ4;
    using namespace fast;
    constexpr int WPS = 4;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 tw2s[256];
    __shared__ double stats[2 * MSTAT];
    __shared__ double trip[2 * MTRIP];
    __shared__ int next_s[2];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hi = t >> 4, lo = t & 15;
    double2 *const xw = xbuf + XW * wave;
    const int pad = PADDED ? 4096 - p.N : 0;
    const double invN = PADDED ? 1.0 / (double)p.N : 1.0 / 4096.0, invNm1 = PADDED ? 1.0 / (double)(p.N - 1) : 1.0 / 4095.0;
    const int R = p.R;

    tw2s[t] = p.tw2[t];
    if (t < 2)
        trip[MTRIP * t + 24] = -1.0;
    __syncthreads();
    double2 *const zs = p.zscratch + (size_t)blockIdx.x * 4096; // resident grid: one slice per workgroup

    int ip = 0, pp = 0;
    const long long total = p.npairs;
    // prefetch buffer: rows (pre[i] = (A[t + 256 i], B[t + 256 i]), ka/kb = first samples) or Z
    double2 pre[16];
    double ka, kb;
    {
        RawPair raw;
        issue_row_loads<PADDED>(raw, p, blockIdx.x < total ? (long long)blockIdx.x : 0ll, t, pad);
#pragma unroll
        for (int i = 0; i < 16; i++)
            pre[i] = make_double2(raw.a[i], raw.b[i]);
        ka = raw.ka;
        kb = raw.kb;
    }
    // one flat loop over (pair, reference) iterations: the prefetch buffer is live across
    // exactly one back-edge
    double s1a = 0.0, s1b = 0.0;
    long long nextpair = 0;
    int r = 0;
#pragma clang loop unroll(disable)
    for (long long pair = blockIdx.x; pair < total;) {
        {
            const long long rA = 2 * pair;
            const bool hasB = rA + 1 < p.M;
            double *const st = stats + MSTAT * pp;
            double *const tr = trip + MTRIP * ip;
            double2 v[16], ta[8];
            double2 w[16];
            if (r == 0) {
                // ---------- rows -> Z = FFT(dA + i dB), as in xcorr_fused_n4096_fast
                if (t == 0) // claim the pair after this one (read at the last reference, many barriers later)
                    next_s[pp] = (int)gridDim.x + atomicAdd(p.work_counter, 1);
                {
                    double qa = 0.0, qb = 0.0;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const double da = pre[i].x - ka, db = pre[i].y - kb;
                        v[i] = make_double2(da, db);
                        qa = fma(da, da, qa);
                        qb = fma(db, db, qb);
                    }
                    qa = wave_sum_dpp(qa);
                    qb = wave_sum_dpp(qb);
                    if (lane == 0) {
                        st[2 * wave] = qa;
                        st[2 * wave + 1] = qb;
                    }
                    if (t == 0) {
                        st[10] = (double)rA;
                        st[11] = hasB ? 1.0 : 0.0;
                    }
                }
                twiddle_pass<WPS>(v, ta, Tw1Fetch{p.tw1, t});
                exchange_cross<0>(v, xbuf, wave, t);
                finalize_prev_multi(trip, stats, ip, t, invN, invNm1, p.mv_many, p.lag_many, p.ovf_count, p.ovf_list);
                twiddle_pass<WPS>(v, ta, Tw2Fetch{tw2s, lo});
                lds_barrier();
                exchange_local(v, xw, hi & 3, lo);
                dft16(v);
#pragma unroll
                for (int k = 0; k < 16; k++)
                    w[k] = v[P16(k)];
                s1a = readlane_f64(w[0].x, 0);
                s1b = readlane_f64(w[0].y, 0);
                if (!PADDED) {
                    w[0].x = (t == 0) ? 0.0 : w[0].x;
                    w[0].y = (t == 0) ? 0.0 : w[0].y;
                } else if (wave == 0 && lane == 0) { // every lane needs the means before each argmax of this pair
                    st[8] = s1a;
                    st[9] = s1b;
                }
                if (R > 1) { // park Z: lane t owns zs[256 k + t] (scalar bases: no hoisted VGPR addresses)
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        d2v z;
                        z.x = w[k].x;
                        z.y = w[k].y;
                        zstore(zs, 256 * ((k + 1) & ~1), t - 256 * (k & 1), z);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    w[k] = pre[k];
            }
            // ---------- V = Z * conj(X_r)/n ; cc = FFT(V) ; argmax
            {
                const double2 *xr;
                { // the table pointer is wave-uniform: keep it in SGPRs
                    const unsigned long long u = (unsigned long long)p.xcp_many[r];
                    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)u);
                    const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
                    xr = (const double2 *)(((unsigned long long)hi32 << 32) | lo32);
                }
                const auto xcl = [&](int j) { return ldg2(scalar_ptr_at(xr, 256 * ((j + 1) & ~1)), t - 256 * (j & 1)); };
                double2 xa[4], xb[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    xa[j] = xcl(j);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    xb[j] = xcl(4 + j);
#pragma unroll
                for (int b = 0; b < 4; b += 2) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        v[b * 4 + j] = cmul(w[b * 4 + j], xa[j]);
                    fence();
                    if ((b + 2) * 4 < 16) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            xa[j] = xcl((b + 2) * 4 + j);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        v[(b + 1) * 4 + j] = cmul(w[(b + 1) * 4 + j], xb[j]);
                    fence();
                    if ((b + 3) * 4 < 16) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            xb[j] = xcl((b + 3) * 4 + j);
                    }
                }
            }
            twiddle_pass<WPS>(v, ta, Tw1Fetch{p.tw1p, t});
            exchange_local(v, xw, hi & 3, lo);
            twiddle_pass<WPS>(v, ta, Tw2Fetch{tw2s, hi});
            exchange_cross<1>(v, xbuf, wave, t);
            if (r > 0) // this iteration's first workgroup barriers were the four above
                finalize_prev_multi(trip, stats, ip, t, invN, invNm1, p.mv_many, p.lag_many, p.ovf_count, p.ovf_list);
            dft16(v);
#pragma unroll
            for (int k = 0; k < 16; k++)
                w[k] = v[P16(k)];
            if (PADDED) { // cc(d - m 1_valid) = cc(d) - m c1_r
                const double *c1r;
                {
                    const unsigned long long u = (unsigned long long)p.c1_many[r];
                    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)u);
                    const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
                    c1r = (const double *)(((unsigned long long)hi32 << 32) | lo32);
                }
                const double mA = st[8] * invN, mB = st[9] * invN;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const double c = scalar_ptr_at(c1r, 256 * ((k + 1) & ~1))[t - 256 * (k & 1)];
                    w[k] = make_double2(fma(-mA, c, w[k].x), fma(-mB, c, w[k].y));
                }
            }
            double ma = 0.0, mb = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                ma = fmax(ma, fabs(w[k].x));
                mb = fmax(mb, fabs(w[k].y));
            }
            const double wa = wave_max_dpp(ma), wb = wave_max_dpp(mb);
            int widxA = 0x7fffffff, widxB = 0x7fffffff;
            double svA = 0.0, svB = 0.0;
            {
                unsigned long long selA = 0ull, selB = 0ull;
                int kA = 0, kB = 0, hiA = 0, hiB = 0;
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const unsigned long long mA_ = __ballot(fabs(w[k].x) == wa);
                    const unsigned long long mB_ = __ballot(fabs(w[k].y) == wb);
                    const bool hA = mA_ != 0ull, hB = mB_ != 0ull;
                    selA = hA ? mA_ : selA;
                    kA = hA ? k : kA;
                    hiA = hA ? __double2hiint(w[k].x) : hiA;
                    selB = hB ? mB_ : selB;
                    kB = hB ? k : kB;
                    hiB = hB ? __double2hiint(w[k].y) : hiB;
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
            const double cc0a = w[0].x, cc0b = w[0].y;
            // ---------- request the next iteration's input: Z again, or the next pair's rows
            fence();
            if (r + 1 < R) {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    pre[k] = ldg2(scalar_ptr_at((const double2 *)zs, 256 * ((k + 1) & ~1)), t - 256 * (k & 1));
                ka = 0.0;
                kb = 0.0;
            } else {
                nextpair = __builtin_amdgcn_readfirstlane(next_s[pp]);
                long long nxt = nextpair < total ? nextpair : 0; // nothing left: pair 0 (L2-resident dummy)
                RawPair raw;
                issue_row_loads<PADDED>(raw, p, nxt, t, pad);
#pragma unroll
                for (int i = 0; i < 16; i++)
                    pre[i] = make_double2(raw.a[i], raw.b[i]);
                ka = raw.ka;
                kb = raw.kb;
            }
            fence();
            if (lane == 0) {
                double *ra_ = tr + 6 * wave;
                ra_[0] = widxA == 0x7fffffff ? 0.0 : wa;
                ra_[1] = widxA == 0x7fffffff ? cc0a : svA;
                ra_[2] = (double)widxA;
                ra_[3] = widxB == 0x7fffffff ? 0.0 : wb;
                ra_[4] = widxB == 0x7fffffff ? cc0b : svB;
                ra_[5] = (double)widxB;
                if (wave == 0) {
                    tr[24] = (double)r;
                    tr[25] = (double)pp;
                    if (r == 0) {
                        st[8] = s1a;
                        st[9] = s1b;
                    }
                }
            }
            ip ^= 1;
        }
        if (++r == R) {
            r = 0;
            pair = nextpair;
            pp ^= 1;
        }
    }
    lds_barrier();
    finalize_prev_multi(trip, stats, ip, t, invN, invNm1, p.mv_many, p.lag_many, p.ovf_count, p.ovf_list);
}

// R >= 2 references, N == n == 4096; p.ovf_count and p.work_counter zeroed; a resident grid
// (pairs are handed out dynamically), one scratch slice per workgroup
hipError_t launch_fused_multi(const FusedParams &p, int num_cus, hipStream_t stream)
{
    int wpc = 4;
    if (const char *d = getenv("MUSE_HIP_MULTI_WPC")) // tuning aid: resident workgroups per CU (1..4)
        wpc = std::max(1, std::min(4, atoi(d)));
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * wpc);
    if (p.zslots < grid || !p.zscratch || !p.work_counter || p.R < 1)
        return hipErrorInvalidValue;
    if (p.N < 4096) {
        if (!p.c1_many)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL((xcorr_fused_n4096_multi<false, true>), dim3((unsigned)grid), dim3(OCC_THREADS), 0, stream, p);
    } else {
        hipLaunchKernelGGL((xcorr_fused_n4096_multi<false, false>), dim3((unsigned)grid), dim3(OCC_THREADS), 0, stream, p);
    }
    return hipGetLastError();
}

// n == 4096 (N < 4096: p.c1 required); p.ovf_count / work_counter must be zeroed and p.ovf_list hold 2*npairs entries
hipError_t launch_fused_fast(const FusedParams &p, int num_cus, hipStream_t stream)
{
    long long grid = p.npairs;
    int mult = 16; // see launch_fused_occ4
    if (const char *m = getenv("MUSE_HIP_GRID_MULT"))
        mult = atoi(m) > 0 ? atoi(m) : mult;
    // 4 waves/SIMD (128 VGPRs, late row request) measured 10.7-10.9 ms per 1 M series against
    // 11.0-11.1 ms for the 3-wave build with the early row request (same box, interleaved)
    int wps = 4;
    if (const char *w = getenv("MUSE_HIP_FAST_WPS")) // tuning aid
        wps = atoi(w) == 3 ? 3 : 4;
    const long long cap = (long long)num_cus * wps * mult;
    if (grid > cap)
        grid = cap;
    // dynamic hand-out on a resident grid measured 10.58 ms against 10.74 ms for the static stride
    // over 16x the resident set (same box, interleaved)
    int dyn = 1;
    if (const char *d = getenv("MUSE_HIP_FAST_DYN")) // tuning aid
        dyn = atoi(d) != 0;
    FusedParams q = p;
    if (const char *d = getenv("MUSE_HIP_FAST_TUNE")) // tuning aid: experiment bits
        q.tune = atoi(d);
    if (p.N < 4096) { // leading zero pad: needs the batch's correction table
        if (!p.c1 || !p.work_counter)
            return hipErrorInvalidValue;
        grid = std::min<long long>(p.npairs, (long long)num_cus * 4);
        hipLaunchKernelGGL((xcorr_fused_n4096_fast<4, false, true, true>), dim3((unsigned)grid), dim3(OCC_THREADS), 0, stream, q);
        return hipGetLastError();
    }
    if (dyn && p.work_counter) {
        int wpc = wps;
        if (const char *d = getenv("MUSE_HIP_FAST_WPC")) // tuning aid: resident workgroups per CU
            wpc = std::max(1, std::min(wps, atoi(d)));
        grid = std::min<long long>(p.npairs, (long long)num_cus * wpc); // resident workgroups only
        if (wps == 4)
            hipLaunchKernelGGL((xcorr_fused_n4096_fast<4, false, true>), dim3((unsigned)grid), dim3(OCC_THREADS), 0, stream, q);
        else
            hipLaunchKernelGGL((xcorr_fused_n4096_fast<3, false, true>), dim3((unsigned)grid), dim3(OCC_THREADS), 0, stream, q);
    } else if (wps == 4)
        hipLaunchKernelGGL((xcorr_fused_n4096_fast<4, false>), dim3((unsigned)grid), dim3(OCC_THREADS), 0, stream, q);
    else
        hipLaunchKernelGGL((xcorr_fused_n4096_fast<3, false>), dim3((unsigned)grid), dim3(OCC_THREADS), 0, stream, q);
    return hipGetLastError();
}

} // namespace muse
