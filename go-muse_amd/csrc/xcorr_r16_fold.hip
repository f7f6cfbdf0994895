// xcorr_r16_fold.hip -- the default n = 4096 fp64 kernel (2048 < N <= 4096), round 2.
//
// Mathematics: xCorrWithX, /root/reference/xcorr.go:160-197 (z-normalise, leading zero pad, forward
// transform, multiply by the conjugate reference spectrum, inverse transform, 1/n, global argmax of |cc|),
// two series per complex transform as in xcorr_r16_fast.hip, whose data flow this kernel keeps: 256 threads x
// 16 points, three radix-16 passes per transform, four LDS transposes per pair of which two stay inside a
// wave, nine workgroup barriers, statistics off the DC bin, a resident grid with dynamic pair hand-out,
// NaN/Inf and sigma-spread pairs handed to the rescaling kernel.
//
// What is new is the ARITHMETIC (fold_device.h).  The round-1 kernel was bound by fp64 VALU issue
// (1 456 v_*_f64 per wave per pair).  n = 16^3, input index j = 256 a + 16 b + c, output f = k1 + 16 k2 + 256 k3:
//     X[f] = sum_c W_16^(c (k3 + (k1 + 16 k2)/256))  sum_b W_16^(b (k2 + k1/16))  sum_a W_16^(a k1) x[j]
// i.e. a plain 16-point DFT followed by two GENERALISED 16-point DFTs whose per-thread phase shift delta
// (k1/16 for pass 2, (k1 + 16 k2)/256 for pass 3) carries what used to be 2 x 15 twiddle multiplications per
// thread; each generalised pass is 32 butterflies of 6 FMAs (192 instead of 160 + 60), the plain pass 148
// instead of 160, and the multiplication by the reference spectrum is folded into the first stage of the second
// transform's plain pass (196 instead of 64 + 160).  Per thread and pair: 532 + 580 instead of 2 x 600 + 64
// fp64 instructions in the transforms, and 8 + 16 + 8 table loads from L2 instead of 15 + 16 + 15
// (fold_device.h: every stage's twiddles are one of eight per-thread constants times 1 or -i).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

#include "foldk_device.h"

namespace muse {


// PADDED (2048 < N < 4096, leading zero pad): as in xcorr_r16_fast.hip -- the transforms run on d, sum d is read
// off the DC bin and the 16 values a lane ends with are corrected by -m c1[index] before the argmax.
// F32: float32-storage group (half the HBM bytes; samples widened exactly on consumption, same float64 arithmetic)
template <bool TIMING = false, bool PADDED = false, bool F32 = false>
__global__ __launch_bounds__(OCC_THREADS, 4) void xcorr_fused_n4096_fold(const FusedParams p)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[2 * REC];
    __shared__ int next_s[2];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform by construction
    const int hi = t >> 4, lo = t & 15;
    double2 *const xw = xbuf + XW * wave;
    const int pad = PADDED ? 4096 - p.N : 0;
    // (PADDED: the launcher's quotients, kernel arguments in scalar registers -- formed here they came out of the vector divider and
    // were parked in scratch across the whole pair loop)
    const double invN = PADDED ? p.invN : 1.0 / 4096.0, invNm1 = PADDED ? p.invNm1 : 1.0 / 4095.0;

    if (t < 128)
        g2s[t] = p.g2[t];
    if (t < 2)
        red[REC * (t) + 34] = -1.0; // no previous pair yet (either parity)
    __syncthreads();
    PhaseClock<TIMING> clk;
    clk.start();
    // (PADDED: the wave-local transposes' LDS addresses are formed where they are used -- hoisted out of the pair loop they were
    // parked in scratch and reloaded behind an s_waitcnt vmcnt(0) in the middle of every pair)
    const auto tl = [&]() __attribute__((always_inline)) {
        int x = t;
        if (PADDED)
            asm volatile("" : "+v"(x));
        return x;
    };

    int parity = 0;
    const long long total = p.npairs;
    RawPair raw;
    // N == n float64 rows: 16-byte requests on relabelled columns (r16_device.h, issue_row_loads_wide): -1.6 % (profiles/r03_fold_variants.txt)
    constexpr bool WIDE = !PADDED && !F32;
    const auto request_rows = [&](long long pr) __attribute__((always_inline)) {
        if (WIDE)
            issue_row_loads_wide(raw, p, pr, t);
        else
            issue_row_loads<PADDED, F32>(raw, p, pr, t, pad);
    };
    request_rows(blockIdx.x < total ? (long long)blockIdx.x : 0ll);

    long long nextpair = 0;
    for (long long pair = blockIdx.x; pair < total; pair = nextpair) {
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        double *const rec = red + REC * parity;
        const double *const prec = red + REC * (parity ^ 1);
        if (t == 0) // the pair after this one: claimed now, read behind this pair's barriers
            next_s[parity] = (int)gridDim.x + atomicAdd(p.work_counter, 1);
        // ---- consume the prefetched rows: d = x - K (K = the row's first sample) bounds the cancellation in
        // sum d^2 - (sum d)^2 / N and keeps a large level out of the transform's rounding; sum d is read off the DC bin
        double2 v[16];
        {
            const double KA = raw.ka, KB = raw.kb;
            double qa = 0.0, qb = 0.0;
            if (TIMING)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            clk.template stamp<0>();
            if (WIDE)
                widen_rows(raw);
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
        // pass 1: plain DFT over a (thread (b, c) = (hi, lo)) -> k1 at v[BR16(k1)]
        dft16_nr(v);
        clk.template stamp<2>();
        // -> thread (k1 = hi, c = lo), input b at v[b]; the tail barrier frees the wave's private quarter for the
        // wave-local transpose below while the waves are still in step
        exchange_cross<0, 1, true>(v, xbuf, wave, t, WIDE ? wide_column(t) : -1);
        // the previous pair's record is complete and visible: lane 0 of waves 0 / 1 writes one series' result each
        if (lane == 0 && wave < 2 && prec[34] >= 0.0 && (wave == 0 || prec[35] != 0.0)) {
            const long long row = (long long)prec[34] + wave;
            if (finalize(prec, wave, invN, invNm1, p.mv + row, p.lag + row)) {
                const int slot = atomicAdd(p.ovf_count, 1);
                p.ovf_list[slot] = row >> 1;
            }
        }
        clk.template stamp<3>();
        // pass 2: generalised DFT over b, delta = k1 / 16 (carries W_256^(b k1))
        gdft16_nr(v, G2Fetch{g2s, t >> 4});
        clk.template stamp<4>();
        exchange_local<1>(v, xw, tl()); // -> thread (k1 = hi, k2 = lo), input c at v[c]
        clk.template stamp<5>();
        // pass 3: generalised DFT over c, delta = (k1 + 16 k2) / 256 (carries W_4096^(c k1) W_256^(c k2));
        // Z[hi + 16 lo + 256 k3] at v[BR16(k3)]
        gdft16_nr_l2(v, G3Derived(p.g3a, t));
        double s1a, s1b;
        {
            // bin 0 (lane 0 of wave 0) = (sum dA, sum dB): kept in SGPRs until the record is written; the centred
            // series' DC bin is exactly 0.  Branch-free on purpose.
            s1a = readlane_f64(v[0].x, 0);
            s1b = readlane_f64(v[0].y, 0);
            if (!PADDED) {
                v[0].x = (t == 0) ? 0.0 : v[0].x;
                v[0].y = (t == 0) ? 0.0 : v[0].y;
            }
        }
        clk.template stamp<6>();
        // ================= ccA + i ccB = FFT(Z conj(X)/n) (unscaled by 1/sigma) =================
        // element f = 256 a' + 16 b' + c' with a' = k3 (register BR16(a')), b' = lo, c' = hi
        // pass 1: plain DFT over a' with the spectrum factors folded into its first stage -> m1 at v[m1]
        xc_stage1(v, [&](int j) __attribute__((always_inline)) {
            return ldg2(scalar_ptr_at(p.xcp, 256 * ((j + 1) & ~1)), t - 256 * (j & 1));
        });
        dft16_rn_s234(v);
        clk.template stamp<7>();
        exchange_local<0>(v, xw, tl()); // (c' = hi, b' = lo) -> (c' = hi, m1 = lo), input b' at v[b']: same wave
        if (PADDED && wave == 0 && lane == 0) { // every lane needs the means before its argmax: visible behind
            rec[32] = s1a;                      // the four barriers of the transpose below
            rec[33] = s1b;
        }
        clk.template stamp<8>();
        // pass 2: generalised DFT over b', delta = m1 / 16, m1 = lo
        gdft16_nr(v, G2Fetch{g2s, t & 15});
        clk.template stamp<9>();
        exchange_cross<1, 1>(v, xbuf, wave, t); // -> thread (m1 = lo, m2 = hi), input c' at v[c']
        clk.template stamp<10>();
        // pass 3: generalised DFT over c', delta = (m1 + 16 m2) / 256 = t / 256: cc index t + 256 m3 at v[BR16(m3)]
        nextpair = __builtin_amdgcn_readfirstlane(next_s[parity]);
        long long nxt = nextpair; // last iteration: pair 0 (L2-resident dummy)
        nxt = nxt < total ? nxt : 0;
        gdft16_nr_l2(v, G3Derived(p.g3b, t));
        if (PADDED) { // cc(d - m 1_valid) = cc(d) - m c1, m = sum d / N
            const auto c1l = [&](int k) __attribute__((always_inline)) {
                return scalar_ptr_at(p.c1, 256 * ((k + 1) & ~1))[t - 256 * (k & 1)];
            };
            const double mA = rec[32] * invN, mB = rec[33] * invN;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                double cq[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    cq[k] = c1l(8 * h + k);
                fence();
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int r = BR16(8 * h + k);
                    v[r] = make_double2(fma(-mA, cq[k], v[r].x), fma(-mB, cq[k], v[r].y));
                }
            }
        }
        clk.template stamp<11>();
        // float32 rows take 34 registers instead of 66: room to request them BEFORE the argmax, which then runs under
        // the HBM latency (float64 rows: behind it, there is no register left to land them in)
        if (F32) {
            fence();
            request_rows(nxt);
            fence();
        }
        wave_argmax_store(v, wave, lane, rec + 6 * wave);
        clk.template stamp<13>();
        fence();
        if (!F32)
            request_rows(nxt);
        fence();
        if (wave == 0 && lane == 0) {
            rec[32] = s1a;
            rec[33] = s1b;
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
// spectrum Z (DC bin zeroed) STAYS IN REGISTERS for the R references, each of which costs one
// spectrum multiply, one transform and one argmax.  Per (series, reference) that is
// (1 + R) / (2 R) of the single-reference transform work and 1/R of the HBM bytes.
// (Round 2 parked Z in a 64 KB slice of global scratch per resident workgroup: 67 MB against
// 4 MB of L2 per XCD, i.e. 9 x the row bytes through the memory side,
// profiles/r01_many_refs_rocprof_summary.txt; that traffic is gone.)
//
// Two workgroups per CU with 256 registers per lane and 72 KB of LDS each (measured against three at
// 168 registers with a quarter of Z in LDS: +4 %, profiles/r03_many_references.txt), spent on what
// hides latency at that occupancy: every transpose goes through a FULL 16 x 272 buffer (two barriers
// instead of five), the next reference's sixteen spectrum factors are requested one iteration ahead,
// the eight pass-3 factors travel under the transpose in front of the pass (R = 8, 400 000 x 4096:
// 1.64 -> 1.70 -> 1.76 -> 1.86e8 series-references/s step by step, profiles/r03_many_references.txt).  The last reference of
// a pair is peeled out of the reference loop: Z dies in its first stage, and its registers take the
// next pair's rows, requested right behind that stage.
namespace foldk {

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

// the previous iteration's results, written behind the first barrier of the current one
// (a free function on purpose: a by-reference lambda with two call sites is not inlined and
// drags the kernel argument struct into private memory)
__device__ __forceinline__ void finalize_prev_multi(const double *trip, const double *stats, const int cur_ip,
                                                    const int t, const double invN, const double invNm1,
                                                    double *const *mv_many, int *const *lag_many, int *ovf_count,
                                                    long long *ovf_list)
{
    const double *const tr = trip + MTRIP * (cur_ip ^ 1);
    const int series = t >> 6; // lane 0 of waves 0 / 1 writes one series' result each
    if ((t & 63) == 0 && series < 2 && tr[24] >= 0.0) {
        const double *const st = stats + MSTAT * (int)tr[25];
        if (series == 0 || st[11] != 0.0) {
            const int r = (int)tr[24];
            const long long row = (long long)st[10] + series;
            if (finalize_multi(tr, st, series, invN, invNm1, mv_many[r] + row, lag_many[r] + row) && r == 0) {
                const int slot = atomicAdd(ovf_count, 1);
                ovf_list[slot] = row >> 1;
            }
        }
    }
}

} // namespace foldk

constexpr int MULTI_WGS_PER_CU = 2;

// PADDED (2048 < N < 4096): the spectrum keeps its DC bin and every reference's results are corrected by
// -m c1_r[index] (FusedParams::c1_many) before the argmax.
// F32: float32-storage group (rows widened as they are consumed, float64 arithmetic)
// TIMING: the phase-stamped build of tools/ablate/multi_phases.hip (FusedParams::dbg); the product kernel below is <.., false>.
template <bool PADDED, bool F32, bool TIMING>
__device__ __forceinline__ void fold_multi_body(const FusedParams &p)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    __shared__ double2 xbuf[OCC_XBUF_FULL];
    __shared__ double2 g2s[128];
    __shared__ double stats[2 * MSTAT];
    __shared__ double trip[2 * MTRIP];
    __shared__ int next_s[2];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + 1088 * wave;
    const int pad = PADDED ? 4096 - p.N : 0;
    const double invN = PADDED ? p.invN : 1.0 / 4096.0, invNm1 = PADDED ? p.invNm1 : 1.0 / 4095.0; // (the launcher's quotients: scalar registers)
    const int R = p.R;
    PhaseClock<TIMING> clk;

    if (t < 128)
        g2s[t] = p.g2[t];
    if (t < 2)
        trip[MTRIP * t + 24] = -1.0;
    __syncthreads();
    clk.start();

    int ip = 0, pp = 0;
    const long long total = p.npairs;
    RawPair raw;
    constexpr bool WIDE = !PADDED && !F32;
    const auto request_rows = [&](long long pr) __attribute__((always_inline)) {
        if (WIDE)
            issue_row_loads_wide(raw, p, pr, t);
        else
            issue_row_loads<PADDED, F32>(raw, p, pr, t, pad);
    };
    request_rows(blockIdx.x < total ? (long long)blockIdx.x : 0ll);
    // reference r's sixteen spectrum factors of this thread (lane-ordered table, L2): requested one iteration ahead
    const auto request_spectrum = [&](double2 (&xf)[16], const int r) __attribute__((always_inline)) {
        const double2 *xr;
        { // the table pointer is wave-uniform: keep it in SGPRs
            const unsigned long long u = (unsigned long long)p.xcp_many[r];
            const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)u);
            const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            xr = (const double2 *)(((unsigned long long)hi32 << 32) | lo32);
        }
#pragma unroll
        for (int j = 0; j < 16; j++)
            xf[j] = ldg2(scalar_ptr_at(xr, 256 * ((j + 1) & ~1)), t - 256 * (j & 1));
    };

    // one reference: V = Z conj(X_r) / n (factors xf, requested earlier), cc = FFT(V), argmax -> trip[ip]; `v` holds Z on entry.
    // `ahead()` runs in front of the last stage of the last pass: the place to request what the NEXT iteration needs.
    // `early()` runs right behind the spectrum multiply (the last reference: Z is dead there, its registers can take the next pair's rows).
    // `g3_first`: pass 3's factors are requested IN FRONT of early() -- loads return in order, so a factor requested behind the
    // next pair's rows could only be used once those rows (HBM latency) have arrived.
    const auto correlate = [&](double2 (&v)[16], const double2 (&xf)[16], const int r, const double *st, auto g3_first, auto early, auto ahead) __attribute__((always_inline)) {
        double *const tr = trip + MTRIP * ip;
        clk.template stamp<13>();
        xc_stage1_pre(v, xf);
        fence();
        double2 g3[8];
        if (decltype(g3_first)::value) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                g3[q] = G3Fetch{p.g3b, t}(q);
            fence();
        }
        early();
        fence();
        clk.template stamp<6>();
        dft16_rn_s234(v);
        clk.template stamp<7>();
        exchange_local_full<0>(v, xw, t);
        clk.template stamp<8>();
        gdft16_nr(v, G2Fetch{g2s, t & 15});
        if (!decltype(g3_first)::value) { // pass 3's factors travel under the transpose
#pragma unroll
            for (int q = 0; q < 8; q++)
                g3[q] = G3Fetch{p.g3b, t}(q);
        }
        fence();
        clk.template stamp<9>();
        exchange_cross_full<1, 1>(v, xbuf, t);
        if (r > 0) // (r == 0: the previous iteration's results went out behind the forward transform's first barriers)
            finalize_prev_multi(trip, stats, ip, t, invN, invNm1, p.mv_many, p.lag_many, p.ovf_count, p.ovf_list);
        clk.template stamp<10>();
        gdft16_nr_s12(v, g3[0], g3[1]); // cc index t + 256 m3 at v[BR16(m3)]
        gdft16_nr_s3(v, g3[2], g3[3]);
        fence();
        ahead();
        fence();
        gdft16_nr_s4(v, g3[4], g3[5], g3[6], g3[7]);
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
            for (int h = 0; h < 4; h++) {
                double cq[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    cq[k] = scalar_ptr_at(c1r, 256 * ((4 * h + k + 1) & ~1))[t - 256 * ((4 * h + k) & 1)];
                fence();
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int q = BR16(4 * h + k);
                    v[q] = make_double2(fma(-mA, cq[k], v[q].x), fma(-mB, cq[k], v[q].y));
                }
            }
        }
        clk.template stamp<11>();
        wave_argmax_store(v, wave, lane, tr + 6 * wave);
        if (wave == 0 && lane == 0) {
            tr[24] = (double)r;
            tr[25] = (double)pp;
        }
        ip ^= 1;
        clk.template stamp<12>();
    };

    long long nextpair = 0;
#pragma clang loop unroll(disable)
    for (long long pair = blockIdx.x; pair < total; pair = nextpair) {
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        double *const st = stats + MSTAT * pp;
        if (t == 0) // claim the pair after this one (read at the last reference, many barriers later)
            next_s[pp] = (int)gridDim.x + atomicAdd(p.work_counter, 1);
        // ---------- rows -> Z = FFT(dA + i dB), as in xcorr_fused_n4096_fold
        double2 Z[16];
        {
            const double KA = raw.ka, KB = raw.kb;
            double qa = 0.0, qb = 0.0;
            if (WIDE)
                widen_rows(raw);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const double da = raw.a[i] - KA, db = raw.b[i] - KB;
                Z[i] = make_double2(da, db);
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
        double2 xf[16];
        fence();
        clk.template stamp<0>();
        request_spectrum(xf, 0); // (the rows' registers are free again: the first reference's factors travel under the forward transform)
        fence();
        dft16_nr(Z);
        clk.template stamp<1>();
        exchange_cross_full<0, 1>(Z, xbuf, t, WIDE ? wide_column(t) : -1);
        finalize_prev_multi(trip, stats, ip, t, invN, invNm1, p.mv_many, p.lag_many, p.ovf_count, p.ovf_list);
        clk.template stamp<2>();
        gdft16_nr(Z, G2Fetch{g2s, t >> 4});
        {
            double2 g3[8]; // pass 3's factors travel under the transpose
#pragma unroll
            for (int q = 0; q < 8; q++)
                g3[q] = G3Fetch{p.g3a, t}(q);
            fence();
            clk.template stamp<3>();
            exchange_local_full<1>(Z, xw, t);
            clk.template stamp<4>();
            gdft16_nr_s12(Z, g3[0], g3[1]); // Z[hi + 16 lo + 256 k3] at Z[BR16(k3)]
            gdft16_nr_s3(Z, g3[2], g3[3]);
            gdft16_nr_s4(Z, g3[4], g3[5], g3[6], g3[7]);
        }
        {
            const double s1a = readlane_f64(Z[0].x, 0), s1b = readlane_f64(Z[0].y, 0);
            if (!PADDED) {
                Z[0].x = (t == 0) ? 0.0 : Z[0].x;
                Z[0].y = (t == 0) ? 0.0 : Z[0].y;
            }
            if (wave == 0 && lane == 0) { // (PADDED: every lane needs the means before each argmax of this pair: many barriers away)
                st[8] = s1a;
                st[9] = s1b;
            }
        }
        clk.template stamp<5>();
#pragma clang loop unroll(disable)
        for (int r = 0; r + 1 < R; r++) {
            double2 v[16], xn[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                v[k] = Z[k];
            correlate(v, xf, r, st, std::false_type{}, [&]() __attribute__((always_inline)) {}, [&]() __attribute__((always_inline)) { request_spectrum(xn, r + 1); });
#pragma unroll
            for (int k = 0; k < 16; k++)
                xf[k] = xn[k];
        }
        {   // the last reference: Z dies in its first stage and the next pair's rows are requested right behind it -- a whole
            // reference iteration (~ 5 us) ahead of their use
            nextpair = __builtin_amdgcn_readfirstlane(next_s[pp]);
            const long long nxt = nextpair < total ? nextpair : 0; // nothing left: pair 0 (L2-resident dummy)
            correlate(Z, xf, R - 1, st, std::true_type{}, [&]() __attribute__((always_inline)) { request_rows(nxt); }, [&]() __attribute__((always_inline)) {});
        }
        pp ^= 1;
    }
    lds_barrier();
    finalize_prev_multi(trip, stats, ip, t, invN, invNm1, p.mv_many, p.lag_many, p.ovf_count, p.ovf_list);
    if (TIMING && p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < NPHASE; i++)
            p.dbg[((long long)blockIdx.x * 4 + wave) * NPHASE + i] = clk.acc[i];
    }
}

template <bool PADDED = false, bool F32 = false>
__global__ __launch_bounds__(OCC_THREADS, MULTI_WGS_PER_CU) void xcorr_fused_n4096_fold_multi(const FusedParams p)
{
    fold_multi_body<PADDED, F32, false>(p);
}

// R >= 1 references, n == 4096 (N < 4096: p.c1_many); p.ovf_count and p.work_counter zeroed; a resident grid
// (pairs are handed out dynamically)
hipError_t launch_fused_multi(const FusedParams &p_in, int num_cus, hipStream_t stream)
{
    const FusedParams p = with_reciprocals(p_in);
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * MULTI_WGS_PER_CU);
    if (!p.work_counter || p.R < 1 || !p.g2 || !p.g3a || !p.g3b || !p.xcp_many || !p.mv_many || !p.lag_many)
        return hipErrorInvalidValue;
    const dim3 g((unsigned)grid), b(OCC_THREADS);
    if (p.N < 4096) {
        if (!p.c1_many)
            return hipErrorInvalidValue;
        if (p.rows32)
            hipLaunchKernelGGL((xcorr_fused_n4096_fold_multi<true, true>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_fused_n4096_fold_multi<true, false>), g, b, 0, stream, p);
    } else if (p.rows32) {
        hipLaunchKernelGGL((xcorr_fused_n4096_fold_multi<false, true>), g, b, 0, stream, p);
    } else {
        hipLaunchKernelGGL((xcorr_fused_n4096_fold_multi<false, false>), g, b, 0, stream, p);
    }
    return hipGetLastError();
}

// n == 4096 (N < 4096: p.c1 required); p.ovf_count / work_counter must be zeroed and p.ovf_list hold 2*npairs entries;
// a resident grid, pairs handed out by the atomic counter
hipError_t launch_fused_fold(const FusedParams &p_in, int num_cus, hipStream_t stream)
{
    const FusedParams p = with_reciprocals(p_in);
    if (!p.work_counter || !p.g2 || !p.g3a || !p.g3b || !p.xcp)
        return hipErrorInvalidValue;
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * 4);
    if (p.N < 4096 && !p.c1) // leading zero pad: needs the batch's correction table
        return hipErrorInvalidValue;
    const dim3 g((unsigned)grid), b(OCC_THREADS);
    if (p.rows32) {
        if (p.N < 4096)
            hipLaunchKernelGGL((xcorr_fused_n4096_fold<false, true, true>), g, b, 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_fused_n4096_fold<false, false, true>), g, b, 0, stream, p);
    } else if (p.N < 4096) {
        hipLaunchKernelGGL((xcorr_fused_n4096_fold<false, true>), g, b, 0, stream, p);
    } else {
        hipLaunchKernelGGL((xcorr_fused_n4096_fold<false, false>), g, b, 0, stream, p);
    }
    return hipGetLastError();
}

} // namespace muse
