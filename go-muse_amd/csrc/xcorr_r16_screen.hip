// xcorr_r16_screen.hip -- the fp32 screening pass of the OPT-IN filter-and-refine Run, n = 4096
// (muse_ctx_set_screening; docs/HISTORY.md 4.6).  Not on the default path: by default every series is scored by
// the float64 kernel (xcorr_r16_fold.hip), the arithmetic of the reference (xcorr.go:160-197).
//
// What the pass does: z-normalisation statistics in fp64 while the rows arrive, both transforms in fp32 on a
// power-of-two-scaled fp32 copy of the centred samples, and per series an ESTIMATE of the score plus FLAGS that
// describe every lag whose fp32 |cc| lies within the error bound of the fp32 maximum (kernel comment below).
// Nothing the library returns is computed here: rows that can reach the top-N are re-evaluated by the fp64
// kernels and only those are selected (capi_screen.hip, screen_finish).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fft_device.h"
#include "xcorr_kernels.h"

namespace muse {

constexpr int SCR_THREADS = 256;
constexpr int SCR_XBUF = 16 * 272; // f2 elements: 34,816 B

namespace scr {

__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void lds_barrier()
{
    fence();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    fence();
}
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
using gptr = const T __attribute__((address_space(1))) *;
#else
template <typename T>
using gptr = const T *; // host pass only parses this file
#endif
// uniform pointer re-materialised in SGPRs, global address space kept (see
// xcorr_r16_occ4.hip: a generic pointer would turn every load into flat_load)
template <typename T>
__device__ __forceinline__ gptr<T> scalar_ptr(const T *p)
{
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+s"(u));
    return (gptr<T>)u;
}
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 ldg_f2(gptr<float2> p, int i) // one 8-byte global load
{
    const f2v x = ((gptr<f2v>)p)[i];
    return mk2(x.x, x.y);
}

__device__ __forceinline__ double uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Full-size transposes, 8-byte elements (positions in f2 units, row stride 272
// for both so that ds_read_b64's 32-lane groups see 32 distinct slots mod 32):
//   A: writer (b = hi, c = lo) output k1 -> 272*k1 + t ; reader (k1 = hi, c = lo) input b <- 272*hi + 16*b + lo
//   B: writer (k1 = hi, c = lo) output k2 -> 272*k2 + 17*hi + lo ; reader (k1 = lo, k2 = hi) input c <- 272*hi + 17*lo + c
template <bool B>
__device__ __forceinline__ void exchange(f2 (&v)[16], f2 *xbuf, const int t)
{
    const int hi = t >> 4, lo = t & 15;
    const int wbase = B ? 17 * hi + lo : t;
    const int rbase = 272 * hi + (B ? 17 * lo : lo);
    lds_barrier(); // previous readers done
#pragma unroll
    for (int k = 0; k < 16; k++)
        xbuf[272 * k + wbase] = v[P16(k)];
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 16; e++)
        v[e] = xbuf[rbase + (B ? e : 16 * e)];
}

struct Tw1FetchF {
    const float2 *p;
    int t;
    __device__ __forceinline__ f2 operator()(int k) const
    {   // one scalar base per two rows: the odd row sits at immediate offset -2048 B
        unsigned long long u = (unsigned long long)p;
        asm volatile("" : "+s"(u));
        u += (unsigned long long)(((k + 1) & ~1) * 256) * sizeof(float2);
        asm volatile("" : "+s"(u));
        return ldg_f2((gptr<float2>)u, t - 256 * (k & 1));
    }
};

// one series' 16 samples per thread (element t + 256 i) + its first sample; coalesced
// nontemporal loads, one scalar base per four 2 KB slices (no 64-bit VALU address arithmetic)
__device__ __forceinline__ void issue_series(double (&r)[16], double &k0, const double *row, const int t)
{
    k0 = scalar_ptr(row)[0];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int c = (i & ~3) * 256 + 512;
        unsigned long long u = (unsigned long long)row;
        asm volatile("" : "+s"(u));
        u += (unsigned long long)c * sizeof(double);
        asm volatile("" : "+s"(u));
        r[i] = __builtin_nontemporal_load((gptr<double>)u + (256 * i - c) + t);
    }
}

// N < 4096 (leading zero pad of 4096 - N < 2048 samples): element t + 256 i of the padded series is sample
// t + 256 i - pad; a pad position loads sample 0 instead (clamped index), i.e. the shift point x[0], so its
// d = x - x[0] is exactly 0 without a mask
__device__ __forceinline__ void issue_series_padded(double (&r)[16], double &k0, const double *row, const int t, const int pad)
{
    const gptr<double> rp = scalar_ptr(row);
    k0 = rp[0];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i < 8) { // may fall into the pad: clamped 12-bit index (saddr + 32-bit voffset, no sign extension per load)
            int j = t + 256 * i - pad;
            j = j < 0 ? 0 : j;
            r[i] = __builtin_nontemporal_load(rp + ((unsigned)j & 4095u));
        } else { // elements 2048.. are always samples (pad < 2048): scalar base row + 256 i - pad, shared voffset t
            unsigned long long u = (unsigned long long)row;
            asm volatile("" : "+s"(u));
            u += (unsigned long long)(long long)(256 * i - pad) * sizeof(double);
            asm volatile("" : "+s"(u));
            r[i] = __builtin_nontemporal_load((gptr<double>)u + t);
        }
    }
}

// arrived series -> provisional fp32 copy + this thread's shifted fp64 partial sums
__device__ __forceinline__ void reduce_series(const double (&r)[16], const double k0, float (&o)[16], double &s1, double &s2)
{
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const double d = r[i] - k0;
        s1 += d;
        s2 = fma(d, d, s2);
        o[i] = (float)d;
    }
}

} // namespace scr



// ---------------------------------------------------------------------------------------------
// The screening PASS of the filter-and-refine Run (capi_run.hip: run_select): fp32 only, no
// re-evaluation.  Per series it writes
//   mv[row]   sigma times the fp32 estimate of the signed score at the fp32 argmax (the fp32 value with its exact
//             power-of-two scale, as a double) and scr_var[row] = sigma^2: score estimate = mv / sqrt(var), with
//             |estimate - exact| <= E at every lag (E: the caller's bound, docs/HISTORY.md 4.6), and
//   flags[row]: what the lags whose fp32 |cc| lies within `screen_delta` = 2 E (scaled units) of the fp32 maximum -- the
//             only lags that can be the exact argmax -- look like:
//             SCR_IN / SCR_OUT: one of them has |lag| <= / > max_lag;  SCR_POS / SCR_NEG: its value is > 0 / < 0
//             (only computed when the Run's filters look at signs: scr_need_sign);
//             SCR_REFINE: fp32 is not trusted for this series (sigma outside 2^+-100, x[0] a far outlier);
//             SCR_NAN: the exact result is NaN (NaN / Inf samples);  sigma == 0 rows report score 0 at lag 0.
//   (lag[row] is not written: the selection never uses the fp32 argmax, and re-evaluated rows get their exact lag.)
// The caller turns these into a pessimistic and an optimistic selection key per row, refines (fp64 kernel) every
// row whose optimistic key reaches the N-th best pessimistic key, and selects among the refined rows only.
// Registers: <= 168 (three workgroups of 256 per CU); LDS 37 KB.  Global loads per pair: the batch's 16 spectrum
// factors per thread (L2), issued while no HBM load is in flight (vmcnt is in-order), and the next pair's rows, one
// series at a time (see the kernel).
namespace scr {

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// forward fp32 FFT; pass-1 factors W_4096^(k t) from four per-thread base powers W^t, W^2t, W^4t, W^8t
// (every factor is a product of at most four correctly rounded table entries);
// `mid` runs between the first transpose and the second pass, `late` between the second transpose and the last
// butterflies (the pass kernel reduces the next pair's first series and requests its second one there)
template <bool MULXC, typename F = NoHook, typename L = NoHook>
__device__ __forceinline__ void fft4096b(f2 (&v)[16], f2 *xbuf, const f2 *tw2s, const f2 w1, const f2 w2, const f2 w4,
                                         const f2 w8, const f2 (&xq)[16], const int t, F mid = F(), L late = L())
{
    dft16f(v);
    {
        const f2 w3 = cmulf(w1, w2), w5 = cmulf(w4, w1), w6 = cmulf(w4, w2), w7 = cmulf(w4, w3);
        v[P16(1)] = cmulf(v[P16(1)], w1);
        v[P16(2)] = cmulf(v[P16(2)], w2);
        v[P16(3)] = cmulf(v[P16(3)], w3);
        v[P16(4)] = cmulf(v[P16(4)], w4);
        v[P16(5)] = cmulf(v[P16(5)], w5);
        v[P16(6)] = cmulf(v[P16(6)], w6);
        v[P16(7)] = cmulf(v[P16(7)], w7);
        v[P16(8)] = cmulf(v[P16(8)], w8);
        v[P16(9)] = cmulf(v[P16(9)], cmulf(w8, w1));
        v[P16(10)] = cmulf(v[P16(10)], cmulf(w8, w2));
        v[P16(11)] = cmulf(v[P16(11)], cmulf(w8, w3));
        v[P16(12)] = cmulf(v[P16(12)], cmulf(w8, w4));
        v[P16(13)] = cmulf(v[P16(13)], cmulf(w8, w5));
        v[P16(14)] = cmulf(v[P16(14)], cmulf(w8, w6));
        v[P16(15)] = cmulf(v[P16(15)], cmulf(w8, w7));
    }
    exchange<false>(v, xbuf, t);
    fence();
    mid();
    fence();
    dft16f(v);
    {
        const int lo = t & 15;
#pragma unroll
        for (int k = 1; k < 16; k++)
            v[P16(k)] = cmulf(v[P16(k)], tw2s[k * 16 + lo]);
    }
    exchange<true>(v, xbuf, t);
    fence();
    late();
    fence();
    dft16f(v);
    f2 w[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        w[k] = v[P16(k)];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = MULXC ? cmulf(w[k], xq[k]) : w[k];
}

} // namespace scr

// NARROW: 0 <= MaxLag < 256 (only a thread's elements 0 and 15 can be lags inside MaxLag); SIGN: the Run's filters look
// at the sign of a score (SCR_POS / SCR_NEG are computed).  Both are known to the launcher.
template <int WPC, bool TIMING = false, bool PADDED = false, bool NARROW = true, bool SIGN = true>
__global__ __launch_bounds__(SCR_THREADS, WPC) void xcorr_screen_pass_n4096(const FusedParams p)
{
    using namespace scr;
    __shared__ f2 xbuf[SCR_XBUF];
    __shared__ f2 tw2s[256];
    __shared__ double red[32]; // [0,16) statistics; [16,32): 32 floats, the end phase's per-wave maxima
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pad = PADDED ? 4096 - p.N : 0; // PADDED: 2048 < N < 4096, leading zeros (xcorr.go:176-181)
    const double invN = PADDED ? 1.0 / (double)p.N : 1.0 / 4096.0, invNm1 = PADDED ? 1.0 / (double)(p.N - 1) : 1.0 / 4095.0;
    const float window = (float)p.screen_delta;
    const int max_lag = p.scr_max_lag;
    float *redf = reinterpret_cast<float *>(red + 16);
    f2 w1, w2, w4, w8;
    {
        const float2 tw = p.tw2f[t];
        tw2s[t] = mk2(tw.x, tw.y);
        const gptr<float2> tp = scalar_ptr(p.tw1f);
        w1 = ldg_f2(tp, 256 + t);
        w2 = ldg_f2(tp, 512 + t);
        w4 = ldg_f2(tp, 1024 + t);
        w8 = ldg_f2(tp, 2048 + t);
    }
    __syncthreads();
    PhaseClock<TIMING> clk;
    clk.start();
    long long pair = blockIdx.x; // the launcher never starts more workgroups than pairs
    // The next pair's rows come in ONE SERIES AT A TIME through a single 32-register fp64 buffer: series A is requested
    // behind the first transform (after its spectrum factors: vmcnt is in-order) and reduced to fp32 + sums next to the
    // second transform's last butterflies, where series B is requested; B is reduced at the top of the next iteration.
    // (Both series at once cost 64 registers: 8.5 ms against 8.0, tools/ablate/screen_w4.hip.)
    double raw[16], k0, sA1, sA2;
    float na[16];
    auto issue = [&](const double *row) {
        if (PADDED)
            issue_series_padded(raw, k0, row, t, pad);
        else
            issue_series(raw, k0, row, t);
    };
    issue(p.rows + 2 * pair * p.stride);
    fence();
    reduce_series(raw, k0, na, sA1, sA2);
    fence();
    issue(p.rows + (2 * pair + 1 < p.M ? 2 * pair + 1 : 2 * pair) * p.stride);
    fence();
    for (; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        long long nxt = pair + gridDim.x; // clamped: the prefetch is unconditional
        nxt = nxt < p.npairs ? nxt : p.npairs - 1;
        const long long nA = 2 * nxt, nB = (nA + 1 < p.M) ? nA + 1 : nA;
        if (TIMING)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        clk.template stamp<0>();
        // ---- series B arrives: shifted fp64 sums + provisional fp32 copy; the fp64 samples are dropped here
        float nb[16];
        double q[4];
        q[0] = sA1;
        q[1] = sA2;
        reduce_series(raw, k0, nb, q[2], q[3]);
        fence();
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
        for (int k = 0; k < 4; k++)
            q[k] = uniform((red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]));
        const double mA = uniform(q[0] * invN), mB = uniform(q[2] * invN); // mean of d
        const double varA = uniform((q[1] - q[0] * q[0] * invN) * invNm1);
        const double varB = uniform((q[3] - q[2] * q[2] * invN) * invNm1);
        const bool nanA = !__builtin_isfinite(varA), nanB = !__builtin_isfinite(varB);
        const bool zeroA = !nanA && !(varA > 0.0), zeroB = !nanB && !(varB > 0.0);
        const int eA = (int)((__double_as_longlong(varA) >> 52) & 0x7ff) - 1023;
        const int eB = (int)((__double_as_longlong(varB) >> 52) & 0x7ff) - 1023;
        // fp32 is trusted only when sigma is well inside its range and x[0] is no far outlier (|mean d| <= 8 sigma
        // keeps the rounding of fl32(d) at 2^-24 * O(sigma)); otherwise the row is flagged for the fp64 kernel
        const bool redoA = !(zeroA || nanA) && (eA > 200 || eA < -200 || mA * mA > 64.0 * varA);
        const bool redoB = !(zeroB || nanB) && (eB > 200 || eB < -200 || mB * mB > 64.0 * varB);
        const bool offA = zeroA || nanA || redoA, offB = zeroB || nanB || redoB || !hasB;
        const float sclA = offA ? 0.f : __int_as_float((127 - (eA >> 1)) << 23);
        const float sclB = offB ? 0.f : __int_as_float((127 - (eB >> 1)) << 23);
        const float mAf = offA ? 0.f : (float)mA, mBf = offB ? 0.f : (float)mB;
        f2 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            v[i] = mk2((na[i] - mAf) * sclA, (nb[i] - mBf) * sclB);
            if (PADDED && i < 8 && t + 256 * i < pad) // the pad stays zero: only samples are centred
                v[i] = mk2(0.f, 0.f);
        }
        if (offA || offB) { // block-uniform, rare: such a series contributes exact zeros (its samples may be NaN / Inf)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v[i].x = offA ? 0.f : v[i].x;
                v[i].y = offB ? 0.f : v[i].y;
            }
        }
        clk.template stamp<1>();
        // the batch's spectrum factors for the first transform's last pass, while no HBM load is in flight
        fence();
        f2 xq[16];
        {
            const Tw1FetchF fetch{p.xcf, t};
#pragma unroll
            for (int k = 0; k < 16; k++)
                xq[k] = fetch(k);
        }
        fence();
        fft4096b<true>(v, xbuf, tw2s, w1, w2, w4, w8, xq, t);
        clk.template stamp<2>();
        // ---- the next pair streams in behind the second transform (no other global load until it is consumed)
        fence();
        issue(p.rows + nA * p.stride);
        fence();
        {
            const double *rowB = p.rows + nB * p.stride;
            fft4096b<false>(v, xbuf, tw2s, w1, w2, w4, w8, xq, t, NoHook(), [&]() {
                reduce_series(raw, k0, na, sA1, sA2);
                fence();
                issue(rowB);
            });
        }
        clk.template stamp<3>();
        // ---- what the possible exact argmaxes look like.  Only four facts per series are needed once the maximum M is
        // known -- is there a lag with |cc| >= M - window inside MaxLag / outside it / with cc > 0 / with cc < 0 -- so each
        // thread folds its 16 values into maxima BEFORE the barrier (|cc| over its lags inside and outside MaxLag, cc,
        // -cc), every wave reduces them (DPP), the partials meet in LDS behind ONE barrier and lanes 0 / 1 of wave 0 --
        // one series each -- finish the row: flags, estimate (the maximum with its exact power-of-two scale; the selection
        // divides by sigma), variance.  One writer per row: plain stores, no atomics.  A thread's lags inside MaxLag are
        // its elements k <= klo and k >= khi (lag index 256 k + t); the values themselves are dead across the barrier.
        {
            int to = t; // (opaque: hoisted out of the pair loop klo / khi would be carried through every transform)
            asm volatile("" : "+v"(to));
            const int klo = max_lag >= to ? (max_lag - to) >> 8 : -1;
            const int khi = (4096 - max_lag - to + 255) >> 8;
            float inA, outA, inB, outB, posA = 0.f, negA = 0.f, posB = 0.f, negB = 0.f;
            if (NARROW) {
                const float a0 = fabsf(v[0].x), a15 = fabsf(v[15].x), b0 = fabsf(v[0].y), b15 = fabsf(v[15].y);
                const bool i0 = klo >= 0, i15 = khi <= 15;
                inA = fmaxf(i0 ? a0 : -1.f, i15 ? a15 : -1.f);
                outA = fmaxf(i0 ? -1.f : a0, i15 ? -1.f : a15);
                inB = fmaxf(i0 ? b0 : -1.f, i15 ? b15 : -1.f);
                outB = fmaxf(i0 ? -1.f : b0, i15 ? -1.f : b15);
#pragma unroll
                for (int k = 1; k < 15; k++) {
                    outA = fmaxf(outA, fabsf(v[k].x));
                    outB = fmaxf(outB, fabsf(v[k].y));
                }
            } else {
                inA = outA = inB = outB = -1.f;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const bool in = k <= klo || k >= khi;
                    const float aa = fabsf(v[k].x), ab = fabsf(v[k].y);
                    inA = fmaxf(inA, in ? aa : -1.f);
                    outA = fmaxf(outA, in ? -1.f : aa);
                    inB = fmaxf(inB, in ? ab : -1.f);
                    outB = fmaxf(outB, in ? -1.f : ab);
                }
            }
            if (SIGN) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    posA = fmaxf(posA, v[k].x);
                    negA = fmaxf(negA, -v[k].x);
                    posB = fmaxf(posB, v[k].y);
                    negB = fmaxf(negB, -v[k].y);
                }
            }
            // (the DPP maximum is for non-negative values: -1 "no such lag" is shifted to 0)
            const float wiA = wave_max_f32_dpp(inA + 1.f), woA = wave_max_f32_dpp(outA + 1.f);
            const float wiB = wave_max_f32_dpp(inB + 1.f), woB = wave_max_f32_dpp(outB + 1.f);
            float wpA = 0.f, wnA = 0.f, wpB = 0.f, wnB = 0.f;
            if (SIGN) {
                wpA = wave_max_f32_dpp(posA);
                wnA = wave_max_f32_dpp(negA);
                wpB = wave_max_f32_dpp(posB);
                wnB = wave_max_f32_dpp(negB);
            }
            if (lane == 0) {
                redf[4 * wave + 0] = wiA;
                redf[4 * wave + 1] = woA;
                redf[16 + 4 * wave + 0] = wiB;
                redf[16 + 4 * wave + 1] = woB;
                if (SIGN) {
                    redf[4 * wave + 2] = wpA;
                    redf[4 * wave + 3] = wnA;
                    redf[16 + 4 * wave + 2] = wpB;
                    redf[16 + 4 * wave + 3] = wnB;
                }
            }
            lds_barrier();
            if (t < 2 && (t == 0 || hasB)) {
                const float *r = redf + 16 * t;
                const float in = fmaxf(fmaxf(r[0], r[4]), fmaxf(r[8], r[12])) - 1.f, out = fmaxf(fmaxf(r[1], r[5]), fmaxf(r[9], r[13])) - 1.f;
                const float M = fmaxf(in, out), th = M - window;
                unsigned f = (in >= th ? SCR_IN : 0u) | (out >= th ? SCR_OUT : 0u);
                bool positive = true;
                if (SIGN) { // (cc > 0 / < 0 at a lag inside the window: its |cc| is cc or -cc itself)
                    const float pos = fmaxf(fmaxf(r[2], r[6]), fmaxf(r[10], r[14])), neg = fmaxf(fmaxf(r[3], r[7]), fmaxf(r[11], r[15]));
                    f |= ((pos >= th && pos > 0.f) ? SCR_POS : 0u) | ((neg >= th && neg > 0.f) ? SCR_NEG : 0u);
                    positive = pos == M;
                } else {
                    f |= SCR_POS | SCR_NEG; // not looked at
                }
                const int e = t ? eB : eA;
                double est = (double)(positive ? M : -M) * __longlong_as_double((long long)(1023 + (e >> 1)) << 52);
                if (t ? offB : offA) { // the rows fp32 has nothing to say about; sigma == 0: score 0 at lag 0, exactly
                    const bool nan = t ? nanB : nanA;
                    est = nan ? __builtin_nan("") : 0.0;
                    f = nan ? SCR_NAN : ((t ? redoB : redoA) ? SCR_REFINE : SCR_IN);
                }
                p.mv[rA + t] = est;
                p.scr_flags[rA + t] = f;
                p.scr_var[rA + t] = t ? varB : varA;
            }
        }
        clk.template stamp<4>();
    }
    if (TIMING && p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < NPHASE; i++)
            p.dbg[((long long)blockIdx.x * 4 + wave) * NPHASE + i] = clk.acc[i];
    }
}

// ---------------------------------------------------------------------------------------------
// The screening pass for R references that share one group (muse_batch_run_many): every pair of series is read
// from HBM, reduced and forward-transformed ONCE; its fp32 spectrum Z is parked in this workgroup's 32 KB slice of
// FusedParams::zscratch (each thread re-reads only what it wrote itself; L2 resident) and, per reference, multiplied by
// that reference's spectrum factors, transformed back and reported into that reference's arrays (xcf_many / mv_many /
// lag_many / flags_many / var_many: device tables of R pointers).  The next pair's rows are requested during the LAST
// reference's transform only, behind its table loads (in-order vmcnt).  N == n == 4096.
namespace scr {

// a wave-uniform pointer fetched from a device table, forced into SGPRs (the compiler cannot prove uniformity)
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *ptr)
{
    const unsigned long long u = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

struct PairStats {
    double varA, varB;
    int eA, eB;
    bool nanA, nanB, redoA, redoB, offA, offB, hasB;
};

// maxima + flags + estimates of one (pair, reference): the end phase of xcorr_screen_pass_n4096
__device__ __forceinline__ void report_pair(const f2 (&v)[16], const PairStats &st, double *mv, int *lag, unsigned *flags,
                                            double *var, const long long rA, const long long rB, const float window,
                                            const int max_lag, float *redf, const int t, const int lane, const int wave)
{
    float ma = 0.f, mb = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        ma = fmaxf(ma, fabsf(v[k].x));
        mb = fmaxf(mb, fabsf(v[k].y));
    }
    ma = wave_max_f32_dpp(ma);
    mb = wave_max_f32_dpp(mb);
    if (lane == 0) {
        redf[wave] = ma;
        redf[4 + wave] = mb;
    }
    lds_barrier();
    const float MA = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    const float MB = fmaxf(fmaxf(redf[4], redf[5]), fmaxf(redf[6], redf[7]));
    const float thA = MA - window, thB = MB - window;
    unsigned fA = 0u, fB = 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const float xa = v[k].x, xb = v[k].y;
        const bool ha = fabsf(xa) >= thA, hb = fabsf(xb) >= thB;
        if (__ballot(ha || hb) != 0ull) { // wave-uniform
            const int idx = 256 * k + t;
            const int lg = idx > 2048 ? idx - 4096 : idx;
            const unsigned in = (lg < 0 ? -lg : lg) <= max_lag ? SCR_IN : SCR_OUT;
            if (ha)
                fA |= in | (xa > 0.f ? SCR_POS : 0u) | (xa < 0.f ? SCR_NEG : 0u);
            if (hb)
                fB |= in | (xb > 0.f ? SCR_POS : 0u) | (xb < 0.f ? SCR_NEG : 0u);
            if (fabsf(xa) == MA && !st.offA) {
                mv[rA] = (double)xa * __longlong_as_double((long long)(1023 + (st.eA >> 1)) << 52);
                lag[rA] = lg;
            }
            if (fabsf(xb) == MB && !st.offB) {
                mv[rB] = (double)xb * __longlong_as_double((long long)(1023 + (st.eB >> 1)) << 52);
                lag[rB] = lg;
            }
        }
    }
    if (__ballot((fA | fB) != 0u) != 0ull) {
        unsigned wA = 0u, wB = 0u;
#pragma unroll
        for (unsigned bit = 1u; bit <= 8u; bit <<= 1) {
            wA |= __ballot((fA & bit) != 0u) != 0ull ? bit : 0u;
            wB |= __ballot((fB & bit) != 0u) != 0ull ? bit : 0u;
        }
        if (lane == 0) {
            if (wA && !st.offA)
                atomicOr(&flags[rA], wA);
            if (wB && !st.offB)
                atomicOr(&flags[rB], wB);
        }
    }
    if (t == 0) {
        var[rA] = st.varA;
        if (st.offA) {
            mv[rA] = st.nanA ? __builtin_nan("") : 0.0;
            lag[rA] = 0;
            atomicOr(&flags[rA], st.nanA ? SCR_NAN : (st.redoA ? SCR_REFINE : SCR_IN));
        }
        if (st.hasB) {
            var[rB] = st.varB;
            if (st.offB) {
                mv[rB] = st.nanB ? __builtin_nan("") : 0.0;
                lag[rB] = 0;
                atomicOr(&flags[rB], st.nanB ? SCR_NAN : (st.redoB ? SCR_REFINE : SCR_IN));
            }
        }
    }
}

} // namespace scr

template <int WPC>
__global__ __launch_bounds__(SCR_THREADS, WPC) void xcorr_screen_pass_many_n4096(const FusedParams p)
{
    using namespace scr;
    __shared__ f2 xbuf[SCR_XBUF];
    __shared__ f2 tw2s[256];
    __shared__ double red[24];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr double invN = 1.0 / 4096.0, invNm1 = 1.0 / 4095.0;
    const float window = (float)p.screen_delta;
    const int max_lag = p.scr_max_lag;
    const int R = p.R;
    float *redf = reinterpret_cast<float *>(red + 16);
    f2 *const zs = reinterpret_cast<f2 *>(p.zscratch) + (size_t)blockIdx.x * 4096; // resident grid: one slice per workgroup
    f2 w1, w2, w4, w8;
    {
        const float2 tw = p.tw2f[t];
        tw2s[t] = mk2(tw.x, tw.y);
        const gptr<float2> tp = scalar_ptr(p.tw1f);
        w1 = ldg_f2(tp, 256 + t);
        w2 = ldg_f2(tp, 512 + t);
        w4 = ldg_f2(tp, 1024 + t);
        w8 = ldg_f2(tp, 2048 + t);
    }
    __syncthreads();
    long long pair = blockIdx.x;
    double ra[16], rb[16], kA, kB;
    issue_series(ra, kA, p.rows + 2 * pair * p.stride, t);
    issue_series(rb, kB, p.rows + (2 * pair + 1 < p.M ? 2 * pair + 1 : 2 * pair) * p.stride, t);
    for (; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        PairStats st;
        st.hasB = rB < p.M;
        long long nxt = pair + gridDim.x;
        nxt = nxt < p.npairs ? nxt : p.npairs - 1;
        const long long nA = 2 * nxt, nB = (nA + 1 < p.M) ? nA + 1 : nA;
        float na[16], nb[16];
        double q[4];
        reduce_series(ra, kA, na, q[0], q[1]);
        reduce_series(rb, kB, nb, q[2], q[3]);
        fence();
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
        for (int k = 0; k < 4; k++)
            q[k] = uniform((red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]));
        const double mA = uniform(q[0] * invN), mB = uniform(q[2] * invN);
        st.varA = uniform((q[1] - q[0] * q[0] * invN) * invNm1);
        st.varB = uniform((q[3] - q[2] * q[2] * invN) * invNm1);
        st.nanA = !__builtin_isfinite(st.varA);
        st.nanB = !__builtin_isfinite(st.varB);
        const bool zeroA = !st.nanA && !(st.varA > 0.0), zeroB = !st.nanB && !(st.varB > 0.0);
        st.eA = (int)((__double_as_longlong(st.varA) >> 52) & 0x7ff) - 1023;
        st.eB = (int)((__double_as_longlong(st.varB) >> 52) & 0x7ff) - 1023;
        st.redoA = !(zeroA || st.nanA) && (st.eA > 200 || st.eA < -200 || mA * mA > 64.0 * st.varA);
        st.redoB = !(zeroB || st.nanB) && (st.eB > 200 || st.eB < -200 || mB * mB > 64.0 * st.varB);
        st.offA = zeroA || st.nanA || st.redoA;
        st.offB = zeroB || st.nanB || st.redoB || !st.hasB;
        const float sclA = st.offA ? 0.f : __int_as_float((127 - (st.eA >> 1)) << 23);
        const float sclB = st.offB ? 0.f : __int_as_float((127 - (st.eB >> 1)) << 23);
        const float mAf = st.offA ? 0.f : (float)mA, mBf = st.offB ? 0.f : (float)mB;
        f2 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = mk2((na[i] - mAf) * sclA, (nb[i] - mBf) * sclB);
        if (st.offA || st.offB) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v[i].x = st.offA ? 0.f : v[i].x;
                v[i].y = st.offB ? 0.f : v[i].y;
            }
        }
        // ---- the pair's spectrum, once: v[k] = Z[256 k + t]; parked for references 1 .. R-1
        {
            f2 none0[16];
            fft4096b<false>(v, xbuf, tw2s, w1, w2, w4, w8, none0, t);
        }
        if (R > 1) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                unsigned long long u = (unsigned long long)zs;
                asm volatile("" : "+s"(u));
                u += (unsigned long long)(256 * k) * sizeof(f2);
                asm volatile("" : "+s"(u));
                f2v zz;
                zz.x = v[k].x;
                zz.y = v[k].y;
                reinterpret_cast<f2v __attribute__((address_space(1))) *>(u)[t] = zz;
            }
        }
        // per reference: table loads first (L2: this reference's spectrum factors and, from the second reference on, the
        // parked Z), multiply, transform back, report.  The last reference is peeled: only there is the next pair's row
        // prefetch in flight (its registers must not be live across the loop).
        auto load_product = [&](const int r) {
            f2 xq[16];
            {
                const Tw1FetchF fetch{uniform_ptr(p.xcf_many[r]), t};
#pragma unroll
                for (int k = 0; k < 16; k++)
                    xq[k] = fetch(k);
            }
            if (r > 0) {
                const Tw1FetchF fz{reinterpret_cast<const float2 *>(zs), t};
#pragma unroll
                for (int k = 0; k < 16; k++)
                    v[k] = fz(k);
            }
#pragma unroll
            for (int k = 0; k < 16; k++)
                v[k] = cmulf(v[k], xq[k]);
            fence();
        };
        f2 none[16];
        for (int r = 0; r < R - 1; r++) {
            load_product(r);
            fft4096b<false>(v, xbuf, tw2s, w1, w2, w4, w8, none, t);
            report_pair(v, st, uniform_ptr(p.mv_many[r]), uniform_ptr(p.lag_many[r]), uniform_ptr(p.flags_many[r]),
                        uniform_ptr(p.var_many[r]), rA, rB, window, max_lag, redf, t, lane, wave);
        }
        {
            const int r = R - 1;
            load_product(r);
            issue_series(ra, kA, p.rows + nA * p.stride, t); // the next pair streams in behind the last transform
            fence();
            const double *rowB = p.rows + nB * p.stride;
            fft4096b<false>(v, xbuf, tw2s, w1, w2, w4, w8, none, t, [&]() { issue_series(rb, kB, rowB, t); });
            report_pair(v, st, uniform_ptr(p.mv_many[r]), uniform_ptr(p.lag_many[r]), uniform_ptr(p.flags_many[r]),
                        uniform_ptr(p.var_many[r]), rA, rB, window, max_lag, redf, t, lane, wave);
        }
    }
}

hipError_t launch_screen_pass_many(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (p.n != 4096 || p.N != 4096 || p.R < 1 || !p.xcf_many || !p.flags_many || !p.var_many || !p.mv_many || !p.lag_many ||
        !p.zscratch)
        return hipErrorInvalidValue;
    long long grid = p.npairs;
    long long cap = (long long)num_cus * 3;
    if (cap > p.zslots * 2) // (a slice of zscratch is 4096 double2 = two fp32 slices)
        cap = p.zslots * 2;
    if (grid > cap)
        grid = cap;
    hipLaunchKernelGGL((xcorr_screen_pass_many_n4096<3>), dim3((unsigned)grid), dim3(SCR_THREADS), 0, stream, p);
    return hipGetLastError();
}

template <bool PADDED, bool NARROW, bool SIGN>
static void launch_pass_variant(const FusedParams &p, unsigned grid, hipStream_t stream)
{
    hipLaunchKernelGGL((xcorr_screen_pass_n4096<3, false, PADDED, NARROW, SIGN>), dim3(grid), dim3(SCR_THREADS), 0, stream, p);
}

hipError_t launch_screen_pass(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (p.n != 4096 || p.N <= 2048 || p.N > 4096 || !p.scr_flags || !p.scr_var || !p.xcf)
        return hipErrorInvalidValue;
    long long grid = p.npairs;
    // three resident workgroups per CU (168 VGPRs).  A fourth fits the data flow (tools/ablate/screen_w4.hip: 128 VGPRs)
    // but the end phase then spills, and a spilled register reloaded behind the row prefetch costs an HBM latency.
    const long long cap = (long long)num_cus * 3;
    if (grid > cap)
        grid = cap;
    const bool padded = p.N < 4096, narrow = p.scr_max_lag >= 0 && p.scr_max_lag < 256, sign = p.scr_need_sign != 0;
    const unsigned g = (unsigned)grid;
    if (padded)
        narrow ? (sign ? launch_pass_variant<true, true, true>(p, g, stream) : launch_pass_variant<true, true, false>(p, g, stream))
               : (sign ? launch_pass_variant<true, false, true>(p, g, stream) : launch_pass_variant<true, false, false>(p, g, stream));
    else
        narrow ? (sign ? launch_pass_variant<false, true, true>(p, g, stream) : launch_pass_variant<false, true, false>(p, g, stream))
               : (sign ? launch_pass_variant<false, false, true>(p, g, stream) : launch_pass_variant<false, false, false>(p, g, stream));
    return hipGetLastError();
}

} // namespace muse
