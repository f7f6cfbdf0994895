// xcorr_r16_split.hip -- second-generation tuned kernel for n = 4096.
//
// Same mathematics as xcorr_fused_n4096 (see xcorr_kernels.hip header; the
// reference path is xcorr.go:160-197), restructured around what the round-1
// profile of the first kernel showed (profiles/r01_v1_rocprof_summary.txt):
// VALU busy 35 %, waves parked 56 % of their life at s_waitcnt/s_barrier with
// only 2 waves per SIMD resident (69.6 KB LDS, 255 VGPRs).  Changes:
//   * the two LDS transposes of each FFT move real and imaginary parts in
//     separate rounds through one 34.8 KB buffer (ds_write_b64/ds_read_b64,
//     conflict-free layout below), so FOUR workgroups fit a CU (16 waves, 4
//     per SIMD) and each other's HBM / LDS / barrier latency is covered;
//   * VGPR budget 128 (launch bound 4 waves per SIMD);
//   * wave reductions by DPP (no ds_bpermute round trips), z-normalisation
//     statistics in ONE block reduction (shifted sums, shift = first sample:
//     cancellation bounded by N+1, see zn comment), argmax in ONE;
//   * row loads are nontemporal (read once), twiddle / spectrum tables stay in L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_device.h"
#include "xcorr_kernels.h"

namespace muse {

constexpr int SPL_THREADS = 256;
constexpr int SPL_LDS = 16 * 272; // doubles: 34,816 B

// One LDS transpose round of one component (re or im) for both exchanges.
// Positions are in 8-byte units.
//   exchange A: writer (b = hi, c = lo) element k1 -> 272*k1 + 16*b + c = 272*k1 + t
//               reader (k1 = hi, c = lo) element b  <- 272*hi + 16*b + lo
//   exchange B: writer (k1 = hi, c = lo) element k2 -> 272*k2 + 17*hi + lo
//               reader (k1 = lo, k2 = hi) element c <- 272*hi + 17*lo + c
// ds_write_b64 serves 16 consecutive lanes per cycle over 32 banks of 4 B:
// both writers put consecutive lanes on consecutive 8-byte slots.  ds_read_b64
// serves 32 lanes over 64 banks: slot mod 32 is (16*hi + lo) for A (row stride
// 272 = 16 mod 32) and (16*hi + 17*lo + c) for B -- both bijections of the 32
// lanes of a half-wave.

template <bool IM>
__device__ __forceinline__ void xa_write(double *lds, const double2 (&v)[16], int t)
{
#pragma unroll
    for (int k = 0; k < 16; k++)
        lds[272 * k + t] = IM ? v[P16(k)].y : v[P16(k)].x;
}
template <bool IM>
__device__ __forceinline__ void xa_read(const double *lds, double2 (&v)[16], int hi, int lo)
{
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const double e = lds[272 * hi + 16 * b + lo];
        if (IM)
            v[b].y = e;
        else
            v[b].x = e;
    }
}
template <bool IM>
__device__ __forceinline__ void xb_write(double *lds, const double2 (&v)[16], int hi, int lo)
{
#pragma unroll
    for (int k = 0; k < 16; k++)
        lds[272 * k + 17 * hi + lo] = IM ? v[P16(k)].y : v[P16(k)].x;
}
template <bool IM>
__device__ __forceinline__ void xb_read(const double *lds, double2 (&v)[16], int hi, int lo)
{
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const double e = lds[272 * hi + 17 * lo + c];
        if (IM)
            v[c].y = e;
        else
            v[c].x = e;
    }
}

// v[a] = x[t + 256 a]  ->  v[a] = X[t + 256 a]   (forward, 4096 points)
__device__ __forceinline__ void fft4096_split(double2 (&v)[16], double *lds, const double2 *__restrict__ tw1,
                                              const double2 *__restrict__ tw2, const int t)
{
    const int hi = t >> 4, lo = t & 15;
    // pass 1 (DFT over a) + twiddle W_4096^(k1 t)
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; k++)
        v[P16(k)] = cmul(v[P16(k)], tw1[k * 256 + t]);
    __syncthreads(); // buffer free (previous round's readers done)
    xa_write<false>(lds, v, t);
    __syncthreads();
    double2 w[16];
    xa_read<false>(lds, w, hi, lo);
    __syncthreads();
    xa_write<true>(lds, v, t);
    __syncthreads();
    xa_read<true>(lds, w, hi, lo);
    // pass 2 (DFT over b; k1 = hi, c = lo) + twiddle W_256^(k2 c)
    dft16(w);
#pragma unroll
    for (int k = 1; k < 16; k++)
        w[P16(k)] = cmul(w[P16(k)], tw2[k * 16 + lo]);
    __syncthreads();
    xb_write<false>(lds, w, hi, lo);
    __syncthreads();
    xb_read<false>(lds, v, hi, lo);
    __syncthreads();
    xb_write<true>(lds, w, hi, lo);
    __syncthreads();
    xb_read<true>(lds, v, hi, lo);
    // pass 3 (DFT over c; k1 = lo, k2 = hi): f = t + 256 k3
    dft16(v);
#pragma unroll
    for (int k = 0; k < 16; k++)
        w[k] = v[P16(k)];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = w[k];
}

template <int WPS>
__global__ __launch_bounds__(SPL_THREADS, WPS) void xcorr_fused_n4096_split(const FusedParams p)
{
    __shared__ double lds[SPL_LDS];
    __shared__ double red[48];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int N = p.N;
    const int pad = 4096 - N;
    const double invN = 1.0 / (double)N; // only used as a factor of O(1) corrections

    for (long long pair = blockIdx.x; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        const double *__restrict__ ra = p.rows + rA * p.stride;
        const double *__restrict__ rb = p.rows + (hasB ? rB : rA) * p.stride;

        // ---- coalesced nontemporal load: element t + 256*a of the zero-padded rows
        double2 v[16];
        const double KA = ra[0], KB = rb[0]; // shift for the one-pass statistics
#pragma unroll
        for (int a = 0; a < 16; a++) {
            const int j = t + 256 * a - pad;
            double xa = KA, xb = KB;
            if (j >= 0) {
                xa = __builtin_nontemporal_load(ra + j);
                xb = __builtin_nontemporal_load(rb + j);
            }
            v[a] = make_double2(xa - KA, xb - KB); // pads give exactly 0
        }
        // ---- zNormalize (xcorr.go:84-95) from ONE block reduction.
        // d = x - K with K = x[0]:  mean = K + S1/N,  (N-1) var = S2 - S1^2/N.
        // (m - K)^2 <= sum (x - m)^2 because K is a sample, so S2 <= (N+1) *
        // (N-1) var: cancellation amplifies rounding by at most ~N (4e-13 rel).
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < 16; a++) {
            q[0] += v[a].x;
            q[1] = fma(v[a].x, v[a].x, q[1]);
            q[2] += v[a].y;
            q[3] = fma(v[a].y, v[a].y, q[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            q[k] = wave_sum_dpp(q[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                red[wave * 4 + k] = q[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++)
            q[k] = (red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]);
        ZnFlags fa, fb;
        const double ia = zn_scale(q[0], q[1], N, fa);
        const double ib = zn_scale(q[2], q[3], N, fb);
        const bool deadA = fa.zero || fa.nan, deadB = fb.zero || fb.nan || !hasB;
        const double ma_ = q[0] * invN, mb_ = q[2] * invN; // mean of d
#pragma unroll
        for (int a = 0; a < 16; a++) {
            const bool valid = t + 256 * a - pad >= 0;
            v[a].x = (deadA || !valid) ? 0.0 : (v[a].x - ma_) * ia;
            v[a].y = (deadB || !valid) ? 0.0 : (v[a].y - mb_) * ib;
        }
        // ---- Z = FFT(yA + i yB);  V = Z * conj(X)/n;  ccA + i ccB = FFT(V)
        fft4096_split(v, lds, p.tw1, p.tw2, t);
#pragma unroll
        for (int k = 0; k < 16; k++)
            v[k] = cmul(v[k], p.xc[t + 256 * k]);
        fft4096_split(v, lds, p.tw1, p.tw2, t);

        // ---- maxAbsIndex (xcorr.go:39-50), index = t + 256*k, one barrier:
        // each wave publishes (max |cc|, lowest index attaining it, signed value).
        double ma = 0.0, mb = 0.0, sa = 0.0, sb = 0.0;
        int ka = 0, kb = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const double aa = fabs(v[k].x), ab = fabs(v[k].y);
            if (aa > ma) { ma = aa; sa = v[k].x; ka = k; }
            if (ab > mb) { mb = ab; sb = v[k].y; kb = k; }
        }
        const double wa = wave_max_dpp(ma), wb = wave_max_dpp(mb);
        const int ia_ = wave_min_i_dpp((ma == wa && wa > 0.0) ? (t + 256 * ka) : 0x7fffffff);
        const int ib_ = wave_min_i_dpp((mb == wb && wb > 0.0) ? (t + 256 * kb) : 0x7fffffff);
        // red[16 + 3*wave ..] series A, red[28 + 3*wave ..] series B: {|max|, signed, index}
        if (ia_ == 0x7fffffff) {
            if (lane == 0) {
                red[16 + 3 * wave] = 0.0;
                red[17 + 3 * wave] = (wave == 0) ? v[0].x : 0.0; // cc[0] lives in thread 0
                red[18 + 3 * wave] = (double)0x7fffffff;
            }
        } else if (t + 256 * ka == ia_ && ma == wa) {
            red[16 + 3 * wave] = wa;
            red[17 + 3 * wave] = sa;
            red[18 + 3 * wave] = (double)ia_;
        }
        if (ib_ == 0x7fffffff) {
            if (lane == 0) {
                red[28 + 3 * wave] = 0.0;
                red[29 + 3 * wave] = (wave == 0) ? v[0].y : 0.0;
                red[30 + 3 * wave] = (double)0x7fffffff;
            }
        } else if (t + 256 * kb == ib_ && mb == wb) {
            red[28 + 3 * wave] = wb;
            red[29 + 3 * wave] = sb;
            red[30 + 3 * wave] = (double)ib_;
        }
        __syncthreads();
        if (t < 2 && (t == 0 || hasB)) {
            const int base = t == 0 ? 16 : 28;
            double best = red[base], bsv = red[base + 1], bidx = red[base + 2];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const double m = red[base + 3 * w], s = red[base + 3 * w + 1], ix = red[base + 3 * w + 2];
                if (m > best || (m == best && ix < bidx)) {
                    best = m;
                    bsv = s;
                    bidx = ix;
                }
            }
            int idx = (best > 0.0) ? (int)bidx : 0; // nothing above 0 (or all NaN): index 0, mv = cc[0]
            double mv = (best > 0.0) ? bsv : red[base + 1];
            int lag = idx > 2048 ? idx - 4096 : idx;
            const ZnFlags f = t == 0 ? fa : fb;
            if (f.zero) { mv = 0.0; lag = 0; }
            if (f.nan) { mv = __builtin_nan(""); lag = 0; }
            const long long r = t == 0 ? rA : rB;
            p.mv[r] = mv;
            p.lag[r] = lag;
        }
        // red[16..] is rewritten only after the next pair's barriers
    }
}

hipError_t launch_fused_split(const FusedParams &p, int num_cus, int waves_per_simd, hipStream_t stream)
{
    long long grid = p.npairs;
    const long long cap = (long long)num_cus * waves_per_simd * 4; // resident workgroups per CU, x4 for tail balance
    if (grid > cap)
        grid = cap;
    if (waves_per_simd == 3)
        hipLaunchKernelGGL(xcorr_fused_n4096_split<3>, dim3((unsigned)grid), dim3(SPL_THREADS), 0, stream, p);
    else
        hipLaunchKernelGGL(xcorr_fused_n4096_split<4>, dim3((unsigned)grid), dim3(SPL_THREADS), 0, stream, p);
    return hipGetLastError();
}

} // namespace muse
