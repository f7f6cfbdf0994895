// long_device.h -- building blocks of the long-series kernel (xcorr_long.hip): the in-register radix-R1 butterflies of the
// sweeps and the n = 4096 kernel's pair of transforms applied to one 4096-point row.
#pragma once
#include <hip/hip_runtime.h>

#include "xcorr_kernels.h"
#include "fft_device.h"
#include "foldk_device.h"

namespace muse {

namespace lng {

using namespace occ4;
using namespace fold;
using namespace foldk;

template <int R>
__device__ __forceinline__ constexpr int brev(int k)
{
    int r = 0;
    for (int b = 1; b < R; b <<= 1)
        r = (r << 1) | ((k & b) ? 1 : 0);
    return r;
}

// radix-R DFT (forward sign) over the registers m + s Q1, s = 0 .. R-1 natural; output k at register m + brev<R>(k) Q1
template <int R>
__device__ __forceinline__ void sweep_dft(double2 (&v)[16])
{
    constexpr int Q1 = 16 / R;
    if (R == 16) {
        dft16_nr(v); // (148 instructions; output k at v[BR16(k)])
        return;
    }
#pragma unroll
    for (int len = R; len >= 2; len >>= 1) {
        const int half = len / 2;
#pragma unroll
        for (int base = 0; base < R; base += len) {
#pragma unroll
            for (int k = 0; k < half; k++) {
                // W_len^k = c - i s (constants after unrolling)
                const double c = __builtin_cos(6.283185307179586476925 * (double)k / (double)len);
                const double s = __builtin_sin(6.283185307179586476925 * (double)k / (double)len);
#pragma unroll
                for (int m = 0; m < Q1; m++) {
                    double2 &a = v[m + (base + k) * Q1], &b = v[m + (base + k + half) * Q1];
                    const double2 u = a, x = b;
                    a = make_double2(u.x + x.x, u.y + x.y);
                    const double dx = u.x - x.x, dy = u.y - x.y;
                    if (k == 0)
                        b = make_double2(dx, dy);
                    else if (4 * k == len)
                        b = make_double2(dy, -dx);
                    else
                        b = make_double2(fma(dy, s, dx * c), fma(-dx, s, dy * c));
                }
            }
        }
    }
}

__device__ __forceinline__ double2 csqr(const double2 a)
{
    return make_double2(fma(a.x, a.x, -a.y * a.y), (a.x + a.x) * a.y);
}
// The sweeps' factors W_n^(m2 k), k = 1 .. R-1, of one element from ONE table entry w = W_n^(m2): powers formed where they are
// consumed, every one a product of at most two of {w, w^2, w^3} and {w^4, w^8, w^12} (at most four multiplications deep:
// a few 1e-16 of rounding on a unit-modulus factor).  Round 3 loaded all R - 1 from an [R][4096] table: fifteen 16-byte L2 requests
// per thread and chunk at n = 65536, two batches of four in flight beside the chunk's 64 data registers -- 32-49 registers per lane
// ended up in scratch, and the 1 MB table competed with the slices for the XCD's L2.  f(k, w^k) is called for k = 1 .. R-1 in order.
template <int R, typename F>
__device__ __forceinline__ void twiddle_powers(const double2 w, F f)
{
    const double2 w2 = csqr(w), w3 = cmul(w2, w);
    f(1, w);
    f(2, w2);
    f(3, w3);
    if (R > 4) {
        const double2 w4 = csqr(w2);
        f(4, w4);
        f(5, cmul(w4, w));
        f(6, cmul(w4, w2));
        f(7, cmul(w4, w3));
        if (R > 8) {
            const double2 w8 = csqr(w4);
            f(8, w8);
            f(9, cmul(w8, w));
            f(10, cmul(w8, w2));
            f(11, cmul(w8, w3));
            const double2 w12 = cmul(w8, w4);
            f(12, w12);
            f(13, cmul(w12, w));
            f(14, cmul(w12, w2));
            f(15, cmul(w12, w3));
        }
    }
}

// the n = 4096 kernel's pair of transforms on one row (xcorr_r16_fold.hip, default scheduling): x[t + 256 i] at v[i] ->
// FFT, times the lane-ordered spectrum row `xrow`, FFT -> element t + 256 m at v[BR16(m)].  zero0: bin 0 is zeroed.
__device__ __forceinline__ void row_transforms(double2 (&v)[16], double2 *xbuf, double2 *xw, const double2 *g2s,
                                               const double2 *__restrict__ g3a, const double2 *__restrict__ g3b,
                                               const double2 *__restrict__ xrow, const int t, const int wave, const bool zero0)
{
    const auto xcl = [&](int j) __attribute__((always_inline)) {
        return ldg2(scalar_ptr_at(xrow, 256 * ((j + 1) & ~1)), t - 256 * (j & 1));
    };
    // ---- first transform: plain pass over a, generalised passes over b (delta = k1 / 16) and c (delta = (k1 + 16 k2) / 256)
    dft16_nr(v);
    exchange_cross<0, 1, true>(v, xbuf, wave, t);
    gdft16_nr(v, G2Fetch{g2s, t >> 4});
    exchange_local<1>(v, xw, t);
    gdft16_nr_l2(v, G3Fetch{g3a, t});
    if (zero0) {
        v[0].x = (t == 0) ? 0.0 : v[0].x;
        v[0].y = (t == 0) ? 0.0 : v[0].y;
    }
    // ---- second transform: plain pass with the spectrum factors folded into its first stage, then the two generalised ones
    xc_stage1(v, xcl);
    dft16_rn_s234(v);
    exchange_local<0>(v, xw, t);
    gdft16_nr(v, G2Fetch{g2s, t & 15});
    exchange_cross<1, 1>(v, xbuf, wave, t);
    gdft16_nr_l2(v, G3Fetch{g3b, t});
}

// the two halves on their own (many references in one pass, xcorr_fused_long<.., MULTI>): the row's spectrum is stored in
// REGISTER order (register r of thread t at element t + 256 r -- every thread re-reads only what it wrote) and taken from
// there once per reference
__device__ __forceinline__ void row_forward(double2 (&v)[16], double2 *xbuf, double2 *xw, const double2 *g2s,
                                            const double2 *__restrict__ g3a, const int t, const int wave, const bool zero0)
{
    dft16_nr(v);
    exchange_cross<0, 1, true>(v, xbuf, wave, t);
    gdft16_nr(v, G2Fetch{g2s, t >> 4});
    exchange_local<1>(v, xw, t);
    gdft16_nr_l2(v, G3Fetch{g3a, t});
    if (zero0) {
        v[0].x = (t == 0) ? 0.0 : v[0].x;
        v[0].y = (t == 0) ? 0.0 : v[0].y;
    }
}
__device__ __forceinline__ void row_second(double2 (&v)[16], double2 *xbuf, double2 *xw, const double2 *g2s,
                                           const double2 *__restrict__ g3b, const double2 *__restrict__ xrow, const int t,
                                           const int wave)
{
    const auto xcl = [&](int j) __attribute__((always_inline)) {
        return ldg2(scalar_ptr_at(xrow, 256 * ((j + 1) & ~1)), t - 256 * (j & 1));
    };
    xc_stage1(v, xcl);
    dft16_rn_s234(v);
    exchange_local<0>(v, xw, t);
    gdft16_nr(v, G2Fetch{g2s, t & 15});
    exchange_cross<1, 1>(v, xbuf, wave, t);
    gdft16_nr_l2(v, G3Fetch{g3b, t});
}

} // namespace lng

} // namespace muse
