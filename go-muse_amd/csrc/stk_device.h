// stk_device.h -- the radix-16 Stockham transform engine shared by xcorr_stockham.hip (xCorrWithX, every length from 512 up:
// test-hook variant 11 and the re-evaluation kernel of long series) and xcorr_two_sided.hip (the batched two-sided xCorr,
// /root/reference/xcorr.go:102-153): natural-order transforms through a padded LDS work buffer, folded arithmetic
// (fold_device.h).  Structure and layouts: xcorr_stockham.hip's header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fold_device.h"
#include "r16_device.h"

namespace muse {

namespace stk {

using namespace occ4;

// ------------------------------------------------------------ small DFTs (in place, natural order)
__device__ __forceinline__ void dft2(double2 &a, double2 &b)
{
    const double2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}
__device__ __forceinline__ void dft4(double2 &a, double2 &b, double2 &c, double2 &d)
{
    // X0 = a+b+c+d, X1 = a - i b - c + i d, X2 = a-b+c-d, X3 = a + i b - c - i d
    const double2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = make_double2(t1.x + t3.y, t1.y - t3.x);
    d = make_double2(t1.x - t3.y, t1.y + t3.x);
}
__device__ __forceinline__ void dft8(double2 &x0, double2 &x1, double2 &x2, double2 &x3, double2 &x4, double2 &x5,
                                     double2 &x6, double2 &x7)
{
    constexpr double H = 0.70710678118654752440;
    // radix-2 over the high input bit, twiddle W8^k on the odd half, two radix-4s
    double2 e0 = cadd(x0, x4), e1 = cadd(x1, x5), e2 = cadd(x2, x6), e3 = cadd(x3, x7);
    double2 o0 = csub(x0, x4), o1 = csub(x1, x5), o2 = csub(x2, x6), o3 = csub(x3, x7);
    o1 = make_double2((o1.x + o1.y) * H, (o1.y - o1.x) * H); // * W8^1
    o2 = make_double2(o2.y, -o2.x);                          // * W8^2 = -i
    o3 = make_double2((o3.y - o3.x) * H, -(o3.x + o3.y) * H); // * W8^3
    dft4(e0, e1, e2, e3); // X[0], X[2], X[4], X[6]
    dft4(o0, o1, o2, o3); // X[1], X[3], X[5], X[7]
    x0 = e0; x2 = e1; x4 = e2; x6 = e3;
    x1 = o0; x3 = o1; x5 = o2; x7 = o3;
}

// 16/R independent radix-R DFTs on the registers m + s*(16/R) (in place, natural order)
template <int R>
__device__ __forceinline__ void dft_small(double2 (&v)[16])
{
    if (R == 16) { // one radix-16 DFT, un-permuted to natural order (register renaming only)
        dft16(v);
        double2 w[16];
#pragma unroll
        for (int r = 0; r < 16; r++)
            w[r] = v[P16(r)];
#pragma unroll
        for (int r = 0; r < 16; r++)
            v[r] = w[r];
        return;
    }
    constexpr int Q = 16 / R;
#pragma unroll
    for (int m = 0; m < Q; m++) {
        if (R == 2)
            dft2(v[m], v[m + Q]);
        else if (R == 4)
            dft4(v[m], v[m + Q], v[m + 2 * Q], v[m + 3 * Q]);
        else if (R == 8)
            dft8(v[m], v[m + Q], v[m + 2 * Q], v[m + 3 * Q], v[m + 4 * Q], v[m + 5 * Q], v[m + 6 * Q], v[m + 7 * Q]);
    }
}

// w[s] = W_(16 Ns)^(s m), s = 1..15, from the W_65536 half-period table
__device__ __forceinline__ void tw_powers(double2 (&w)[16], const double2 *__restrict__ twm, int m, int ns16)
{
    const int i1 = m * (65536 / ns16);
    w[1] = twm[i1];
    w[2] = twm[2 * i1];
    w[4] = twm[4 * i1];
    w[8] = twm[8 * i1];
    w[3] = cmul(w[1], w[2]);
    w[5] = cmul(w[1], w[4]);
    w[6] = cmul(w[2], w[4]);
    w[7] = cmul(w[3], w[4]);
    w[9] = cmul(w[1], w[8]);
    w[10] = cmul(w[2], w[8]);
    w[11] = cmul(w[3], w[8]);
    w[12] = cmul(w[4], w[8]);
    w[13] = cmul(w[5], w[8]);
    w[14] = cmul(w[6], w[8]);
    w[15] = cmul(w[7], w[8]);
}

// forward radix-16 pass on natural-order registers: pre-twiddle, DFT; output r at v[P16(r)]
__device__ __forceinline__ void fwd16(double2 (&v)[16], const double2 *__restrict__ twm, int m, int ns16)
{
    if (ns16 > 16) { // Ns > 1
        double2 w[16];
        tw_powers(w, twm, m, ns16);
#pragma unroll
        for (int s = 1; s < 16; s++)
            v[s] = cmul(v[s], w[s]);
    }
    dft16(v);
}
// transposed radix-16 pass on natural-order registers: DFT, post-twiddle; output s at v[P16(s)]
__device__ __forceinline__ void trn16(double2 (&v)[16], const double2 *__restrict__ twm, int m, int ns16)
{
    dft16(v);
    if (ns16 > 16) {
        double2 w[16];
        tw_powers(w, twm, m, ns16);
#pragma unroll
        for (int s = 1; s < 16; s++)
            v[P16(s)] = cmul(v[P16(s)], w[s]);
    }
}

// sums / max / min over each 16-lane row (every lane of the row gets the result)
__device__ __forceinline__ double row_sum_dpp(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ double row_max_dpp(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    return v;
}
__device__ __forceinline__ int row_min_i_dpp(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
    return v;
}

__device__ __forceinline__ int padpos(int pos) { return pos + (pos >> 4); }

// forward radix-16 pass as a GENERALISED 16-point transform (fold_device.h): the pre-twiddles W_(16 Ns)^(s m),
// s = 0..15, are a geometric sequence, i.e. the pass is  X[k] = sum_s x[s] W_16^(s (k + m / Ns))  with the
// per-thread phase delta = m / Ns, and every twiddle multiplication folds into the butterflies' FMAs: 192 instructions
// and eight table entries instead of 15 complex multiplies + a plain DFT + eleven twiddle products (264) and four.
// Factors: w2 = W_(2Ns)^m, w4 = W_(4Ns)^m, w8 = W_(8Ns)^m, w8 W_8, w16 W_16^q = W_(16Ns)^(m + q Ns), all inside the
// half-period W_65536 table.  Natural in, output r at v[BR16(r)].
template <int NS>
__device__ __forceinline__ void fwd16g(double2 (&v)[16], const double2 *__restrict__ twm, int m)
{
    static_assert(NS >= 2 && 32768 % NS == 0, "Ns");
    constexpr int U = 4096 / NS; // index step of W_(16 Ns)
    fold::gdft16_nr(v, [&](int s) __attribute__((always_inline)) {
        const int idx = s == 0 ? 8 * U * m : s == 1 ? 4 * U * m : s == 2 ? 2 * U * m : s == 3 ? 2 * U * m + 8192
                                                                                               : U * m + 4096 * (s - 4);
        return twm[idx];
    });
}

// Forward transform of the n points held as v[i] = x[j + i S] by the S = n/16 threads of a pair (work buffer b,
// padded): radix-R1 pass, then NP - 1 generalised radix-16 passes; on return X[j + r S] sits at v[BR16(r)].
// XC: the input is multiplied by xcf(i) first (the reference spectrum, folded in front of the first pass).
// Must be called by every thread of the workgroup (barriers); ENTRY_SYNC: the buffer may still be read by others.
// LDS positions: padpos(x) = x + (x >> 4).  Every access below is written as ONE per-thread base plus a compile-time
// offset (S, S R1 and 16 Ns are multiples of 16, and r < R1 / m < Ns never carry into the next block of 16), so a pass
// costs two address registers instead of sixteen: left to itself the compiler hoists ~100 addresses out of the pair
// loop and the kernels spilled 33 .. 128 registers at 256 (round 1).
constexpr int padk(int x) { return x + (x >> 4); } // for compile-time multiples of 16 (or offsets that do not carry)

template <int LOGN, bool ENTRY_SYNC>
__device__ __forceinline__ void lds_forward(double2 (&v)[16], double2 *b, const double2 *__restrict__ twm, const int j_)
{
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;
    constexpr int NP = (LOGN + 3) / 4;
    constexpr int R1 = n >> (4 * (NP - 1));
    constexpr int Q1 = 16 / R1;
    int j = j_;
    asm volatile("" : "+v"(j)); // (addresses are derived here, per call, not hoisted)
    const int rbase = j + (j >> 4);                 // padpos(j + i S) = rbase + i padk(S)
    const int w1base = j * R1 + ((j * R1) >> 4);    // padpos((j + m S) R1 + r) = w1base + r + m padk(S R1)
    dft_small<R1>(v); // pass 1: Ns = 1, butterflies q = j + m S on registers m + s Q1
    if (ENTRY_SYNC)
        __syncthreads(); // every thread is past its last read of the buffer
#pragma unroll
    for (int m = 0; m < Q1; m++)
#pragma unroll
        for (int r = 0; r < R1; r++)
            b[w1base + r + m * padk(S * R1)] = v[m + r * Q1];
    __syncthreads();
    { // pass 2: Ns = R1
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = b[rbase + i * padk(S)];
        fwd16g<R1>(v, twm, j % R1);
    }
    if (NP >= 3) { // pass 3: Ns = 16 R1
        constexpr int Ns = R1;
        __syncthreads(); // every thread has read its inputs
        const int base = (j / Ns) * (16 * Ns) + (j % Ns);
        const int wbase = base + (base >> 4); // padpos(base + r Ns) = wbase + r Ns + (r Ns >> 4): (base & 15) + (r Ns & 15) < 16
#pragma unroll
        for (int r = 0; r < 16; r++)
            b[wbase + r * Ns + ((r * Ns) >> 4)] = v[BR16(r)];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = b[rbase + i * padk(S)];
        fwd16g<16 * R1>(v, twm, j % (16 * R1));
    }
    if (NP >= 4) { // pass 4: Ns = 256 R1
        constexpr int Ns = 16 * R1;
        __syncthreads();
        const int base = (j / Ns) * (16 * Ns) + (j % Ns);
        const int wbase = base + (base >> 4);
#pragma unroll
        for (int r = 0; r < 16; r++)
            b[wbase + r * padk(Ns)] = v[BR16(r)];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = b[rbase + i * padk(S)];
        fwd16g<256 * R1>(v, twm, j % (256 * R1));
    }
}

// The on-chip part shared by the LDS kernels and the four-step kernel's row stage: forward transform, multiply
// output X[j + r S] by xcf(r), forward transform again (the unnormalised DFT applied twice to Z conj(X)/n gives the
// correlation of the packed pair directly: xcorr_r16_fold.hip); on return v[i] = result[j + i S].  The first transform
// leaves X[j + r S] in the thread that owns j -- exactly the layout the second one starts from, so there is no
// reorder pass (round 1 ran the TRANSPOSED passes backwards for the same reason; post-twiddled passes cannot fold
// their multiplications into FMAs, pre-twiddled ones can).
template <int LOGN, typename XcF>
__device__ __forceinline__ void lds_transforms(double2 (&v)[16], double2 *b, const double2 *__restrict__ twm, const int j,
                                               XcF xcf)
{
    lds_forward<LOGN, false>(v, b, twm, j);
    {
        double2 w[16];
#pragma unroll
        for (int r = 0; r < 16; r++)
            w[r] = cmul(v[BR16(r)], xcf(r));
#pragma unroll
        for (int r = 0; r < 16; r++)
            v[r] = w[r];
    }
    lds_forward<LOGN, true>(v, b, twm, j);
    {
        double2 w[16];
#pragma unroll
        for (int r = 0; r < 16; r++)
            w[r] = v[BR16(r)];
#pragma unroll
        for (int r = 0; r < 16; r++)
            v[r] = w[r];
    }
}

} // namespace stk

} // namespace muse
