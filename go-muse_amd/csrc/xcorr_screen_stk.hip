// xcorr_screen_stk.hip -- the fp32 screening pass of the filter-and-refine Run (docs/HISTORY.md 4.6) for the FFT lengths
// n = 512, 1024, 2048 and 8192 (series of n/2 < N <= n samples, leading zero pad): the radix-16 Stockham structure of
// xcorr_stockham.hip (xcorr_fused_stk_lds) with 8-byte complex values -- half the LDS bytes and twice the resident
// workgroups of the fp64 kernel (three per CU at 168 VGPRs instead of two).  Same contract as xcorr_screen_pass_n4096 (xcorr_r16_screen.hip): per series an
// estimate (mv = sigma * estimate, scr_var = sigma^2), the fp32 argmax lag (informational) and the SCR_* flags of every
// lag inside the window of the fp32 maximum; fp64 shifted statistics; the rows the fp32 path is not trusted for are
// flagged SCR_REFINE.  Follows xcorr.go:160-197 only up to the argmax: the exact scores come from the fp64 kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "r16_device.h"

namespace muse {

namespace sstk {

using namespace occ4;

__device__ __forceinline__ void dft2f(f2 &a, f2 &b)
{
    const f2 t = a;
    a = caddf(t, b);
    b = csubf(t, b);
}
__device__ __forceinline__ void dft4f(f2 &a, f2 &b, f2 &c, f2 &d)
{
    const f2 t0 = caddf(a, c), t1 = csubf(a, c), t2 = caddf(b, d), t3 = csubf(b, d);
    a = caddf(t0, t2);
    c = csubf(t0, t2);
    b = mk2(t1.x + t3.y, t1.y - t3.x);
    d = mk2(t1.x - t3.y, t1.y + t3.x);
}
__device__ __forceinline__ void dft8f(f2 &x0, f2 &x1, f2 &x2, f2 &x3, f2 &x4, f2 &x5, f2 &x6, f2 &x7)
{
    constexpr float H = 0.70710678118654752440f;
    f2 e0 = caddf(x0, x4), e1 = caddf(x1, x5), e2 = caddf(x2, x6), e3 = caddf(x3, x7);
    f2 o0 = csubf(x0, x4), o1 = csubf(x1, x5), o2 = csubf(x2, x6), o3 = csubf(x3, x7);
    o1 = mk2((o1.x + o1.y) * H, (o1.y - o1.x) * H);  // * W8^1
    o2 = mk2(o2.y, -o2.x);                           // * W8^2 = -i
    o3 = mk2((o3.y - o3.x) * H, -(o3.x + o3.y) * H); // * W8^3
    dft4f(e0, e1, e2, e3);
    dft4f(o0, o1, o2, o3);
    x0 = e0; x2 = e1; x4 = e2; x6 = e3;
    x1 = o0; x3 = o1; x5 = o2; x7 = o3;
}
// 16/R independent radix-R DFTs on the registers m + s*(16/R) (in place, natural order)
template <int R>
__device__ __forceinline__ void dft_small_f(f2 (&v)[16])
{
    if (R == 16) { // one radix-16 DFT, un-permuted to natural order (register renaming only)
        dft16f(v);
        f2 w[16];
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
            dft2f(v[m], v[m + Q]);
        else if (R == 4)
            dft4f(v[m], v[m + Q], v[m + 2 * Q], v[m + 3 * Q]);
        else
            dft8f(v[m], v[m + Q], v[m + 2 * Q], v[m + 3 * Q], v[m + 4 * Q], v[m + 5 * Q], v[m + 6 * Q], v[m + 7 * Q]);
    }
}
// w[s] = W_(16 Ns)^(s m), s = 1..15, from the fp32 W_65536 half-period table (products of <= 4 rounded entries)
__device__ __forceinline__ void tw_powers_f(f2 (&w)[16], const float2 *__restrict__ twm, int m, int ns16)
{
    const int i1 = m * (65536 / ns16);
    const float2 a = twm[i1], b = twm[2 * i1], c = twm[4 * i1], d = twm[8 * i1];
    w[1] = mk2(a.x, a.y);
    w[2] = mk2(b.x, b.y);
    w[4] = mk2(c.x, c.y);
    w[8] = mk2(d.x, d.y);
    w[3] = cmulf(w[1], w[2]);
    w[5] = cmulf(w[1], w[4]);
    w[6] = cmulf(w[2], w[4]);
    w[7] = cmulf(w[3], w[4]);
    w[9] = cmulf(w[1], w[8]);
    w[10] = cmulf(w[2], w[8]);
    w[11] = cmulf(w[3], w[8]);
    w[12] = cmulf(w[4], w[8]);
    w[13] = cmulf(w[5], w[8]);
    w[14] = cmulf(w[6], w[8]);
    w[15] = cmulf(w[7], w[8]);
}
__device__ __forceinline__ void fwd16f(f2 (&v)[16], const float2 *__restrict__ twm, int m, int ns16)
{
    if (ns16 > 16) {
        f2 w[16];
        tw_powers_f(w, twm, m, ns16);
#pragma unroll
        for (int s = 1; s < 16; s++)
            v[s] = cmulf(v[s], w[s]);
    }
    dft16f(v);
}
__device__ __forceinline__ void trn16f(f2 (&v)[16], const float2 *__restrict__ twm, int m, int ns16)
{
    dft16f(v);
    if (ns16 > 16) {
        f2 w[16];
        tw_powers_f(w, twm, m, ns16);
#pragma unroll
        for (int s = 1; s < 16; s++)
            v[P16(s)] = cmulf(v[P16(s)], w[s]);
    }
}
__device__ __forceinline__ double row_sum_dpp(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ float row_max_f32_dpp(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ int padpos(int pos) { return pos + (pos >> 4); }

// forward transform of the n points held as v[i] = z[j + i S] by the S = n/16 threads of a pair (work buffer b, padded),
// multiply output Z[j + r S] by xcf(r), transposed transform; on return v[i] = cc[j + i S] (xcorr_stockham.hip:
// lds_transforms, fp32).  Must be called by every thread of the workgroup (barriers).
// (`j` is re-materialised through an empty asm before every pass: otherwise the compiler computes the ~100 LDS
// addresses of all passes once, outside the loop over the pairs, and keeps them in registers)
__device__ __forceinline__ int opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}
template <int LOGN, typename XcF>
__device__ __forceinline__ void lds_transforms_f(f2 (&v)[16], f2 *b, const float2 *__restrict__ twm, int j, XcF xcf)
{
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;
    constexpr int NP = (LOGN + 3) / 4;
    constexpr int R1 = n >> (4 * (NP - 1));
    constexpr int Q1 = 16 / R1;
    dft_small_f<R1>(v);
    j = opaque(j);
#pragma unroll
    for (int m = 0; m < Q1; m++)
#pragma unroll
        for (int r = 0; r < R1; r++)
            b[padpos((j + m * S) * R1 + r)] = v[m + r * Q1];
    __syncthreads();
#pragma unroll
    for (int pp = 2; pp <= NP; pp++) {
        const int Ns = R1 << (4 * (pp - 2));
        j = opaque(j);
        fence(); // (scheduling fences: without them the compiler hoists the next passes' loads over the butterflies
                 // and the kernel needs 250 registers)
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = b[padpos(j + i * S)];
        fence();
        fwd16f(v, twm, j % Ns, 16 * Ns);
        fence();
        if (pp < NP) {
            __syncthreads();
            const int base = (j / Ns) * (16 * Ns) + (j % Ns);
#pragma unroll
            for (int r = 0; r < 16; r++)
                b[padpos(base + r * Ns)] = v[P16(r)];
            __syncthreads();
        }
    }
    fence();
    j = opaque(j);
    {
        f2 w[16];
#pragma unroll
        for (int r = 0; r < 16; r++)
            w[r] = cmulf(v[P16(r)], xcf(j, r));
#pragma unroll
        for (int r = 0; r < 16; r++)
            v[r] = w[r];
    }
#pragma unroll
    for (int pp = NP; pp >= 2; pp--) {
        const int Ns = R1 << (4 * (pp - 2));
        j = opaque(j);
        fence();
        trn16f(v, twm, j % Ns, 16 * Ns);
        fence();
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; s++)
            b[padpos(j + s * S)] = v[P16(s)];
        __syncthreads();
        if (pp > 2) {
            const int Np = R1 << (4 * (pp - 3));
            const int base = (j / Np) * (16 * Np) + (j % Np);
#pragma unroll
            for (int r = 0; r < 16; r++)
                v[r] = b[padpos(base + r * Np)];
        }
    }
    j = opaque(j);
#pragma unroll
    for (int m = 0; m < Q1; m++)
#pragma unroll
        for (int r = 0; r < R1; r++)
            v[m + r * Q1] = b[padpos((j + m * S) * R1 + r)];
    dft_small_f<R1>(v);
}

} // namespace sstk

template <int LOGN>
// (second launch-bound: waves per SIMD -- three 256-thread workgroups per CU; the 512-thread build of n = 8192 needs
// 256 registers (166 spilled at 128): one workgroup per CU, as the fp64 kernel)
__global__ __launch_bounds__(((1 << LOGN) / 16 > 256 ? (1 << LOGN) / 16 : 256), ((1 << LOGN) / 16 > 256 ? 2 : 3))
void xcorr_screen_pass_stk(const FusedParams p)
{
    using namespace occ4;
    using namespace sstk;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;                 // threads per pair
    constexpr int G = S >= 256 ? 1 : 256 / S; // pairs per workgroup iteration
    constexpr int TPB = S * G;                // 256 (n <= 4096) or 512 (n = 8192)
    constexpr int ROWS = S / 16;              // 16-lane rows per pair
    constexpr int BUF = n + n / 16;
    static_assert((LOGN >= 9 && LOGN <= 11) || LOGN == 13, "fp32 Stockham screening pass: n = 512, 1024, 2048, 8192");
    __shared__ f2 buf[G * BUF];
    __shared__ double red[(TPB / 16) * 4];  // per 16-lane row: {sum dA, sum dA^2, sum dB, sum dB^2}
    __shared__ float rmax[(TPB / 16) * 2];  // per row: fp32 maxima of |cc| (A, B)
    const int t = threadIdx.x;
    const int g = t / S, j = t % S;
    const int row = t >> 4;
    f2 *const b = buf + g * BUF;
    const int N = p.N, pad = n - N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);
    const float2 *__restrict__ twm = p.twmf;
    const float window = (float)p.screen_delta;
    const int max_lag = p.scr_max_lag;
    const long long ngroups = (p.npairs + G - 1) / G;

    for (long long it = blockIdx.x; it < ngroups; it += gridDim.x) {
        const long long pair_raw = it * G + g;
        const bool live = pair_raw < p.npairs;
        const long long pair = live ? pair_raw : p.npairs - 1; // idle sub-groups shadow the last pair (and store nothing)
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        const double *__restrict__ ra = p.rows + rA * p.stride;
        const double *__restrict__ rb = p.rows + (hasB ? rB : rA) * p.stride;
        // ---- rows (leading zero pad): fp64 shifted statistics, provisional fp32 copy of d = x - x[0]
        float na[16], nb[16];
        const double KA = ra[0], KB = rb[0];
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        // (two batches of 16 loads: all 32 in flight at once would hold 64 data + 64 address registers)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double xa[8], xb[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int e = j + (8 * h + i) * S - pad;
                const int ec = e < 0 ? 0 : e; // a pad position loads x[0]: d = 0 without a mask
                xa[i] = __builtin_nontemporal_load(ra + ec);
                xb[i] = __builtin_nontemporal_load(rb + ec);
            }
            fence();
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double da = xa[i] - KA, db = xb[i] - KB;
                na[8 * h + i] = (float)da;
                nb[8 * h + i] = (float)db;
                q[0] += da;
                q[1] = fma(da, da, q[1]);
                q[2] += db;
                q[3] = fma(db, db, q[3]);
            }
            fence();
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            q[k] = row_sum_dpp(q[k]);
        if ((t & 15) == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                red[row * 4 + k] = q[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double s = 0.0;
            for (int r = 0; r < ROWS; r++)
                s += red[(g * ROWS + r) * 4 + k];
            q[k] = s;
        }
        const double mA = q[0] * invN, mB = q[2] * invN;
        const double varA = (q[1] - q[0] * q[0] * invN) * invNm1, varB = (q[3] - q[2] * q[2] * invN) * invNm1;
        const bool nanA = !__builtin_isfinite(varA), nanB = !__builtin_isfinite(varB);
        const bool zeroA = !nanA && !(varA > 0.0), zeroB = !nanB && !(varB > 0.0);
        const int eA = (int)((__double_as_longlong(varA) >> 52) & 0x7ff) - 1023;
        const int eB = (int)((__double_as_longlong(varB) >> 52) & 0x7ff) - 1023;
        const bool redoA = !(zeroA || nanA) && (eA > 200 || eA < -200 || mA * mA > 64.0 * varA);
        const bool redoB = !(zeroB || nanB) && (eB > 200 || eB < -200 || mB * mB > 64.0 * varB);
        const bool offA = zeroA || nanA || redoA, offB = zeroB || nanB || redoB || !hasB;
        const float sclA = offA ? 0.f : __int_as_float((127 - (eA >> 1)) << 23);
        const float sclB = offB ? 0.f : __int_as_float((127 - (eB >> 1)) << 23);
        const float mAf = offA ? 0.f : (float)mA, mBf = offB ? 0.f : (float)mB;
        f2 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool valid = j + i * S - pad >= 0; // the pad stays zero: only samples are centred
            v[i] = mk2((valid && !offA) ? (na[i] - mAf) * sclA : 0.f, (valid && !offB) ? (nb[i] - mBf) * sclB : 0.f);
        }
        fence();
        lds_transforms_f<LOGN>(v, b, twm, j, [&](int jj, int r) __attribute__((always_inline)) {
            const float2 x = p.xcf[jj + r * S];
            return mk2(x.x, x.y);
        });
        fence();
        // ---- fp32 maxima per series over the pair's threads
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            ma = fmaxf(ma, fabsf(v[i].x));
            mb = fmaxf(mb, fabsf(v[i].y));
        }
        ma = row_max_f32_dpp(ma);
        mb = row_max_f32_dpp(mb);
        if ((t & 15) == 0) {
            rmax[row * 2] = ma;
            rmax[row * 2 + 1] = mb;
        }
        __syncthreads();
        float MA = 0.f, MB = 0.f;
        for (int r = 0; r < ROWS; r++) {
            MA = fmaxf(MA, rmax[(g * ROWS + r) * 2]);
            MB = fmaxf(MB, rmax[(g * ROWS + r) * 2 + 1]);
        }
        // ---- every lag within the window of the maximum: flags; the maximum itself: the estimate
        if (live) {
            const float thA = MA - window, thB = MB - window;
            unsigned fA = 0u, fB = 0u;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float xa = v[i].x, xb = v[i].y;
                const bool ha = fabsf(xa) >= thA, hb = fabsf(xb) >= thB;
                if (ha || hb) { // rare: the argmax and at most a few neighbours
                    const int idx = j + i * S;
                    const int lg = idx > n / 2 ? idx - n : idx;
                    const unsigned in = (lg < 0 ? -lg : lg) <= max_lag ? SCR_IN : SCR_OUT;
                    if (ha)
                        fA |= in | (xa > 0.f ? SCR_POS : 0u) | (xa < 0.f ? SCR_NEG : 0u);
                    if (hb)
                        fB |= in | (xb > 0.f ? SCR_POS : 0u) | (xb < 0.f ? SCR_NEG : 0u);
                    if (ha && fabsf(xa) == MA && !offA) {
                        p.mv[rA] = (double)xa * __longlong_as_double((long long)(1023 + (eA >> 1)) << 52);
                        p.lag[rA] = lg;
                    }
                    if (hb && fabsf(xb) == MB && !offB) {
                        p.mv[rB] = (double)xb * __longlong_as_double((long long)(1023 + (eB >> 1)) << 52);
                        p.lag[rB] = lg;
                    }
                }
            }
            if (fA && !offA)
                atomicOr(&p.scr_flags[rA], fA);
            if (fB && !offB)
                atomicOr(&p.scr_flags[rB], fB);
            if (j == 0) {
                p.scr_var[rA] = varA;
                if (offA) {
                    p.mv[rA] = nanA ? __builtin_nan("") : 0.0;
                    p.lag[rA] = 0;
                    atomicOr(&p.scr_flags[rA], nanA ? SCR_NAN : (redoA ? SCR_REFINE : SCR_IN));
                }
                if (hasB) {
                    p.scr_var[rB] = varB;
                    if (offB) {
                        p.mv[rB] = nanB ? __builtin_nan("") : 0.0;
                        p.lag[rB] = 0;
                        atomicOr(&p.scr_flags[rB], nanB ? SCR_NAN : (redoB ? SCR_REFINE : SCR_IN));
                    }
                }
            }
        }
        __syncthreads(); // red / rmax / buf free for the next iteration
    }
}

// ---------------------------------------------------------------------------------------------
// n = 16384, 32768, 65536 (n = R1 * 4096, R1 = 4, 8, 16): the four-step structure of xcorr_fused_stk_4step
// (xcorr_stockham.hip) in fp32.  One n-element fp32 slice of FusedParams::gscratch per workgroup (8 n bytes: a quarter of
// the fp64 kernel's two slices), 34.8 KB of LDS for the on-chip 4096-point rows:
//   sweep 0: rows -> fp64 shifted statistics, provisional fp32 copy of d = x - x[0] into the slice;
//   sweep 1: centre + scale, radix R1 over m1, twiddle W_n^(m2 k1), in place;
//   rows:    each of the R1 rows of 4096 points through lds_transforms_f<12> (times xcf[k1 + R1 k2]), in place;
//   sweep 2: twiddle + radix R1 over k1 -> cc; first time for the maxima, second time (the slice is read again, the
//            cheapest way to see every lag once the maximum is known) for the flags and the estimate.
namespace sstk {

template <int K>
__device__ __forceinline__ void block_sum_f64(double (&q)[K], double *red, const int lane, const int wave)
{
#pragma unroll
    for (int k = 0; k < K; k++)
        q[k] = wave_sum_dpp(q[k]);
    __syncthreads(); // red free
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++)
            red[wave * K + k] = q[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++)
        q[k] = (red[k] + red[K + k]) + (red[2 * K + k] + red[3 * K + k]);
}

} // namespace sstk

template <int LOGN>
__global__ __launch_bounds__(256, 3) void xcorr_screen_pass_4step(const FusedParams p)
{
    using namespace occ4;
    using namespace sstk;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;
    constexpr int CH = S / 256;
    constexpr int R1 = n / 4096;
    constexpr int Q1 = 16 / R1;
    static_assert(LOGN >= 14 && LOGN <= 16, "fp32 four-step screening pass: n = 16384 ... 65536");
    __shared__ f2 buf[4096 + 256];
    __shared__ double red[16];
    __shared__ float redf[8];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    f2 *const Y = reinterpret_cast<f2 *>(p.gscratch) + (size_t)blockIdx.x * (size_t)n;
    const int N = p.N, pad = n - N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);
    const float2 *__restrict__ twm = p.twmf;
    const float window = (float)p.screen_delta;
    const int max_lag = p.scr_max_lag;

    for (long long pair = blockIdx.x; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        const double *__restrict__ ra = p.rows + rA * p.stride;
        const double *__restrict__ rb = p.rows + (hasB ? rB : rA) * p.stride;
        const double KA = ra[0], KB = rb[0];
        // twiddle of sweeps 1 and 2: v[m + r Q1] *= W_n^(m2 r), m2 = j + m S (single rounded table entries)
        const auto twiddle_rows = [&](f2 (&v)[16], const int j) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < Q1; m++) {
                const int m2 = j + m * S;
#pragma unroll
                for (int r = 1; r < R1; r++) {
                    const int e = (m2 * r * (65536 / n)) & 65535;
                    const float2 w = twm[e & 32767];
                    const f2 ws = e >= 32768 ? mk2(-w.x, -w.y) : mk2(w.x, w.y);
                    v[m + r * Q1] = cmulf(v[m + r * Q1], ws);
                }
            }
        };
        // ---- sweep 0: statistics (fp64), provisional fp32 copy into the slice
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        __syncthreads(); // the previous pair's last reads of the slice are done
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = t + 256 * ch;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = j + i * S - pad;
                const int ec = e < 0 ? 0 : e; // a pad position loads x[0]: d = 0 without a mask
                const double da = __builtin_nontemporal_load(ra + ec) - KA, db = __builtin_nontemporal_load(rb + ec) - KB;
                Y[j + i * S] = mk2((float)da, (float)db);
                q[0] += da;
                q[1] = fma(da, da, q[1]);
                q[2] += db;
                q[3] = fma(db, db, q[3]);
            }
        }
        block_sum_f64<4>(q, red, lane, wave); // (its barriers also order sweep 0 before sweep 1)
        const double mA = q[0] * invN, mB = q[2] * invN;
        const double varA = (q[1] - q[0] * q[0] * invN) * invNm1, varB = (q[3] - q[2] * q[2] * invN) * invNm1;
        const bool nanA = !__builtin_isfinite(varA), nanB = !__builtin_isfinite(varB);
        const bool zeroA = !nanA && !(varA > 0.0), zeroB = !nanB && !(varB > 0.0);
        const int eA = (int)((__double_as_longlong(varA) >> 52) & 0x7ff) - 1023;
        const int eB = (int)((__double_as_longlong(varB) >> 52) & 0x7ff) - 1023;
        const bool redoA = !(zeroA || nanA) && (eA > 200 || eA < -200 || mA * mA > 64.0 * varA);
        const bool redoB = !(zeroB || nanB) && (eB > 200 || eB < -200 || mB * mB > 64.0 * varB);
        const bool offA = zeroA || nanA || redoA, offB = zeroB || nanB || redoB || !hasB;
        const float sclA = offA ? 0.f : __int_as_float((127 - (eA >> 1)) << 23);
        const float sclB = offB ? 0.f : __int_as_float((127 - (eB >> 1)) << 23);
        const float mAf = offA ? 0.f : (float)mA, mBf = offB ? 0.f : (float)mB;
        // ---- sweep 1: centre + scale, radix R1 over m1, twiddle, in place
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = t + 256 * ch;
            f2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool valid = j + i * S - pad >= 0;
                const f2 d = Y[j + i * S];
                v[i] = mk2((valid && !offA) ? (d.x - mAf) * sclA : 0.f, (valid && !offB) ? (d.y - mBf) * sclB : 0.f);
            }
            dft_small_f<R1>(v);
            twiddle_rows(v, j);
#pragma unroll
            for (int i = 0; i < 16; i++)
                Y[j + i * S] = v[i];
        }
        __syncthreads();
        // ---- rows: k1 = 0 .. R1-1, 4096 points each, on chip
#pragma clang loop unroll(disable)
        for (int k1 = 0; k1 < R1; k1++) {
            f2 *const row = Y + k1 * 4096;
            f2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = row[t + 256 * i];
            lds_transforms_f<12>(v, buf, twm, t, [&](int jj, int r) __attribute__((always_inline)) {
                const float2 x = p.xcf[k1 + R1 * (jj + 256 * r)];
                return mk2(x.x, x.y);
            });
#pragma unroll
            for (int i = 0; i < 16; i++)
                row[t + 256 * i] = v[i];
        }
        __syncthreads();
        // ---- sweep 2, first time: the fp32 maxima
        float ma = 0.f, mb = 0.f;
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = t + 256 * ch;
            f2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = Y[j + i * S];
            twiddle_rows(v, j);
            dft_small_f<R1>(v);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                ma = fmaxf(ma, fabsf(v[i].x));
                mb = fmaxf(mb, fabsf(v[i].y));
            }
        }
        ma = wave_max_f32_dpp(ma);
        mb = wave_max_f32_dpp(mb);
        if (lane == 0) {
            redf[wave] = ma;
            redf[4 + wave] = mb;
        }
        __syncthreads();
        const float MA = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        const float MB = fmaxf(fmaxf(redf[4], redf[5]), fmaxf(redf[6], redf[7]));
        // ---- sweep 2, second time: every lag within the window of the maximum.  The values are RECOMPUTED here, by a
        // second copy of the same code: the compiler is free to contract multiply-adds differently in the two copies, so
        // the holder of the maximum is found with a one-part-in-a-million tolerance instead of an equality (several
        // threads may then report; any of their values is the maximum to 1e-6 relative, far inside E)
        {
            const float thA = MA - window, thB = MB - window;
            const float topA = MA * 0.999999f, topB = MB * 0.999999f;
            unsigned fA = 0u, fB = 0u;
#pragma clang loop unroll(disable)
            for (int ch = 0; ch < CH; ch++) {
                const int j = t + 256 * ch;
                f2 v[16];
#pragma unroll
                for (int i = 0; i < 16; i++)
                    v[i] = Y[j + i * S];
                twiddle_rows(v, j);
                dft_small_f<R1>(v);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float xa = v[i].x, xb = v[i].y;
                    const bool ha = fabsf(xa) >= thA, hb = fabsf(xb) >= thB;
                    if (ha || hb) {
                        const int idx = j + i * S;
                        const int lg = idx > n / 2 ? idx - n : idx;
                        const unsigned in = (lg < 0 ? -lg : lg) <= max_lag ? SCR_IN : SCR_OUT;
                        if (ha)
                            fA |= in | (xa > 0.f ? SCR_POS : 0u) | (xa < 0.f ? SCR_NEG : 0u);
                        if (hb)
                            fB |= in | (xb > 0.f ? SCR_POS : 0u) | (xb < 0.f ? SCR_NEG : 0u);
                        if (ha && fabsf(xa) >= topA && !offA) {
                            p.mv[rA] = (double)xa * __longlong_as_double((long long)(1023 + (eA >> 1)) << 52);
                            p.lag[rA] = lg;
                        }
                        if (hb && fabsf(xb) >= topB && !offB) {
                            p.mv[rB] = (double)xb * __longlong_as_double((long long)(1023 + (eB >> 1)) << 52);
                            p.lag[rB] = lg;
                        }
                    }
                }
            }
            if (fA && !offA)
                atomicOr(&p.scr_flags[rA], fA);
            if (fB && !offB && hasB)
                atomicOr(&p.scr_flags[rB], fB);
            if (t == 0) {
                p.scr_var[rA] = varA;
                if (offA) {
                    p.mv[rA] = nanA ? __builtin_nan("") : 0.0;
                    p.lag[rA] = 0;
                    atomicOr(&p.scr_flags[rA], nanA ? SCR_NAN : (redoA ? SCR_REFINE : SCR_IN));
                }
                if (hasB) {
                    p.scr_var[rB] = varB;
                    if (offB) {
                        p.mv[rB] = nanB ? __builtin_nan("") : 0.0;
                        p.lag[rB] = 0;
                        atomicOr(&p.scr_flags[rB], nanB ? SCR_NAN : (redoB ? SCR_REFINE : SCR_IN));
                    }
                }
            }
        }
    }
}

template <int LOGN>
static hipError_t launch_4step(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (!p.gscratch)
        return hipErrorInvalidValue;
    // (one n-element fp32 slice per workgroup = half an fp64 slice)
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * 3);
    if (grid > 2 * p.gscratch_slices)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL((xcorr_screen_pass_4step<LOGN>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_one(const FusedParams &p, int num_cus, hipStream_t stream)
{
    constexpr int S = (1 << LOGN) / 16;
    constexpr int G = S >= 256 ? 1 : 256 / S;
    constexpr int TPB = S * G;
    const long long ngroups = (p.npairs + G - 1) / G;
    const long long grid = std::min<long long>(ngroups, (long long)num_cus * (TPB > 256 ? 1 : 3) * 4);
    hipLaunchKernelGGL((xcorr_screen_pass_stk<LOGN>), dim3((unsigned)grid), dim3(TPB), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_screen_pass_stk(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (!p.scr_flags || !p.scr_var || !p.xcf || !p.twmf || p.N <= p.n / 2 || p.N > p.n)
        return hipErrorInvalidValue;
    switch (p.logn) {
    case 9: return launch_one<9>(p, num_cus, stream);
    case 10: return launch_one<10>(p, num_cus, stream);
    case 11: return launch_one<11>(p, num_cus, stream);
    case 13: return launch_one<13>(p, num_cus, stream);
    case 14: return launch_4step<14>(p, num_cus, stream);
    case 15: return launch_4step<15>(p, num_cus, stream);
    case 16: return launch_4step<16>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace muse
