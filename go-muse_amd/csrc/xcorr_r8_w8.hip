// xcorr_r8_w8.hip -- n = 4096 as 8^4: 512 threads (8 waves) per pair of series,
// 8 complex points per thread, four radix-8 passes, three full-size LDS
// transposes per FFT.
//
// Mathematics: identical to xcorr_fused_n4096 (xcorr_kernels.hip header; the
// reference path is xCorrWithX, /root/reference/xcorr.go:160-197).
//
// Why: the 16-points-per-thread kernels need >= 64 VGPRs for data alone and are
// stuck at 3 waves per SIMD with half-round transposes (18 barriers per pair).
// Eight points per thread need ~110 VGPRs including the prefetch of the next
// pair, so two 512-thread workgroups = 16 waves = 4 per SIMD fit a CU with the
// FULL 69.6 KB transpose buffer: two barriers per transpose, 14 per pair.
//
// Index algebra (i = 512a + 64b + 8c + d, f = k1 + 8k2 + 64k3 + 512k4):
//   pass 1  thread t = 64b+8c+d holds a     -> k1, twiddle W_4096^(k1 t)
//   X1      (k1 | b,c,d) -> (b | k1,c,d)       pos = 512 k1 + t
//   pass 2  thread 64k1+8c+d holds b        -> k2, twiddle W_512^(k2 (8c+d))
//   X2      (k2 | k1,c,d) -> (c | k1,k2,d)     pos = 520 k2 + 64 k1 + 8c + d
//   pass 3  thread 64k1+8k2+d holds c       -> k3, twiddle W_64^(k3 d)
//   X3      (k3 | k1,k2,d) -> (d | k1,k2,k3)   pos = 544 k3 + 8e + (e>>1) + d, e = k1+8k2
//   pass 4  thread k1+8k2+64k3 holds d      -> k4;  f = t + 512 k4
// so a thread starts and ends with elements t + 512*j (coalesced loads, the
// second FFT consumes the first one's layout).  Bank checks (ds_write_b128: 8
// consecutive lanes on 8 distinct 16-byte slots mod 8; ds_read_b128: 16-lane
// groups on 16 distinct slots mod 16): every writer puts consecutive lanes on
// consecutive slots; X1 readers read slot == lane (mod 64); X2 readers see
// (8 k2 + d) mod 16; X3 readers see (8 (e&1) + (e>>1) + d) mod 16 -- bijections
// on any aligned 16 lanes and on the hardware's {0-3,12-15,20-27}-style groups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fft_device.h"
#include "xcorr_kernels.h"

namespace muse {

constexpr int W8_THREADS = 512;
constexpr int W8_XBUF = 8 * 544; // double2 elements: 69,632 B

namespace w8 {

__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void lds_barrier() // LDS-only: does not drain global loads
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
typedef double d2v __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ gptr<T> scalar_ptr(const T *p) // SGPR base, global address space kept
{
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+s"(u));
    return (gptr<T>)u;
}
__device__ __forceinline__ double2 ldg2(gptr<double2> p, int i)
{
    const d2v x = ((gptr<d2v>)p)[i];
    return make_double2(x.x, x.y);
}
__device__ __forceinline__ double uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// 8-point forward DFT in registers (radix-2 DIF, 56 real ops), natural in/out.
__device__ __forceinline__ void dft8(double2 (&v)[8])
{
    constexpr double H = 0.70710678118654752440;
    // stage 1: a_j = x_j + x_{j+4};  b_j = (x_j - x_{j+4}) W8^j
    const double2 a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    const double2 b0 = csub(v[0], v[4]);
    double2 u = csub(v[1], v[5]);
    const double2 b1 = make_double2((u.x + u.y) * H, (u.y - u.x) * H); // * (1 - i)/sqrt2
    u = csub(v[2], v[6]);
    const double2 b2 = make_double2(u.y, -u.x); // * -i
    u = csub(v[3], v[7]);
    const double2 b3 = make_double2((u.y - u.x) * H, -(u.x + u.y) * H); // * (-1 - i)/sqrt2
    // 4-point DFTs of a (even outputs) and b (odd outputs)
    const double2 c0 = cadd(a0, a2), c1 = cadd(a1, a3), c2 = csub(a0, a2);
    u = csub(a1, a3);
    const double2 c3 = make_double2(u.y, -u.x);
    const double2 e0 = cadd(b0, b2), e1 = cadd(b1, b3), e2 = csub(b0, b2);
    u = csub(b1, b3);
    const double2 e3 = make_double2(u.y, -u.x);
    v[0] = cadd(c0, c1);
    v[4] = csub(c0, c1);
    v[2] = cadd(c2, c3);
    v[6] = csub(c2, c3);
    v[1] = cadd(e0, e1);
    v[5] = csub(e0, e1);
    v[3] = cadd(e2, e3);
    v[7] = csub(e2, e3);
}

struct RawPair {
    double a[8], b[8]; // element t + 512*j of the two (zero-padded) rows
    double ka, kb;     // first sample of each row
};
template <bool PADDED>
__device__ __forceinline__ void issue_row_loads(RawPair &r, const FusedParams &p, long long pair, int t, int pad)
{
    const long long rA = 2 * pair;
    const long long rB = (rA + 1 < p.M) ? rA + 1 : rA;
    const gptr<double> ra = scalar_ptr(p.rows + rA * p.stride);
    const gptr<double> rb = scalar_ptr(p.rows + rB * p.stride);
    r.ka = ra[0];
    r.kb = rb[0];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (PADDED) {
            int i = t + 512 * j - pad;
            i = i < 0 ? 0 : i; // clamped: always load, masked later
            r.a[j] = __builtin_nontemporal_load(ra + i);
            r.b[j] = __builtin_nontemporal_load(rb + i);
        } else {
            r.a[j] = __builtin_nontemporal_load(ra + 512 * j + t);
            r.b[j] = __builtin_nontemporal_load(rb + 512 * j + t);
        }
    }
}

// forward FFT: v[j] = x[t + 512 j] -> v[j] = X[t + 512 j].  MULXC: multiply by the
// batch's conj(X)/n table (after the DC correction dc for N == n).  PREFETCH: issue
// the next pair's row loads before the last pass.
template <bool MULXC, bool PADDED>
__device__ __forceinline__ void fft4096(double2 (&v)[8], double2 *xbuf, const double2 *tw2s, const double2 *tw3s,
                                        const double2 *tw1g, const double2 *xcg, const double2 dc, const int t,
                                        RawPair &raw, const FusedParams &p, long long next_pair, int pad)
{
    const int w = t >> 6, l = t & 63; // wave index (k1 or k3 of the reader), lane
    const int k2r = (t >> 3) & 7, d = t & 7;
    // ---- pass 1: factors fetched before the butterflies (L2 latency overlapped)
    {
        const gptr<double2> tp = scalar_ptr(tw1g);
        double2 tw[7];
#pragma unroll
        for (int k = 1; k < 8; k++)
            tw[k - 1] = ldg2(tp, k * 512 + t);
        fence();
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; k++)
            v[k] = cmul(v[k], tw[k - 1]);
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 8; k++)
        xbuf[512 * k + t] = v[k];
    lds_barrier();
#pragma unroll
    for (int b = 0; b < 8; b++)
        v[b] = xbuf[512 * w + 64 * b + l];
    // ---- pass 2 (k1 = w; 8c + d = l)
    {
        double2 tw[7];
#pragma unroll
        for (int k = 1; k < 8; k++)
            tw[k - 1] = tw2s[k * 64 + l];
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; k++)
            v[k] = cmul(v[k], tw[k - 1]);
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 8; k++)
        xbuf[520 * k + t] = v[k];
    lds_barrier();
#pragma unroll
    for (int c = 0; c < 8; c++)
        v[c] = xbuf[520 * k2r + 64 * w + 8 * c + d];
    // ---- pass 3 (k1 = w, k2 = k2r, d)
    {
        double2 tw[7];
#pragma unroll
        for (int k = 1; k < 8; k++)
            tw[k - 1] = tw3s[k * 8 + d];
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; k++)
            v[k] = cmul(v[k], tw[k - 1]);
    }
    lds_barrier();
    {
        const int e = w + 8 * k2r; // k1 + 8 k2 of this writer
        const int wb = 8 * e + (e >> 1) + d;
#pragma unroll
        for (int k = 0; k < 8; k++)
            xbuf[544 * k + wb] = v[k];
    }
    lds_barrier();
    {
        const int rb = 544 * w + 8 * l + (l >> 1); // reader: k3 = w, e = l
#pragma unroll
        for (int dd = 0; dd < 8; dd++)
            v[dd] = xbuf[rb + dd];
    }
    // ---- pass 4
    if (MULXC) {
        const gptr<double2> xp = scalar_ptr(xcg);
        double2 xc[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            xc[k] = ldg2(xp, 512 * k + t);
        fence();
        dft8(v);
        if (t == 0) { // N == n: FFT(d - m)[0] = FFT(d)[0] - n m   (dc = 0 otherwise)
            v[0].x -= dc.x;
            v[0].y -= dc.y;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = cmul(v[k], xc[k]);
    } else {
        fence();
        issue_row_loads<PADDED>(raw, p, next_pair, t, pad);
        fence();
        dft8(v);
    }
}

struct Stat {
    double s1, s2; // shifted sums: sum d, sum d^2 (d = x - x[0])
};
__device__ __forceinline__ double variance(const Stat &s, double invN, double invNm1, bool &zero, bool &nan)
{
    const double var = (s.s2 - s.s1 * s.s1 * invN) * invNm1;
    nan = !__builtin_isfinite(var); // (not `var - var != 0`: fp-contract breaks it)
    zero = !nan && !(var > 0.0);
    return var;
}
// cross-wave combine of one series' argmax + result store: r[6*w + {0,1,2}] =
// wave w's {max |cc|, signed value, first index}, 8 waves
__device__ __forceinline__ void finalize(const double *r, const Stat &st, double invN, double invNm1, double *mv_out,
                                         int *lag_out)
{
    double best = r[0], bsv = r[1], bidx = r[2];
    const double cc0 = r[1];
#pragma unroll
    for (int w = 1; w < 8; w++) {
        const double m = r[6 * w], s = r[6 * w + 1], ix = r[6 * w + 2];
        if (m > best || (m == best && ix < bidx)) {
            best = m;
            bsv = s;
            bidx = ix;
        }
    }
    bool zero, nan;
    const double var = variance(st, invN, invNm1, zero, nan);
    const int idx = (best > 0.0) ? (int)bidx : 0; // nothing above 0: index 0, mv = cc[0]
    double mv = ((best > 0.0) ? bsv : cc0) * (1.0 / sqrt(var));
    int lag = idx > 2048 ? idx - 4096 : idx;
    if (zero) { mv = 0.0; lag = 0; }              // xcorr.go:166-167
    if (nan) { mv = __builtin_nan(""); lag = 0; } // NaN sigma: every cc is NaN
    *mv_out = mv;
    *lag_out = lag;
}

} // namespace w8

template <bool PADDED>
__global__ __launch_bounds__(W8_THREADS, 4) void xcorr_fused_n4096_w8(const FusedParams p)
{
    using namespace w8;
    __shared__ double2 xbuf[W8_XBUF];
    __shared__ double2 tw2s[8 * 64]; // W_512^(k l), row k = 0 unused
    __shared__ double2 tw3s[64];     // W_64^(k d)
    __shared__ double red[32 + 2 * 48]; // [0,32): z-norm partials (8 waves x 4); then 2 parities x 8 waves x 6
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int N = p.N;
    const int pad = 4096 - N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);

    // workgroup-lifetime twiddle tables from the W_65536 master table (half period stored):
    // W_512^m = W_65536^(128 m), W_64^m = W_65536^(1024 m)
    {
        const int k = t >> 6, l = t & 63;
        const int i2 = ((k * l) & 511) * 128;
        double2 w = p.twm[i2 & 32767];
        if (i2 >= 32768) // W^(x + 32768) = -W^x
            w = make_double2(-w.x, -w.y);
        tw2s[t] = w;
        if (t < 64) {
            const int kk = t >> 3, dd = t & 7;
            const int i3 = ((kk * dd) & 63) * 1024;
            double2 w3 = p.twm[i3 & 32767];
            if (i3 >= 32768)
                w3 = make_double2(-w3.x, -w3.y);
            tw3s[t] = w3;
        }
    }
    __syncthreads();

    long long prev_row = -1; // threads 0 / 1 keep the previous pair's row + statistics until its
    Stat prev_stat{0.0, 0.0}; // cross-wave argmax combine runs behind this pair's first barrier
    int parity = 0;

    RawPair raw;
    issue_row_loads<PADDED>(raw, p, blockIdx.x < p.npairs ? (long long)blockIdx.x : 0, t, pad);

    for (long long pair = blockIdx.x; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        // ---- consume the prefetched rows: d = x - K (pads -> 0), shifted one-pass statistics
        double2 v[8];
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        {
            const double KA = raw.ka, KB = raw.kb;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                double da = raw.a[j] - KA, db = raw.b[j] - KB;
                if (PADDED) {
                    const bool valid = t + 512 * j - pad >= 0;
                    da = valid ? da : 0.0;
                    db = valid ? db : 0.0;
                }
                v[j] = make_double2(da, db);
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
                red[wave * 4 + k] = q[k];
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++)
            q[k] = uniform(((red[k] + red[4 + k]) + (red[8 + k] + red[12 + k])) +
                           ((red[16 + k] + red[20 + k]) + (red[24 + k] + red[28 + k])));
        if (t < 2 && prev_row >= 0) // previous pair's argmax triples are visible now
            finalize(red + 32 + 48 * (parity ^ 1) + 3 * t, prev_stat, invN, invNm1, p.mv + prev_row, p.lag + prev_row);
        Stat stA{q[0], q[1]}, stB{q[2], q[3]};
        bool zeroA, nanA, zeroB, nanB;
        const double varA0 = variance(stA, invN, invNm1, zeroA, nanA);
        const double varB0 = variance(stB, invN, invNm1, zeroB, nanB);
        double mA = uniform(q[0] * invN), mB = uniform(q[2] * invN);
        if (PADDED) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const bool valid = t + 512 * j - pad >= 0;
                v[j].x = valid ? v[j].x - mA : 0.0;
                v[j].y = valid ? v[j].y - mB : 0.0;
            }
        }
        const bool deadA = zeroA || nanA, deadB = zeroB || nanB || !hasB;
        if (deadA || deadB) { // block-uniform, rare: exact zeros into the shared complex transform
#pragma unroll
            for (int j = 0; j < 8; j++) {
                v[j].x = deadA ? 0.0 : v[j].x;
                v[j].y = deadB ? 0.0 : v[j].y;
            }
        }
        if (hasB && !nanA && !nanB && sigma_spread_too_wide(varA0, varB0)) { // block-uniform, rare: see xcorr_r16_occ4.hip
            const double sA = pow2_inv_sigma(varA0), sB = pow2_inv_sigma(varB0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                v[j].x *= sA;
                v[j].y *= sB;
            }
            stA.s1 *= sA;
            stA.s2 *= sA * sA;
            stB.s1 *= sB;
            stB.s2 *= sB * sB;
            mA *= sA;
            mB *= sB;
        }
        const double2 dc = PADDED ? make_double2(0.0, 0.0)
                                  : make_double2(uniform(deadA ? 0.0 : 4096.0 * mA), uniform(deadB ? 0.0 : 4096.0 * mB));
        long long nxt = pair + gridDim.x; // last iteration: harmless re-read of this pair
        nxt = nxt < p.npairs ? nxt : pair;
        // ---- Z = FFT(yA + i yB); V = Z conj(X)/n; ccA + i ccB = FFT(V)
        fft4096<true, PADDED>(v, xbuf, tw2s, tw3s, p.tw1w8, p.xc, dc, t, raw, p, nxt, pad);
        fft4096<false, PADDED>(v, xbuf, tw2s, tw3s, p.tw1w8, p.xc, dc, t, raw, p, nxt, pad);

        // ---- maxAbsIndex (xcorr.go:39-50), index = t + 512 j: per-thread max, wave max by
        // DPP, first index + sign by ballots (lowest j, then lowest lane == lowest index)
        double ma = 0.0, mb = 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            ma = fmax(ma, fabs(v[j].x));
            mb = fmax(mb, fabs(v[j].y));
        }
        const double wa = wave_max_dpp(ma), wb = wave_max_dpp(mb);
        int widxA = 0x7fffffff, widxB = 0x7fffffff;
        double svA = 0.0, svB = 0.0;
        if (wa > 0.0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned long long m = __ballot(fabs(v[j].x) == wa);
                if (m != 0ull && widxA == 0x7fffffff) {
                    const int l = __ffsll((long long)m) - 1;
                    widxA = wave * 64 + l + 512 * j;
                    const unsigned long long ng = __ballot(v[j].x < 0.0);
                    svA = ((ng >> l) & 1ull) ? -wa : wa;
                }
            }
        }
        if (wb > 0.0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned long long m = __ballot(fabs(v[j].y) == wb);
                if (m != 0ull && widxB == 0x7fffffff) {
                    const int l = __ffsll((long long)m) - 1;
                    widxB = wave * 64 + l + 512 * j;
                    const unsigned long long ng = __ballot(v[j].y < 0.0);
                    svB = ((ng >> l) & 1ull) ? -wb : wb;
                }
            }
        }
        if (lane == 0) { // {max |cc|, signed value (cc[0] when nothing is above 0), index}
            double *r = red + 32 + 48 * parity + 6 * wave;
            r[0] = widxA == 0x7fffffff ? 0.0 : wa;
            r[1] = widxA == 0x7fffffff ? v[0].x : svA; // wave 0 lane 0 holds cc[0]
            r[2] = (double)widxA;
            r[3] = widxB == 0x7fffffff ? 0.0 : wb;
            r[4] = widxB == 0x7fffffff ? v[0].y : svB;
            r[5] = (double)widxB;
        }
        if (t == 0) {
            prev_row = rA;
            prev_stat = stA;
        } else if (t == 1) {
            prev_row = hasB ? rB : -1;
            prev_stat = stB;
        }
        parity ^= 1;
    }
    lds_barrier();
    if (t < 2 && prev_row >= 0)
        finalize(red + 32 + 48 * (parity ^ 1) + 3 * t, prev_stat, invN, invNm1, p.mv + prev_row, p.lag + prev_row);
}

hipError_t launch_fused_w8(const FusedParams &p, int num_cus, hipStream_t stream)
{
    long long grid = p.npairs;
    int mult = 1; // resident workgroups per CU x mult (MUSE_HIP_GRID_MULT: tuning aid)
    if (const char *m = getenv("MUSE_HIP_GRID_MULT"))
        mult = atoi(m) > 0 ? atoi(m) : mult;
    const long long cap = (long long)num_cus * 2 * mult;
    if (grid > cap)
        grid = cap;
    if (p.N < 4096)
        hipLaunchKernelGGL(xcorr_fused_n4096_w8<true>, dim3((unsigned)grid), dim3(W8_THREADS), 0, stream, p);
    else
        hipLaunchKernelGGL(xcorr_fused_n4096_w8<false>, dim3((unsigned)grid), dim3(W8_THREADS), 0, stream, p);
    return hipGetLastError();
}

} // namespace muse
