// xcorr_long_batched.hip -- long series (n = 32768, 65536), the four-step transform of xcorr_long.hip with ONE KERNEL PER
// PHASE over a batch of pairs, so that the scratch slices of the pairs in flight stay in the 256 MiB Infinity Cache.
//
// xcorr_long.hip keeps a pair inside one workgroup from its rows to its result: 1 024 workgroups, 1 024 slices (1 GB) in
// flight, a slice line is read back a third of a pair's time (~ 300 us) after it was written, ~ 1.5 GB of other traffic later --
// every crossing of the slice goes to HBM and the kernel runs at the 4.5 TB/s the memory side gives.
// Here the whole chip works on a batch of B pairs at a time (B x n x 16 bytes <= 64 MB), one phase per launch:
//   long_sweep1  grid (n/4096 chunks, B): rows -> d = x - K, partial statistics, radix R1 over m1, twiddle -> slice
//   long_rows    grid (R1, B): one 4096-point row through the n = 4096 kernel's pair of transforms, in place
//   long_sweep2  grid (chunks, B): twiddle, radix R1 over k1, (N < n: - mean c1[lag]), the chunk's first maximum -> candidates
//   long_final   one thread per pair: candidates -> (lag, score), statistics, redo list
// A phase lasts tens of microseconds and the kernel boundary is the barrier between phases (no spinning, nothing to
// deadlock); consecutive batches alternate between two streams and two slice regions so that one batch's rows stage
// overlaps the other's sweeps.  Semantics, tables and the redo path are those of xcorr_long.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "long_device.h"

namespace muse {

namespace lngb {

using namespace occ4;
using namespace fold;
using namespace foldk;
using namespace lng;

typedef d2v __attribute__((address_space(1))) *gd2;
__device__ __forceinline__ int opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// per pair of the batch: partial sums [chunk][4] (sum dA, sum dA^2, sum dB, sum dB^2) and candidates [chunk][8]
// (|max| A, signed A, index A, |max| B, signed B, index B, cc[0] A, cc[0] B -- the last two from chunk 0 only)
constexpr int PART = 4, CAND = 8;

template <int LOGN, bool PADDED>
__global__ __launch_bounds__(256, 4) void long_sweep1(const FusedParams p)
{
    constexpr int n = 1 << LOGN, S = n / 16, CH = S / 256, R1 = n / 4096, Q1 = 16 / R1;
    constexpr int T = Q1 * (R1 - 1), NB = (T + 3) / 4;
    __shared__ double red[16];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ch = blockIdx.x, b = blockIdx.y;
    const long long pair = p.pair0 + b;
    const long long rA = 2 * pair;
    const bool hasB = rA + 1 < p.M;
    const double *__restrict__ ra = p.rows + rA * p.stride;
    const double *__restrict__ rb = p.rows + (hasB ? rA + 1 : rA) * p.stride;
    double2 *const Y = p.gscratch + (size_t)b * (size_t)n;
    const double2 *__restrict__ twl = p.twl;
    const int N = PADDED ? p.N : n, pad = n - N;
    const double KA = ra[0], KB = rb[0];
    const auto tw_load = [&](int f, unsigned jj) __attribute__((always_inline)) {
        const int m = f / (R1 - 1), k1 = 1 + f % (R1 - 1);
        return ldg2u(scalar_ptr_at(twl, k1 * 4096 + m * S), jj);
    };
    const int j = opaque(t + 256 * ch) & (S - 1);
    double2 v[16];
    double xa[16], xb[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (PADDED) {
            const int e = j + i * S - pad;
            const unsigned ec = (unsigned)(e < 0 ? 0 : e);
            xa[i] = __builtin_nontemporal_load(scalar_ptr(ra) + ec);
            xb[i] = __builtin_nontemporal_load(scalar_ptr(rb) + ec);
        } else {
            xa[i] = __builtin_nontemporal_load(scalar_ptr_at(ra, i * S) + (unsigned)j);
            xb[i] = __builtin_nontemporal_load(scalar_ptr_at(rb, i * S) + (unsigned)j);
        }
    }
    double2 wq[2][4];
#pragma unroll
    for (int f = 0; f < 4 && f < T; f++)
        wq[0][f] = tw_load(f, (unsigned)j);
    fence();
    double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        double da = xa[i] - KA, db = xb[i] - KB;
        if (PADDED) {
            const bool valid = j + i * S - pad >= 0;
            da = valid ? da : 0.0;
            db = valid ? db : 0.0;
        }
        v[i] = make_double2(da, db);
        q[0] += da;
        q[1] = fma(da, da, q[1]);
        q[2] += db;
        q[3] = fma(db, db, q[3]);
    }
    sweep_dft<R1>(v);
    const unsigned js = (unsigned)(opaque(t + 256 * ch) & (S - 1));
#pragma unroll
    for (int m = 0; m < Q1; m++)
        ((gd2)scalar_ptr_at(Y, (long long)m * S))[js] = d2v{v[m].x, v[m].y};
#pragma unroll
    for (int bt = 0; bt < NB; bt++) {
        fence();
        if (bt + 1 < NB) {
#pragma unroll
            for (int f = 4 * (bt + 1); f < 4 * (bt + 2) && f < T; f++)
                wq[(bt + 1) & 1][f & 3] = tw_load(f, js);
        }
        fence();
#pragma unroll
        for (int f = 4 * bt; f < 4 * (bt + 1) && f < T; f++) {
            const int m = f / (R1 - 1), k1 = 1 + f % (R1 - 1);
            const double2 z = cmul(v[m + brev<R1>(k1) * Q1], wq[bt & 1][f & 3]);
            ((gd2)scalar_ptr_at(Y, (long long)(m + k1 * Q1) * S))[js] = d2v{z.x, z.y};
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double w = wave_sum_dpp(q[k]);
        if (lane == 0)
            red[4 * wave + k] = w;
    }
    __syncthreads();
    if (t < 4)
        p.lpart[((size_t)b * CH + ch) * PART + t] = (red[t] + red[4 + t]) + (red[8 + t] + red[12 + t]);
}

template <int LOGN, bool PADDED>
__global__ __launch_bounds__(256, 4) void long_rows(const FusedParams p)
{
    constexpr int n = 1 << LOGN;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int k1 = blockIdx.x, b = blockIdx.y;
    double2 *const row = p.gscratch + (size_t)b * (size_t)n + (size_t)k1 * 4096;
    if (t < 128)
        g2s[t] = p.g2[t];
    double2 v[16];
    {
        const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const d2v z = __builtin_nontemporal_load((gd2)scalar_ptr_at(row, 256 * i) + tl);
            v[i] = make_double2(z.x, z.y);
        }
    }
    __syncthreads();
    row_transforms(v, xbuf, xbuf + XW * wave, g2s, p.g3a, p.g3b, p.xcp + k1 * 4096, t, wave, !PADDED && k1 == 0);
    {
        const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
        for (int m = 0; m < 16; m++)
            ((gd2)scalar_ptr_at(row, 256 * m))[tl] = d2v{v[BR16(m)].x, v[BR16(m)].y};
    }
}

template <int LOGN, bool PADDED>
__global__ __launch_bounds__(256, 4) void long_sweep2(const FusedParams p)
{
    constexpr int n = 1 << LOGN, S = n / 16, CH = S / 256, R1 = n / 4096, Q1 = 16 / R1;
    constexpr int T = Q1 * (R1 - 1), NB = (T + 3) / 4;
    __shared__ double red[16];
    __shared__ int redi[8];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ch = blockIdx.x, b = blockIdx.y;
    const double2 *const Y = p.gscratch + (size_t)b * (size_t)n;
    const double2 *__restrict__ twl = p.twl;
    const auto tw_load = [&](int f, unsigned jj) __attribute__((always_inline)) {
        const int m = f / (R1 - 1), k1 = 1 + f % (R1 - 1);
        return ldg2u(scalar_ptr_at(twl, k1 * 4096 + m * S), jj);
    };
    double mA = 0.0, mB = 0.0;
    if (PADDED) { // the pair's means: sums of the chunks' partial sums (sweep 1)
        const double invN = 1.0 / (double)p.N;
        double s0 = 0.0, s2 = 0.0;
        for (int c = 0; c < CH; c++) {
            s0 += p.lpart[((size_t)b * CH + c) * PART + 0];
            s2 += p.lpart[((size_t)b * CH + c) * PART + 2];
        }
        mA = s0 * invN;
        mB = s2 * invN;
    }
    const int j = opaque(t + 256 * ch) & (S - 1);
    double2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const d2v z = __builtin_nontemporal_load((gd2)scalar_ptr_at(Y, (long long)i * S) + (unsigned)j);
        v[i] = make_double2(z.x, z.y);
    }
    {
        double2 wq[2][4];
#pragma unroll
        for (int f = 0; f < 4 && f < T; f++)
            wq[0][f] = tw_load(f, (unsigned)j);
#pragma unroll
        for (int bt = 0; bt < NB; bt++) {
            fence();
            if (bt + 1 < NB) {
#pragma unroll
                for (int f = 4 * (bt + 1); f < 4 * (bt + 2) && f < T; f++)
                    wq[(bt + 1) & 1][f & 3] = tw_load(f, (unsigned)j);
            }
            fence();
#pragma unroll
            for (int f = 4 * bt; f < 4 * (bt + 1) && f < T; f++) {
                const int m = f / (R1 - 1), k1 = 1 + f % (R1 - 1);
                v[m + k1 * Q1] = cmul(v[m + k1 * Q1], wq[bt & 1][f & 3]);
            }
        }
    }
    sweep_dft<R1>(v);
    // the lane's first maximum (ascending i = ascending lag index j + i S: strictly greater keeps the first)
    double csa = 0.0, csb = 0.0, cc0a = 0.0, cc0b = 0.0;
    int cia = 0, cib = 0;
    const int jc = opaque(t + 256 * ch) & (S - 1);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int m = i % Q1, l1 = i / Q1;
        double2 c = v[m + brev<R1>(l1) * Q1];
        if (PADDED) {
            const double c1 = scalar_ptr_at(p.c1, i * S)[(unsigned)jc];
            c = make_double2(fma(-mA, c1, c.x), fma(-mB, c1, c.y));
        }
        if (i == 0) {
            cc0a = c.x;
            cc0b = c.y;
        }
        const bool ga = fabs(c.x) > fabs(csa), gb = fabs(c.y) > fabs(csb);
        csa = ga ? c.x : csa;
        cia = ga ? i : cia;
        csb = gb ? c.y : csb;
        cib = gb ? i : cib;
    }
    const double ma = fabs(csa), mb = fabs(csb);
    const int ia = jc + cia * S, ib = jc + cib * S;
    // the chunk's first maximum: greatest |cc|, lowest index among equals
    const double wa = wave_max(ma), wb = wave_max(mb);
    if (lane == 0) {
        red[wave] = wa;
        red[4 + wave] = wb;
    }
    __syncthreads();
    const double MA = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const double MB = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    int ca = (ma == MA && MA > 0.0) ? ia : 0x7fffffff;
    int cb = (mb == MB && MB > 0.0) ? ib : 0x7fffffff;
    ca = wave_min_i(ca);
    cb = wave_min_i(cb);
    if (lane == 0) {
        redi[wave] = ca;
        redi[4 + wave] = cb;
    }
    __syncthreads();
    const int IA = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
    const int IB = min(min(redi[4], redi[5]), min(redi[6], redi[7]));
    double *const cand = p.lcand + ((size_t)b * CH + ch) * CAND;
    if (IA == 0x7fffffff ? t == 0 : (ia == IA && ma == MA)) {
        cand[0] = IA == 0x7fffffff ? 0.0 : MA;
        cand[1] = csa;
        cand[2] = (double)IA;
    }
    if (IB == 0x7fffffff ? t == 0 : (ib == IB && mb == MB)) {
        cand[3] = IB == 0x7fffffff ? 0.0 : MB;
        cand[4] = csb;
        cand[5] = (double)IB;
    }
    if (t == 0) { // (chunk 0, lane 0: cc[0], the value reported when nothing is above 0)
        cand[6] = cc0a;
        cand[7] = cc0b;
    }
}

template <int LOGN>
__global__ __launch_bounds__(64) void long_final(const FusedParams p)
{
    constexpr int n = 1 << LOGN, CH = n / 4096;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= p.nb)
        return;
    const long long pair = p.pair0 + b;
    const long long rA = 2 * pair;
    const bool hasB = rA + 1 < p.M;
    const int N = p.N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < CH; c++)
        for (int k = 0; k < 4; k++)
            q[k] += p.lpart[((size_t)b * CH + c) * lngb::PART + k];
    bool zero[2], nan[2];
    double var[2];
    var[0] = occ4::variance(occ4::Stat{q[0], q[1]}, invN, invNm1, zero[0], nan[0]);
    var[1] = occ4::variance(occ4::Stat{q[2], q[3]}, invN, invNm1, zero[1], nan[1]);
    for (int s = 0; s < (hasB ? 2 : 1); s++) {
        double best = 0.0, bsv = 0.0;
        int bidx = 0x7fffffff;
        for (int c = 0; c < CH; c++) {
            const double *cand = p.lcand + ((size_t)b * CH + c) * lngb::CAND + 3 * s;
            const double m = cand[0];
            const int ix = (int)cand[2];
            if (m > best || (m == best && m > 0.0 && ix < bidx)) {
                best = m;
                bsv = cand[1];
                bidx = ix;
            }
        }
        const bool none = !(best > 0.0);
        double y = __builtin_amdgcn_rsq(var[s]);
        y = y * fma(-0.5 * var[s] * y, y, 1.5);
        y = y * fma(-0.5 * var[s] * y, y, 1.5);
        const int idx = none ? 0 : bidx;
        double mv = (none ? p.lcand[(size_t)b * CH * lngb::CAND + 6 + s] : bsv) * y;
        int lag = idx > n / 2 ? idx - n : idx;
        if (zero[s]) { mv = 0.0; lag = 0; }              // xcorr.go:166-167
        if (nan[s]) { mv = __builtin_nan(""); lag = 0; } // placeholder: the pair is redone
        p.mv[rA + s] = mv;
        p.lag[rA + s] = lag;
    }
    if (nan[0] || (hasB && (nan[1] || sigma_spread_too_wide(var[0], var[1])))) {
        const int slot = atomicAdd(p.ovf_count, 1);
        p.ovf_list[slot] = pair;
    }
}

template <int LOGN, bool PADDED>
static hipError_t launch_batch(const FusedParams &p, hipStream_t s)
{
    constexpr int n = 1 << LOGN, CH = n / 4096, R1 = n / 4096;
    hipLaunchKernelGGL((long_sweep1<LOGN, PADDED>), dim3(CH, (unsigned)p.nb), dim3(256), 0, s, p);
    hipLaunchKernelGGL((long_rows<LOGN, PADDED>), dim3(R1, (unsigned)p.nb), dim3(256), 0, s, p);
    hipLaunchKernelGGL((long_sweep2<LOGN, PADDED>), dim3(CH, (unsigned)p.nb), dim3(256), 0, s, p);
    hipLaunchKernelGGL((long_final<LOGN>), dim3((unsigned)((p.nb + 63) / 64)), dim3(64), 0, s, p);
    return hipGetLastError();
}

} // namespace lngb

// n = 32768, 65536 (float64 rows, every pair: no pair list); N in (n/2, n], N < n needs p.c1.  `aux`: two streams the batches
// alternate between, forked from / joined to `stream` through the three events; p.gscratch holds 2 x batch slices, p.lpart /
// p.lcand 2 x batch x (n / 4096) x 4 / 8 doubles.
hipError_t launch_fused_long_batched(const FusedParams &p0, int batch, hipStream_t stream, hipStream_t aux0, hipStream_t aux1,
                                     hipEvent_t fork, hipEvent_t join0, hipEvent_t join1)
{
    if (!p0.rows || !p0.gscratch || !p0.twl || !p0.xcp || !p0.g2 || !p0.g3a || !p0.g3b || !p0.ovf_list || !p0.ovf_count || p0.pair_list ||
        !p0.lpart || !p0.lcand || batch < 1 || (p0.N < p0.n && !p0.c1) || (p0.logn != 15 && p0.logn != 16))
        return hipErrorInvalidValue;
    const int CH = p0.n / 4096;
    hipError_t e = hipEventRecord(fork, stream);
    if (e != hipSuccess)
        return e;
    hipStream_t aux[2] = {aux0, aux1};
    for (int k = 0; k < 2; k++) {
        e = hipStreamWaitEvent(aux[k], fork, 0);
        if (e != hipSuccess)
            return e;
    }
    int which = 0;
    for (long long first = 0; first < p0.npairs; first += batch, which ^= 1) {
        FusedParams p = p0;
        p.pair0 = first;
        p.nb = (int)std::min<long long>(batch, p0.npairs - first);
        p.gscratch = p0.gscratch + (size_t)which * (size_t)batch * (size_t)p0.n;
        p.lpart = p0.lpart + (size_t)which * (size_t)batch * CH * lngb::PART;
        p.lcand = p0.lcand + (size_t)which * (size_t)batch * CH * lngb::CAND;
        const bool padded = p.N < p.n;
        if (p.logn == 15)
            e = padded ? lngb::launch_batch<15, true>(p, aux[which]) : lngb::launch_batch<15, false>(p, aux[which]);
        else
            e = padded ? lngb::launch_batch<16, true>(p, aux[which]) : lngb::launch_batch<16, false>(p, aux[which]);
        if (e != hipSuccess)
            return e;
    }
    e = hipEventRecord(join0, aux0);
    if (e == hipSuccess)
        e = hipEventRecord(join1, aux1);
    if (e == hipSuccess)
        e = hipStreamWaitEvent(stream, join0, 0);
    if (e == hipSuccess)
        e = hipStreamWaitEvent(stream, join1, 0);
    return e;
}

} // namespace muse
