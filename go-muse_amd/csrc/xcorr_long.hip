// xcorr_long.hip -- long series, n = 16384 (test hook), 32768, 65536: xCorrWithX (/root/reference/xcorr.go:160-197) as a
// four-step transform whose 4096-point rows run on the n = 4096 kernel's folded-arithmetic machinery
// (foldk_device.h) at 16 waves per CU.
//
// n = R1 * 4096 (R1 = 4, 8, 16), input index m1 * 4096 + m2, spectrum index k1 + R1 * k2, lag index l1 * 4096 + l2:
//   Z[k1 + R1 k2]   = sum_m2 W_4096^(m2 k2) [ W_n^(m2 k1) sum_m1 W_R1^(m1 k1) d[m1 4096 + m2] ]
//   cc[l1 4096 + l2] = sum_k1 W_R1^(k1 l1) [ W_n^(k1 l2) sum_k2 W_4096^(k2 l2) Z[k1 + R1 k2] xc[k1 + R1 k2] ]
// sweep 1: rows of the group -> d = x - K (K = the series' first sample), shifted statistics, radix R1 over m1 in
//          registers, twiddle W_n^(m2 k1) -> the workgroup's scratch slice Y[k1][m2] (n complex, L2 / MALL / HBM);
// rows:    for each k1 the 4096 points Y[k1][.] -> forward transform, multiply by xc[k1 + R1 k2] (folded into the first
//          stage of the second transform), forward transform -> back to Y[k1][.]: exactly the n = 4096 kernel's two
//          transforms (three radix-16 passes each, half-round LDS transposes, 34.8 KB of LDS per workgroup);
// sweep 2: twiddle W_n^(k1 l2), radix R1 over k1 -> cc; (N < n: minus mean * c1[lag]) ; running argmax.
// The pair crosses the scratch slice four times (2 writes + 2 reads of 16 n bytes) next to its 16 N bytes of HBM input:
// that traffic, not arithmetic, bounds the kernel (docs/HISTORY.md section 4.3).
//
// As in xcorr_r16_fold.hip both series of a pair share one complex transform UNSCALED (their statistics are only known
// behind sweep 1): pairs with a NaN / Inf series or with sigmas too far apart are listed and redone by the kernel that
// rescales first (xcorr_stockham.hip, xcorr_fused_stk_4step).  N == n: the mean is never subtracted -- bin 0 of the
// centred series is exactly 0 and is zeroed in row 0.  N < n (leading zero pad, xcorr.go:176-181): the transforms run on
// d and every lag is corrected by -mean(d) c1[lag], c1 = the correlation of the valid-sample indicator with the
// reference (one table per batch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "long_device.h"

// Cache policy of the scratch slice (measured, profiles/r02_long_series.txt): plain stores, NON-TEMPORAL loads (+5 %: a
// slice line is read once, and the tables every workgroup shares keep their place in L2).
#define SLICE_ST(ptr, val) (*(ptr) = (val))
#define SLICE_LD(ptr) __builtin_nontemporal_load(ptr)

namespace muse {

// MULTI: R references against the group in ONE pass (muse_batch_score_many): sweep 1 and the rows' first transforms
// run once per pair, the row spectra stay in the workgroup's slice in register order, and every reference takes product, second
// transform (into a SECOND slice per workgroup), sweep 2 and argmax from there: 3 + 3 R slice crossings per pair instead of 4 R
// (and one read of the rows instead of R) -- x 1.25 per reference at R = 4, x 1.4 at R = 8.
template <int LOGN, bool PADDED, bool MULTI = false>
__global__ __launch_bounds__(256, 4) void xcorr_fused_long(const FusedParams p)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    using namespace lng;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;     // a thread's 16 elements of a sweep: j + i S
    constexpr int CH = S / 256;   // chunks of 256 threads x 16 elements per sweep
    constexpr int R1 = n / 4096;  // rows
    constexpr int Q1 = 16 / R1;   // butterflies of radix R1 per thread and chunk
    constexpr int NW = 4;         // waves
    static_assert(LOGN >= 14 && LOGN <= 16, "n = 16384, 32768, 65536");
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[4 * NW + 2 * NW + 2];
    __shared__ int redi[2 * NW];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + XW * wave;
    double2 *const Y = p.gscratch + (size_t)blockIdx.x * (size_t)n * (MULTI ? 2 : 1); // the workgroup's slice (MULTI: two)
    double2 *const Y2 = MULTI ? Y + n : Y; // where the second transforms land and sweep 2 reads
    const int N = PADDED ? p.N : n, pad = n - N;
    const double invN = PADDED ? p.invN : 1.0 / (double)n, invNm1 = PADDED ? p.invNm1 : 1.0 / (double)(n - 1); // (the launcher's quotients: scalar registers)
    const double2 *__restrict__ twl = p.twl; // [4096] W_n^(m2)
    // addresses: a scalar base formed where it is used plus one 32-bit lane offset (nothing 64-bit per lane, nothing
    // hoisted out of the loops into registers the transforms need)
    typedef d2v __attribute__((address_space(1))) *gd2;
    const auto yat = [&](long long off) __attribute__((always_inline)) { return (gd2)scalar_ptr_at(Y, off); };
    const auto y2at = [&](long long off) __attribute__((always_inline)) { return (gd2)scalar_ptr_at(Y2, off); };
    const auto opaque = [](int x) __attribute__((always_inline)) {
        asm volatile("" : "+v"(x));
        return x;
    };
    // the sweeps' factors W_n^(m2 k1): one table entry W_n^(m2) per element, its powers formed in registers (long_device.h)
    const auto tw_base = [&](int m, unsigned jj) __attribute__((always_inline)) { return ldg2u(scalar_ptr_at(twl, m * S), jj); };
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();

    for (long long pair = blockIdx.x; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        const double *__restrict__ ra = p.rows + rA * p.stride;
        const double *__restrict__ rb = p.rows + (hasB ? rA + 1 : rA) * p.stride;
        const double KA = ra[0], KB = rb[0];
        // ---------------- sweep 1
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = opaque(t + 256 * ch) & (S - 1);
            double2 v[16];
            // all 32 requests first (a request behind a consumer would wait for it: the address asm statements keep program
            // order), then the values in request order
            double xa[16], xb[16], c[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (PADDED && i >= 8) { // (pad < n / 2: always inside the row -- scalar base, no clamp)
                    xa[i] = __builtin_nontemporal_load(scalar_ptr_at(ra, (long long)i * S - pad) + (unsigned)j);
                    xb[i] = __builtin_nontemporal_load(scalar_ptr_at(rb, (long long)i * S - pad) + (unsigned)j);
                } else if (PADDED) {
                    const int e = j + i * S - pad;
                    const unsigned ec = (unsigned)(e < 0 ? 0 : e);
                    xa[i] = __builtin_nontemporal_load(scalar_ptr(ra) + ec);
                    xb[i] = __builtin_nontemporal_load(scalar_ptr(rb) + ec);
                } else {
                    xa[i] = __builtin_nontemporal_load(scalar_ptr_at(ra, i * S) + (unsigned)j);
                    xb[i] = __builtin_nontemporal_load(scalar_ptr_at(rb, i * S) + (unsigned)j);
                }
            }
            fence();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                double da = xa[i] - KA, db = xb[i] - KB;
                if (PADDED) {
                    const bool valid = i >= 8 || j + i * S - pad >= 0; // (pad < n / 2: the upper half is always data)
                    da = valid ? da : 0.0;
                    db = valid ? db : 0.0;
                }
                v[i] = make_double2(da, db);
                c[0] += da;
                c[1] = fma(da, da, c[1]);
                c[2] += db;
                c[3] = fma(db, db, c[3]);
            }
            // the chunk's sums leave the vector registers before the butterflies: reduced over the wave and added to the wave's
            // running sums in SGPRs (as per-lane accumulators across the chunks they were what the allocator parked in scratch
            // around every chunk's transform)
#pragma unroll
            for (int k = 0; k < 4; k++)
                q[k] = uniform(q[k] + wave_sum_dpp(c[k]));
            // the sweep's twiddles W_n^(m2 k1): the Q1 base entries travel behind the row requests
            double2 wb[Q1];
            {
                const unsigned jw = (unsigned)(opaque(t + 256 * ch) & (S - 1));
#pragma unroll
                for (int m = 0; m < Q1; m++)
                    wb[m] = tw_base(m, jw);
            }
            fence();
            sweep_dft<R1>(v);
            const unsigned js = (unsigned)(opaque(t + 256 * ch) & (S - 1));
#pragma unroll
            for (int m = 0; m < Q1; m++) { // row 0: no twiddle
                SLICE_ST(yat((long long)m * S) + js, (d2v{v[m].x, v[m].y}));
            }
#pragma unroll
            for (int m = 0; m < Q1; m++) {
                // element m2 = j + m S of row k1: register m + brev(k1) Q1, position j + (m + k1 Q1) S
                twiddle_powers<R1>(wb[m], [&](const int k1, const double2 w) __attribute__((always_inline)) {
                    const double2 z = cmul(v[m + brev<R1>(k1) * Q1], w);
                    SLICE_ST(yat((long long)(m + k1 * Q1) * S) + js, (d2v{z.x, z.y}));
                });
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (lane == 0)
                red[4 * wave + k] = q[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double w = red[k];
#pragma unroll
            for (int x = 1; x < NW; x++)
                w += red[4 * x + k];
            q[k] = w;
        }
        q[0] = readlane_f64(q[0], 0); // (workgroup-uniform: into SGPRs, not four VGPR pairs across the row transforms)
        q[1] = readlane_f64(q[1], 0);
        q[2] = readlane_f64(q[2], 0);
        q[3] = readlane_f64(q[3], 0);
        bool zeroA, nanA, zeroB, nanB;
        const double varA = variance(Stat{q[0], q[1]}, invN, invNm1, zeroA, nanA);
        const double varB = variance(Stat{q[2], q[3]}, invN, invNm1, zeroB, nanB);
        const double mA = q[0] * invN, mB = q[2] * invN;
        __syncthreads(); // the slice is complete (and `red` is free again)
        // ---------------- rows
        if (MULTI) { // first transforms once: the row's spectrum back into its row, register r of thread t at t + 256 r
#pragma clang loop unroll(disable)
            for (int k1 = 0; k1 < R1; k1++) {
                double2 *const row = Y + k1 * 4096;
                double2 v[16];
                const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const d2v z = SLICE_LD((gd2)scalar_ptr_at(row, 256 * i) + tl);
                    v[i] = make_double2(z.x, z.y);
                }
                row_forward(v, xbuf, xw, g2s, p.g3a, opaque(t), wave, !PADDED && k1 == 0);
                const unsigned ts = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int r = 0; r < 16; r++)
                    SLICE_ST((gd2)scalar_ptr_at(row, 256 * r) + ts, (d2v{v[r].x, v[r].y}));
            }
        }
        const int R = MULTI ? p.R : 1;
#pragma clang loop unroll(disable)
        for (int ref = 0; ref < R; ref++) {
        const double2 *__restrict__ xcp = MULTI ? uniform_ptr(p.xcp_many[ref]) : p.xcp;
        const double *__restrict__ c1t = !PADDED ? nullptr : MULTI ? uniform_ptr(p.c1_many[ref]) : p.c1;
        double *const mv_out = MULTI ? uniform_ptr(p.mv_many[ref]) : p.mv;
        int *const lag_out = MULTI ? uniform_ptr(p.lag_many[ref]) : p.lag;
#pragma clang loop unroll(disable)
        for (int k1 = 0; k1 < R1; k1++) {
            double2 *const row = Y + k1 * 4096;
            double2 *const row2 = Y2 + k1 * 4096;
            double2 v[16];
            {
                const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    // (MULTI: the spectrum is read once per reference -- plain loads, the line may still be in L2 for the next one)
                    const d2v z = MULTI ? *((gd2)scalar_ptr_at(row, 256 * i) + tl) : SLICE_LD((gd2)scalar_ptr_at(row, 256 * i) + tl);
                    v[i] = make_double2(z.x, z.y);
                }
            }
            if (MULTI)
                row_second(v, xbuf, xw, g2s, p.g3b, xcp + k1 * 4096, opaque(t), wave);
            else
                row_transforms(v, xbuf, xw, g2s, p.g3a, p.g3b, xcp + k1 * 4096, opaque(t), wave, !PADDED && k1 == 0);
            {
                const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int m = 0; m < 16; m++)
                    SLICE_ST((gd2)scalar_ptr_at(row2, 256 * m) + tl, (d2v{v[BR16(m)].x, v[BR16(m)].y}));
            }
        }
        __syncthreads();
        // ---------------- sweep 2 with the running argmax (maxAbsIndex, xcorr.go:39-50: first index of the greatest |cc|)
        double ma = 0.0, mb = 0.0, sa = 0.0, sb = 0.0, cc0a = 0.0, cc0b = 0.0;
        int ia = 0x7fffffff, ib = 0x7fffffff;
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < CH; ch++) {
            const int j = opaque(t + 256 * ch) & (S - 1);
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const d2v z = SLICE_LD(y2at((long long)i * S) + (unsigned)j);
                v[i] = make_double2(z.x, z.y);
            }
            {
                double2 wb[Q1];
#pragma unroll
                for (int m = 0; m < Q1; m++)
                    wb[m] = tw_base(m, (unsigned)j);
                fence();
#pragma unroll
                for (int m = 0; m < Q1; m++)
                    twiddle_powers<R1>(wb[m], [&](const int k1, const double2 w) __attribute__((always_inline)) {
                        v[m + k1 * Q1] = cmul(v[m + k1 * Q1], w);
                    });
            }
            sweep_dft<R1>(v);
            // the chunk's first maximum per lane (ascending i = ascending lag index: strictly greater keeps the first) ...
            double csa = 0.0, csb = 0.0;
            int cia = 0, cib = 0;
            const int jc = opaque(t + 256 * ch) & (S - 1);
#pragma unroll
            for (int i = 0; i < 16; i++) { // i = m + l1 Q1: lag index j + i S
                const int m = i % Q1, l1 = i / Q1;
                double2 c = v[m + brev<R1>(l1) * Q1];
                if (PADDED) {
                    const double c1 = scalar_ptr_at(c1t, i * S)[(unsigned)jc];
                    c = make_double2(fma(-mA, c1, c.x), fma(-mB, c1, c.y));
                }
                if (i == 0) { // (lane 0 of chunk 0: cc[0], the value reported when nothing is above 0)
                    cc0a = ch == 0 ? c.x : cc0a;
                    cc0b = ch == 0 ? c.y : cc0b;
                }
                const bool ga = fabs(c.x) > fabs(csa), gb = fabs(c.y) > fabs(csb);
                csa = ga ? c.x : csa;
                cia = ga ? i : cia;
                csb = gb ? c.y : csb;
                cib = gb ? i : cib;
            }
            // ... merged into the lane's running one (chunks are not in index order: ties go to the lower index)
            {
                const int xa_ = jc + cia * S, xb_ = jc + cib * S;
                const double ca_ = fabs(csa), cb_ = fabs(csb);
                const bool ta = (ca_ > ma) | ((ca_ == ma) & (ca_ > 0.0) & (xa_ < ia));
                const bool tb = (cb_ > mb) | ((cb_ == mb) & (cb_ > 0.0) & (xb_ < ib));
                ma = ta ? ca_ : ma;
                sa = ta ? csa : sa;
                ia = ta ? xa_ : ia;
                mb = tb ? cb_ : mb;
                sb = tb ? csb : sb;
                ib = tb ? xb_ : ib;
            }
        }
        // ---------------- workgroup argmax (first index of the maximum); the owner thread stores
        {
            constexpr int RM = 4 * NW, RC = 4 * NW + 2 * NW; // red: [RM + x] wave maxima A, [RM + NW + x] B, [RC], [RC + 1] cc[0]
            const double wa = wave_max(ma), wb = wave_max(mb);
            if (lane == 0) {
                red[RM + wave] = wa;
                red[RM + NW + wave] = wb;
            }
            if (t == 0) {
                red[RC] = cc0a;
                red[RC + 1] = cc0b;
            }
            __syncthreads();
            double MA = red[RM], MB = red[RM + NW];
#pragma unroll
            for (int x = 1; x < NW; x++) {
                MA = fmax(MA, red[RM + x]);
                MB = fmax(MB, red[RM + NW + x]);
            }
            int ca = (ma == MA && MA > 0.0) ? ia : 0x7fffffff;
            int cb = (mb == MB && MB > 0.0) ? ib : 0x7fffffff;
            ca = wave_min_i(ca);
            cb = wave_min_i(cb);
            if (lane == 0) {
                redi[wave] = ca;
                redi[NW + wave] = cb;
            }
            __syncthreads();
            int IA = redi[0], IB = redi[NW];
#pragma unroll
            for (int x = 1; x < NW; x++) {
                IA = min(IA, redi[x]);
                IB = min(IB, redi[NW + x]);
            }
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++) {
                if (sidx == 1 && !hasB)
                    break;
                const int I = sidx ? IB : IA;
                const bool none = I == 0x7fffffff;
                const bool owner = none ? (t == 0) : ((sidx ? ib : ia) == I && (sidx ? mb : ma) == (sidx ? MB : MA));
                if (owner) {
                    const double var = sidx ? varB : varA;
                    const bool zero = sidx ? zeroB : zeroA, nan = sidx ? nanB : nanA;
                    double y = __builtin_amdgcn_rsq(var);
                    y = y * fma(-0.5 * var * y, y, 1.5);
                    y = y * fma(-0.5 * var * y, y, 1.5);
                    const int idx = none ? 0 : I;
                    double mv = (none ? red[RC + sidx] : (sidx ? sb : sa)) * y;
                    int lag = idx > n / 2 ? idx - n : idx;
                    if (zero) { mv = 0.0; lag = 0; }              // xcorr.go:166-167
                    if (nan) { mv = __builtin_nan(""); lag = 0; } // placeholder: the pair is redone
                    mv_out[rA + sidx] = mv;
                    lag_out[rA + sidx] = lag;
                }
            }
            // a NaN / Inf series poisons its partner through the shared transform, and sigmas too far apart cost the
            // smaller series its precision: such pairs are redone by the kernel that isolates and rescales first
            if (ref == 0 && t == 0 && (nanA || (hasB && (nanB || sigma_spread_too_wide(varA, varB))))) { // (listed once per pair)
                const int slot = atomicAdd(p.ovf_count, 1);
                p.ovf_list[slot] = pair;
            }
            __syncthreads();
        }
        } // (references)
    }
}

template <int LOGN>
static hipError_t launch_long_n(const FusedParams &p, int num_cus, hipStream_t stream)
{
    const long long grid = std::min<long long>(p.npairs, (long long)num_cus * LONG_WGS_PER_CU);
    if (p.R > 1) { // many references in one pass: two n-element slices per workgroup
        if (2 * grid > p.gscratch_slices || !p.xcp_many || !p.mv_many || !p.lag_many || (p.N < (1 << LOGN) && !p.c1_many))
            return hipErrorInvalidValue;
        if (p.N < (1 << LOGN))
            hipLaunchKernelGGL((xcorr_fused_long<LOGN, true, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_fused_long<LOGN, false, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if (grid > p.gscratch_slices) // one n-element slice per workgroup
        return hipErrorInvalidValue;
    if (p.N < (1 << LOGN))
        hipLaunchKernelGGL((xcorr_fused_long<LOGN, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((xcorr_fused_long<LOGN, false>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// n = 16384, 32768, 65536 (float64 rows, every pair: no pair list); N in (n/2, n], N < n needs p.c1
hipError_t launch_fused_long(const FusedParams &p_in, int num_cus, hipStream_t stream)
{
    const FusedParams p = with_reciprocals(p_in);
    if (!p.rows || !p.gscratch || !p.twl || (!p.xcp && p.R <= 1) || !p.g2 || !p.g3a || !p.g3b || !p.ovf_list || !p.ovf_count || p.pair_list ||
        (p.N < p.n && !p.c1 && p.R <= 1))
        return hipErrorInvalidValue;
    switch (p.logn) {
    case 14: return launch_long_n<14>(p, num_cus, stream);
    case 15: return launch_long_n<15>(p, num_cus, stream);
    case 16: return launch_long_n<16>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace muse
