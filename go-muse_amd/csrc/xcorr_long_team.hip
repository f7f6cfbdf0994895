// xcorr_long_team.hip -- long series (n = 32768, 65536): the four-step transform of xcorr_long.hip with a TEAM of T = n / 4096
// workgroups per pair, so that a pair's scratch slice is written and read back within tens of microseconds and is served by the
// 256 MiB Infinity Cache instead of HBM.
//
// xcorr_long.hip keeps a pair inside one workgroup: 1 024 slices (1 GB) in flight, a slice line is read ~ 300 us and ~ 1.5 GB of
// other traffic after it was written, every crossing goes to HBM.  Here workgroup r of a team takes
// chunk r of sweep 1, row r of the rows stage and chunk r of sweep 2: a pair is done in 1 / T of the time, 64 (n = 65536) or 128
// (n = 32768) pairs are in flight, two slices per team (128 MB) and ~ 100 MB of traffic between a line's write and its read.
// What the team needs:
//   * all its workgroups resident at once: the grid is the resident set (occupancy query) rounded down to whole teams;
//   * two barriers per pair between workgroups on DIFFERENT CUs / XCDs: every slice byte is stored and loaded with sc1 (device
//     scope: written through to the memory side, never served from a stale L2 line of another XCD), each wave waits for its
//     stores, the workgroup barrier collects the waves, ONE lane adds to the team's counter (agent scope) and polls it with sc1
//     loads, a second workgroup barrier releases the others (MI355X_MICROARCH.md, cross-workgroup hand-offs: the first row of
//     the table);  the third synchronisation of a pair needs no spinning -- the workgroup whose add to the `done` counter came
//     last combines the team's argmax candidates and writes the pair's result;
//   * buffers that the next pair may write while a slow team mate still reads the current one's: slices, partial statistics and
//     candidates are double-buffered by the parity of the team's pair count (nobody can be two pairs ahead: the barriers);
//   * a way out: a poll that does not see its team arrive within ~ 1 s (it cannot happen while the grid is resident) gives up;
//     rank 0 of the team then lists the team's remaining pairs for the kernel that redoes NaN / sigma-spread pairs anyway.
// Semantics, tables and the redo path are those of xcorr_long.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "long_device.h"

namespace muse {

namespace lngt {

using namespace occ4;
using namespace fold;
using namespace foldk;
using namespace lng;

typedef int v4i __attribute__((ext_vector_type(4)));
// MUSE_TEAM_EXP (tools/ablate only; results may be wrong): bit 0 = slice traffic without sc1, bit 1 = the team barrier does not wait
#ifndef MUSE_TEAM_EXP
#define MUSE_TEAM_EXP 0
#endif
constexpr int SC1 = (MUSE_TEAM_EXP & 1) ? 0 : 16; // cache-policy bit of the buffer intrinsics: device scope
constexpr int WS_DOUBLES = 1024; // per team: [0] barrier counter, [16] done counter (own 128-byte lines), partials, candidates
constexpr int WS_PART = 32, WS_CAND = WS_PART + 2 * 16 * 4;
constexpr int SPIN_LIMIT = 1 << 19;

__device__ __forceinline__ int opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double2 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    const v4i x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, SC1);
    return make_double2(__hiloint2double(x.y, x.x), __hiloint2double(x.w, x.z));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const double2 v)
{
    const v4i x = {__double2loint(v.x), __double2hiint(v.x), __double2loint(v.y), __double2hiint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)byte_off, 0, SC1);
}

// every wave of the workgroup: own stores landed -> workgroup barrier -> one lane signals and waits for the team -> barrier.
// Returns false (workgroup-uniform) when the team did not arrive.
__device__ __forceinline__ bool team_barrier(unsigned *ctr, const unsigned target, int *flag_lds)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (!(MUSE_TEAM_EXP & 2) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < SPIN_LIMIT) {
            __builtin_amdgcn_s_sleep(16);
            spins++;
        }
        *flag_lds = spins < SPIN_LIMIT;
    }
    __syncthreads();
    return *flag_lds != 0;
}

} // namespace lngt

template <int LOGN, bool PADDED>
__global__ __launch_bounds__(256, 4) void xcorr_fused_long_team(const FusedParams p)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    using namespace lng;
    using namespace lngt;
    constexpr int n = 1 << LOGN, S = n / 16, R1 = n / 4096, Q1 = 16 / R1;
    constexpr int TEAM = R1;                       // = chunks per sweep = rows
    constexpr int TF = Q1 * (R1 - 1), NB = (TF + 3) / 4;
    static_assert(LOGN == 15 || LOGN == 16, "n = 32768, 65536");
    static_assert(S / 256 == TEAM, "one chunk per team member");
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[16];
    __shared__ int redi[8];
    __shared__ int flag;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int team = (int)blockIdx.x / TEAM, rank = (int)blockIdx.x % TEAM, nteams = (int)gridDim.x / TEAM;
    double *const ws = p.team_ws + (size_t)team * WS_DOUBLES;
    unsigned *const bar_ctr = (unsigned *)ws, *const done_ctr = (unsigned *)(ws + 16);
    const __amdgpu_buffer_rsrc_t wsr = rsrc_of(ws);
    const int N = PADDED ? p.N : n, pad = n - N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);
    const double2 *__restrict__ twl = p.twl;
    const auto tw_load = [&](int f, unsigned jj) __attribute__((always_inline)) {
        const int m = f / (R1 - 1), k1 = 1 + f % (R1 - 1);
        return ldg2u(scalar_ptr_at(twl, k1 * 4096 + m * S), jj);
    };
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();

    unsigned epoch = 0; // team barriers passed
    int parity = 0;
    long long pair = team;
    bool alive = true;
    for (; pair < p.npairs; pair += nteams, parity ^= 1) {
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;
        const double *__restrict__ ra = p.rows + rA * p.stride;
        const double *__restrict__ rb = p.rows + (hasB ? rA + 1 : rA) * p.stride;
        const __amdgpu_buffer_rsrc_t yr = rsrc_of(p.gscratch + ((size_t)team * 2 + parity) * (size_t)n);
        const unsigned part_off = (unsigned)((WS_PART + parity * 16 * 4) * sizeof(double));
        const unsigned cand_off = (unsigned)((WS_CAND + parity * 16 * 8) * sizeof(double));
        // ---------------- sweep 1, chunk `rank`
        {
            const double KA = ra[0], KB = rb[0];
            const int j = opaque(t + 256 * rank) & (S - 1);
            double2 v[16];
            double xa[16], xb[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (PADDED && i >= 8) { // (pad < n / 2: always inside the row)
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
            double2 wq[2][4];
#pragma unroll
            for (int f = 0; f < 4 && f < TF; f++)
                wq[0][f] = tw_load(f, (unsigned)j);
            fence();
            double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                double da = xa[i] - KA, db = xb[i] - KB;
                if (PADDED) {
                    const bool valid = i >= 8 || j + i * S - pad >= 0;
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
            const unsigned js = (unsigned)(opaque(t + 256 * rank) & (S - 1));
#pragma unroll
            for (int m = 0; m < Q1; m++)
                st_sc1(yr, (unsigned)((m * S + js) * sizeof(double2)), v[m]);
#pragma unroll
            for (int bt = 0; bt < NB; bt++) {
                fence();
                if (bt + 1 < NB) {
#pragma unroll
                    for (int f = 4 * (bt + 1); f < 4 * (bt + 2) && f < TF; f++)
                        wq[(bt + 1) & 1][f & 3] = tw_load(f, js);
                }
                fence();
#pragma unroll
                for (int f = 4 * bt; f < 4 * (bt + 1) && f < TF; f++) {
                    const int m = f / (R1 - 1), k1 = 1 + f % (R1 - 1);
                    st_sc1(yr, (unsigned)(((m + k1 * Q1) * S + js) * sizeof(double2)), cmul(v[m + brev<R1>(k1) * Q1], wq[bt & 1][f & 3]));
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double w = wave_sum_dpp(q[k]);
                if (lane == 0)
                    red[4 * wave + k] = w;
            }
            __syncthreads();
            if (t < 2) // this chunk's partial sums: (sum dA, sum dA^2) / (sum dB, sum dB^2) as two 16-byte stores
                st_sc1(wsr, part_off + (unsigned)((rank * 4 + 2 * t) * sizeof(double)),
                       make_double2((red[2 * t] + red[4 + 2 * t]) + (red[8 + 2 * t] + red[12 + 2 * t]),
                                    (red[2 * t + 1] + red[5 + 2 * t]) + (red[9 + 2 * t] + red[13 + 2 * t])));
        }
        epoch++;
        if (!(alive = team_barrier(bar_ctr, epoch * TEAM, &flag)))
            break;
        // ---------------- row `rank`
        {
            double2 v[16];
            const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = ld_sc1(yr, (unsigned)((rank * 4096 + 256 * i + tl) * sizeof(double2)));
            row_transforms(v, xbuf, xbuf + XW * wave, g2s, p.g3a, p.g3b, p.xcp + rank * 4096, t, wave, !PADDED && rank == 0);
            const unsigned ts = (unsigned)(opaque(t) & 255);
#pragma unroll
            for (int m = 0; m < 16; m++)
                st_sc1(yr, (unsigned)((rank * 4096 + 256 * m + ts) * sizeof(double2)), v[BR16(m)]);
        }
        epoch++;
        if (!(alive = team_barrier(bar_ctr, epoch * TEAM, &flag)))
            break;
        // ---------------- the pair's statistics: every member sums the team's partials (workgroup-uniform values)
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < TEAM; c++) {
            const double2 a = ld_sc1(wsr, part_off + (unsigned)((c * 4) * sizeof(double)));
            const double2 b = ld_sc1(wsr, part_off + (unsigned)((c * 4 + 2) * sizeof(double)));
            q[0] += a.x;
            q[1] += a.y;
            q[2] += b.x;
            q[3] += b.y;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            q[k] = readlane_f64(q[k], 0);
        const double mA = q[0] * invN, mB = q[2] * invN;
        // ---------------- sweep 2, chunk `rank`: the chunk's first maximum -> candidate
        {
            const int j = opaque(t + 256 * rank) & (S - 1);
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = ld_sc1(yr, (unsigned)((i * S + j) * sizeof(double2)));
            {
                double2 wq[2][4];
#pragma unroll
                for (int f = 0; f < 4 && f < TF; f++)
                    wq[0][f] = tw_load(f, (unsigned)j);
#pragma unroll
                for (int bt = 0; bt < NB; bt++) {
                    fence();
                    if (bt + 1 < NB) {
#pragma unroll
                        for (int f = 4 * (bt + 1); f < 4 * (bt + 2) && f < TF; f++)
                            wq[(bt + 1) & 1][f & 3] = tw_load(f, (unsigned)j);
                    }
                    fence();
#pragma unroll
                    for (int f = 4 * bt; f < 4 * (bt + 1) && f < TF; f++) {
                        const int m = f / (R1 - 1), k1 = 1 + f % (R1 - 1);
                        v[m + k1 * Q1] = cmul(v[m + k1 * Q1], wq[bt & 1][f & 3]);
                    }
                }
            }
            sweep_dft<R1>(v);
            double csa = 0.0, csb = 0.0, cc0a = 0.0, cc0b = 0.0;
            int cia = 0, cib = 0;
            const int jc = opaque(t + 256 * rank) & (S - 1);
#pragma unroll
            for (int i = 0; i < 16; i++) { // i = m + l1 Q1: lag index j + i S, ascending
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
            // candidate record of the chunk: {|max| A, signed A}, {index A, |max| B}, {signed B, index B}, {cc[0] A, cc[0] B}
            const unsigned co = cand_off + (unsigned)(rank * 8 * sizeof(double));
            const bool ownA = IA == 0x7fffffff ? t == 0 : (ia == IA && ma == MA);
            const bool ownB = IB == 0x7fffffff ? t == 0 : (ib == IB && mb == MB);
            // (the two owners may be different threads: each stores its series' three values as 8-byte pieces through LDS first)
            if (ownA) {
                red[8] = IA == 0x7fffffff ? 0.0 : MA;
                red[9] = csa;
                red[10] = (double)IA;
            }
            if (ownB) {
                red[11] = IB == 0x7fffffff ? 0.0 : MB;
                red[12] = csb;
                red[13] = (double)IB;
            }
            if (t == 0) {
                red[14] = cc0a;
                red[15] = cc0b;
            }
            __syncthreads();
            if (t < 4)
                st_sc1(wsr, co + (unsigned)(2 * t * sizeof(double)), make_double2(red[8 + 2 * t], red[9 + 2 * t]));
        }
        // ---------------- the member whose `done` add came last combines the candidates and writes the result
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0)
            flag = (int)(__hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % TEAM);
        __syncthreads();
        if (flag == TEAM - 1 && t == 0) {
            bool zero[2], nan[2];
            double var[2];
            var[0] = variance(Stat{q[0], q[1]}, invN, invNm1, zero[0], nan[0]);
            var[1] = variance(Stat{q[2], q[3]}, invN, invNm1, zero[1], nan[1]);
            double cc0[2] = {0.0, 0.0};
            double best[2] = {0.0, 0.0}, bsv[2] = {0.0, 0.0};
            int bidx[2] = {0x7fffffff, 0x7fffffff};
            for (int c = 0; c < TEAM; c++) {
                const unsigned co = cand_off + (unsigned)(c * 8 * sizeof(double));
                const double2 r0 = ld_sc1(wsr, co), r1 = ld_sc1(wsr, co + 16), r2 = ld_sc1(wsr, co + 32), r3 = ld_sc1(wsr, co + 48);
                const double m[2] = {r0.x, r1.y}, sv[2] = {r0.y, r2.x};
                const int ix[2] = {(int)r1.x, (int)r2.y};
                if (c == 0) {
                    cc0[0] = r3.x;
                    cc0[1] = r3.y;
                }
                for (int s = 0; s < 2; s++)
                    if (m[s] > best[s] || (m[s] == best[s] && m[s] > 0.0 && ix[s] < bidx[s])) {
                        best[s] = m[s];
                        bsv[s] = sv[s];
                        bidx[s] = ix[s];
                    }
            }
            for (int s = 0; s < (hasB ? 2 : 1); s++) {
                const bool none = !(best[s] > 0.0);
                double y = __builtin_amdgcn_rsq(var[s]);
                y = y * fma(-0.5 * var[s] * y, y, 1.5);
                y = y * fma(-0.5 * var[s] * y, y, 1.5);
                const int idx = none ? 0 : bidx[s];
                double mv = (none ? cc0[s] : bsv[s]) * y;
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
        __syncthreads(); // (`flag`, `red`)
    }
    // the team did not arrive (never while the whole grid is resident): its remaining pairs go to the kernel that redoes pairs
    if (!alive && rank == 0 && t == 0) {
        for (; pair < p.npairs; pair += nteams) {
            const int slot = atomicAdd(p.ovf_count, 1);
            p.ovf_list[slot] = pair;
        }
    }
}

template <int LOGN>
static hipError_t launch_team_n(const FusedParams &p, int num_cus, hipStream_t stream)
{
    constexpr int TEAM = (1 << LOGN) / 4096;
    const bool padded = p.N < (1 << LOGN);
    int occ = 0;
    hipError_t e = padded ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, xcorr_fused_long_team<LOGN, true>, 256, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, xcorr_fused_long_team<LOGN, false>, 256, 0);
    if (e != hipSuccess)
        return e;
    long long teams = std::min<long long>((long long)num_cus * std::min(occ, 4) / TEAM, p.team_cap);
    teams = std::min<long long>(teams, p.npairs);
    if (teams < 1)
        return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(teams * TEAM);
    if (padded)
        hipLaunchKernelGGL((xcorr_fused_long_team<LOGN, true>), dim3(grid), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((xcorr_fused_long_team<LOGN, false>), dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// n = 32768, 65536 (float64 rows, every pair: no pair list); N in (n/2, n], N < n needs p.c1.  p.team_ws: team_cap x 1024 zeroed
// doubles (counters, partial statistics, candidates); p.gscratch: team_cap x 2 slices of n complex.
hipError_t launch_fused_long_team(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (!p.rows || !p.gscratch || !p.twl || !p.xcp || !p.g2 || !p.g3a || !p.g3b || !p.ovf_list || !p.ovf_count || p.pair_list ||
        !p.team_ws || p.team_cap < 1 || (p.N < p.n && !p.c1))
        return hipErrorInvalidValue;
    switch (p.logn) {
    case 15: return launch_team_n<15>(p, num_cus, stream);
    case 16: return launch_team_n<16>(p, num_cus, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace muse
