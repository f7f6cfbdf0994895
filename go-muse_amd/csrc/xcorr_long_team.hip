// xcorr_long_team.hip -- long series, n = 32768 and 65536: xCorrWithX (/root/reference/xcorr.go:160-197) as the four-step
// transform of xcorr_long.hip, with its two exchanges kept inside ONE XCD's L2 (round 4 experiment).
//
// xcorr_long.hip gives every workgroup a pair of its own and an n-element scratch slice: 1 024 slices (1 GB at n = 65536) are in
// flight, so each of the four crossings of a slice (2 writes + 2 reads of 16 n bytes) goes to HBM: 5.7 x the algorithmic bytes.
// Here the unit of work is a TASK -- one chunk of sweep 1, one 4096-point row, one chunk of sweep 2 (R1 = n / 4096 of each per
// pair) -- and the workgroups of one XCD share a short ring of slices (NS per XCD, <= 3-6 MB against 4 MiB of L2):
//   * every workgroup reads the id of the XCD it runs on (HW_REG_XCC_ID: a hardware fact, not an assumption about dispatch) and
//     draws tickets from THAT XCD's counter; ticket tau -> round u = tau / 3 R1, task (phase, idx) of sequence u - phase: a round
//     holds sweep 1 of sequence u, the rows of sequence u - 1 and sweep 2 of sequence u - 2, interleaved, so that the phases of
//     three pairs overlap on the XCD's 32 CUs and every dependency of a task was drawn a whole round earlier;
//   * all tasks of a sequence run on one XCD, so producer and consumer share one L2: slice bytes are written with PLAIN stores (they
//     stay in that L2; sc1 would write them through and drop the line) and read with L1-bypassing loads; no agent-scope release
//     (buffer_wbl2) and no L1 invalidate is needed for them.  Hand-offs: every storing wave drains its stores (s_waitcnt vmcnt(0)),
//     the workgroup's barrier, then ONE lane adds to the sequence's cumulative counter; the consumer's one lane polls that counter
//     (sc1 loads), then the workgroup's barrier, then the loads;
//   * a task only ever waits for tasks with SMALLER tickets of its own XCD, all of which were drawn by running workgroups: no
//     assumption about co-residency, dispatch order or the number of workgroups an XCD receives.  Every wait is bounded (a
//     time-out sets TeamCtl::error, after which no wait blocks any more and the host reports the failure);
//   * statistics (per-chunk partial sums) and the argmax (per-chunk records) are combined in a fixed order by the workgroup that
//     finishes a sequence's last sweep-2 chunk: results do not depend on timing.
// Pairs are handed to XCDs by one global counter.  NaN / Inf and sigma-spread pairs are listed for the rescaling kernel as in
// xcorr_long.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "long_device.h"

namespace muse {

#ifndef MUSE_TEAM_AGENT_SCOPE_COUNTERS
#define MUSE_TEAM_AGENT_SCOPE_COUNTERS 0
#endif
namespace team {

constexpr int XCDS = 8;       // MI355X: 8 XCDs (a workgroup that reads another id reports an error and leaves)
constexpr int MAX_SLOTS = 8;  // slices per XCD the control block has room for
constexpr unsigned SPIN_MAX = 1u << 21;

struct alignas(128) Slot {
    unsigned long long tag; // (sequence + 1) << 32 | (pair + 1) of the pair that owns the slot
    unsigned long long cnt; // cumulative counts: finished sweep-1 chunks (low word) and rows (high word) -- one load reads both
    unsigned s2, freed;     // finished sweep-2 chunks, finished sequences
    char pad[128 - 24];
    double stat[16][4]; // sweep 1, per chunk: sum dA, sum dA^2, sum dB, sum dB^2
    double part[16][8]; // sweep 2, per chunk: max |ccA|, its signed value, its index, the same for B, cc[0] of A and B (chunk 0)
};
struct alignas(128) Xcd {
    unsigned long long ticket;
    char pad0[120];
    // low word: sequences 0 .. decided - 1 have been given a pair or found none; high word: 1 + the first sequence that found
    // no pair (0: none yet) -- every later one finds none either.  One writer at a time (the sequences are decided in order).
    unsigned long long state;
    char pad1[120];
    Slot slot[MAX_SLOTS];
};
struct alignas(128) Ctl {
    unsigned long long next_pair;
    unsigned error; // 1: a wait timed out, 2: an XCC id outside 0 .. XCDS - 1
    char pad[128 - 12];
    Xcd xcd[XCDS];
};
static_assert(sizeof(Slot) % 128 == 0 && sizeof(Xcd) % 128 == 0, "control records on lines of their own");

// Control words.  Loads: relaxed agent-scope (sc1: never served by a CU's L1).  The counters of an XCD's own record are only
// ever touched from that XCD, so their read-modify-writes need no wider scope than its L2, where every atomic executes anyway:
// a workgroup-scope add stays in the L2 the polling loads read (an agent-scope one goes out to the fabric and drops the line:
// MI355X_MICROARCH.md, stores of each flavour).  The pair counter all XCDs share is agent-scope.
template <typename T>
__device__ __forceinline__ T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T add_agent(T *p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#if MUSE_TEAM_AGENT_SCOPE_COUNTERS
constexpr int OWN_SCOPE = __HIP_MEMORY_SCOPE_AGENT;
#else
constexpr int OWN_SCOPE = __HIP_MEMORY_SCOPE_WORKGROUP;
#endif
template <typename T>
__device__ __forceinline__ T add(T *p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, OWN_SCOPE); }
template <typename T>
__device__ __forceinline__ void st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, OWN_SCOPE); }
__device__ __forceinline__ double ldf(const double *p)
{
    return __longlong_as_double((long long)ld((const unsigned long long *)p));
}

// one lane waits for cond(); false after a time-out (or once any workgroup has reported one)
template <typename F>
__device__ __forceinline__ bool spin_until(Ctl *ctl, F cond)
{
    for (unsigned it = 0; it < SPIN_MAX; it++) {
        if (cond())
            return true;
        if ((it & 255) == 255 && ld(&ctl->error))
            return false;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_fetch_or(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

} // namespace team

template <int LOGN, bool PADDED>
__global__ __launch_bounds__(256, 4) void xcorr_long_team(const FusedParams p)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    using namespace lng;
    using namespace team;
    constexpr int n = 1 << LOGN;
    constexpr int S = n / 16;    // a thread's 16 elements of a sweep: j + i S
    constexpr int R1 = n / 4096; // rows = chunks of 256 threads x 16 elements per sweep
    constexpr int Q1 = 16 / R1;
    constexpr int NW = 4;
    constexpr int TASKS = 3 * R1; // per round
    const int D = p.team_dist;    // a round holds sweep 1 of sequence u, the rows of u - D, sweep 2 of u - 2 D
    static_assert(LOGN == 15 || LOGN == 16, "n = 32768, 65536");
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[4 * NW + 2 * NW + 2];
    __shared__ int redi[2 * NW];
    __shared__ long long bc[4]; // the task, as drawn and resolved by lane 0
    __shared__ double bcd[4];   // sweep 2: the pair's statistics (sums over the chunks in chunk order)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + XW * wave;
    const int N = PADDED ? p.N : n, pad = n - N;
    const double invN = 1.0 / (double)N, invNm1 = 1.0 / (double)(N - 1);
    const double2 *__restrict__ twl = p.twl; // [4096] W_n^(m2)
    Ctl *const ctl = (Ctl *)p.team_ctl;
    const int NS = p.team_slots;
    // hwreg(HW_REG_XCC_ID = 20, offset 0, width 4): the XCD this workgroup runs on
    const int xcc = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u));
    if (xcc >= XCDS) {
        if (t == 0)
            __hip_atomic_fetch_or(&ctl->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    Xcd *const X = &ctl->xcd[xcc];
    double2 *const slices = p.gscratch + (size_t)xcc * (size_t)NS * (size_t)n;
    typedef d2v __attribute__((address_space(1))) *gd2;
    const auto opaque = [](int x) __attribute__((always_inline)) {
        asm volatile("" : "+v"(x));
        return x;
    };
    const auto tw_base = [&](int m, unsigned jj) __attribute__((always_inline)) { return ldg2u(scalar_ptr_at(twl, m * S), jj); };
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();

    unsigned long long tau_next = 0; // lane 0: the next ticket, drawn while the current task runs
    if (t == 0)
        tau_next = add(&X->ticket, 1ull);
    for (;;) {
        // ---------------- lane 0 takes its ticket, decides the round's new sequence if it holds the round's first ticket, resolves
        // the task's sequence to a pair and waits for what the task depends on: typically two round trips to L2 (the XCD's state
        // word; the slot's tag and counters together)
        if (t == 0) {
            const unsigned long long tau = tau_next;
            const long long u = (long long)(tau / TASKS);
            const int w = (int)(tau % TASKS), phase = w % 3, idx = w / 3;
            bool ok = true;
            unsigned long long state = 0;
            if (w == 0) { // sequence u: in sequence order (so that "found none" is monotonic per XCD), into a free slot
                Slot *sl = &X->slot[u % NS];
                ok = spin_until(ctl, [&]() { state = ld(&X->state); return (state & 0xffffffffull) >= (unsigned long long)u; });
                unsigned long long fe = state >> 32;
                if (ok && fe == 0) {
                    ok = spin_until(ctl, [&]() { return ld(&sl->freed) >= (unsigned)(u / NS); });
                    const unsigned long long pair = add_agent(&ctl->next_pair, 1ull);
                    if (pair < (unsigned long long)p.npairs) {
                        st(&sl->tag, ((unsigned long long)(u + 1) << 32) | (pair + 1));
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else
                        fe = (unsigned long long)(u + 1);
                }
                state = (fe << 32) | (unsigned long long)(u + 1);
                st(&X->state, state);
            }
            const long long s = u - (long long)D * phase; // the task's sequence
            long long pair = -1;
            bool leave = false;
            if (s >= 0) {
                if ((state & 0xffffffffull) <= (unsigned long long)s)
                    ok = ok && spin_until(ctl, [&]() { state = ld(&X->state); return (state & 0xffffffffull) > (unsigned long long)s; });
                const unsigned long long fe = state >> 32;
                const auto found_none = [&](long long s_) { return fe != 0 && fe - 1 <= (unsigned long long)s_; };
                if (ok && !found_none(s)) {
                    const Slot *sl = &X->slot[s % NS];
                    const unsigned long long tag = ld(&sl->tag);
                    unsigned long long cnt = ld(&sl->cnt);
                    pair = (long long)(tag & 0xffffffffull) - 1;
                    const unsigned gen1 = (unsigned)(s / NS + 1) * R1;
                    if (phase == 1 && (unsigned)cnt < gen1)
                        ok = spin_until(ctl, [&]() { return (unsigned)ld(&sl->cnt) >= gen1; });
                    else if (phase == 2 && (unsigned)(cnt >> 32) < gen1)
                        ok = spin_until(ctl, [&]() { return (unsigned)(ld(&sl->cnt) >> 32) >= gen1; });
                }
                // leave when the OLDEST sequence of the round (decided before s) found no pair: every later one found none either
                leave = u >= 2 * D && found_none(u - 2 * D);
            }
            leave = leave || !ok;
            if (!leave)
                tau_next = add(&X->ticket, 1ull); // (in flight during the task: the value is read at the top of the next one)
            bc[0] = leave ? -2 : pair;
            bc[1] = s;
            bc[2] = phase;
            bc[3] = idx;
            if (PADDED && !leave && pair >= 0 && phase == 2) { // the series' means (the pad correction of sweep 2): the chunks'
                                                               // partial sums in chunk order
                const Slot *sl = &X->slot[s % NS];
                double qa = 0.0, qb = 0.0;
                for (int c = 0; c < R1; c++) {
                    qa += ldf(&sl->stat[c][0]);
                    qb += ldf(&sl->stat[c][2]);
                }
                bcd[0] = qa;
                bcd[2] = qb;
            }
        }
        __syncthreads();
        const long long pair = bc[0];
        const long long seq = bc[1];
        const int phase = __builtin_amdgcn_readfirstlane((int)bc[2]), idx = __builtin_amdgcn_readfirstlane((int)bc[3]);
        const double q0 = bcd[0], q2 = bcd[2];
        __syncthreads(); // (bc / bcd are rewritten by the next draw)
        if (pair == -2)
            break;
        if (pair < 0)
            continue; // a task of a sequence that found no pair
        Slot *const sl = &X->slot[seq % NS];
        double2 *const Y = slices + (size_t)(seq % NS) * (size_t)n;
        const auto yat = [&](long long off) __attribute__((always_inline)) { return (gd2)scalar_ptr_at(Y, off); };
        const long long rA = 2 * pair;
        const bool hasB = rA + 1 < p.M;

        if (phase == 0) {
            // ---------------- sweep 1, chunk idx: rows of the group -> d = x - K, partial statistics, radix R1 over m1, twiddle ->
            // the slice Y[k1][m2]
            const int ch = idx;
            const double *__restrict__ ra = p.rows + rA * p.stride;
            const double *__restrict__ rb = p.rows + (hasB ? rA + 1 : rA) * p.stride;
            const double KA = ra[0], KB = rb[0];
            const int j = opaque(t + 256 * ch) & (S - 1);
            double2 v[16];
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
            double2 wb[Q1];
            {
                const unsigned jw = (unsigned)(opaque(t + 256 * ch) & (S - 1));
#pragma unroll
                for (int m = 0; m < Q1; m++)
                    wb[m] = tw_base(m, jw);
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
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double w = wave_sum_dpp(c[k]);
                if (lane == 0)
                    red[4 * wave + k] = w;
            }
            fence();
            sweep_dft<R1>(v);
            const unsigned js = (unsigned)(opaque(t + 256 * ch) & (S - 1));
#pragma unroll
            for (int m = 0; m < Q1; m++) // row 0: no twiddle
                *(yat((long long)m * S) + js) = d2v{v[m].x, v[m].y};
#pragma unroll
            for (int m = 0; m < Q1; m++) {
                // element m2 = j + m S of row k1: register m + brev(k1) Q1, position j + (m + k1 Q1) S
                twiddle_powers<R1>(wb[m], [&](const int k1, const double2 w) __attribute__((always_inline)) {
                    const double2 z = cmul(v[m + brev<R1>(k1) * Q1], w);
                    *(yat((long long)(m + k1 * Q1) * S) + js) = d2v{z.x, z.y};
                });
            }
            __syncthreads(); // (red complete)
            if (t < 4)
                sl->stat[ch][t] = (red[t] + red[4 + t]) + (red[8 + t] + red[12 + t]);
            // hand-off: every wave's stores have left (they sit in this XCD's L2), then one add to the sequence's counter
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0)
                add(&sl->cnt, 1ull);
        } else if (phase == 1) {
            // ---------------- row idx: the n = 4096 kernel's pair of transforms, in place
            const int k1 = idx;
            double2 *const row = Y + k1 * 4096;
            double2 v[16];
            {
                const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const d2v z = __builtin_nontemporal_load((gd2)scalar_ptr_at(row, 256 * i) + tl);
                    v[i] = make_double2(z.x, z.y);
                }
            }
            row_transforms(v, xbuf, xw, g2s, p.g3a, p.g3b, p.xcp + k1 * 4096, t, wave, !PADDED && k1 == 0);
            {
                const unsigned tl = (unsigned)(opaque(t) & 255);
#pragma unroll
                for (int m = 0; m < 16; m++)
                    *((gd2)scalar_ptr_at(row, 256 * m) + tl) = d2v{v[BR16(m)].x, v[BR16(m)].y};
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0)
                add(&sl->cnt, 1ull << 32);
        } else {
            // ---------------- sweep 2, chunk idx: twiddle, radix R1 over k1 -> cc; (N < n: minus mean c1[lag]); the chunk's argmax
            const int ch = idx;
            const double mA = q0 * invN, mB = q2 * invN;
            const double *__restrict__ c1t = PADDED ? p.c1 : nullptr;
            const int j = opaque(t + 256 * ch) & (S - 1);
            double2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const d2v z = __builtin_nontemporal_load(yat((long long)i * S) + (unsigned)j);
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
            // the chunk's first maximum per lane (ascending i = ascending lag index: strictly greater keeps the first)
            double sa = 0.0, sb = 0.0, cc0a = 0.0, cc0b = 0.0;
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
                    cc0a = c.x;
                    cc0b = c.y;
                }
                const bool ga = fabs(c.x) > fabs(sa), gb = fabs(c.y) > fabs(sb);
                sa = ga ? c.x : sa;
                cia = ga ? i : cia;
                sb = gb ? c.y : sb;
                cib = gb ? i : cib;
            }
            const double ma = fabs(sa), mb = fabs(sb);
            const int ia = jc + cia * S, ib = jc + cib * S;
            {
                constexpr int RM = 4 * NW;
                const double wa = wave_max(ma), wbm = wave_max(mb);
                if (lane == 0) {
                    red[RM + wave] = wa;
                    red[RM + NW + wave] = wbm;
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
                // the chunk's record: {max |cc|, its signed value, its index} per series (nothing above 0: max 0, index "none")
                double *const rec = sl->part[ch];
                if (IA == 0x7fffffff ? t == 0 : (ia == IA && ma == MA)) {
                    rec[0] = IA == 0x7fffffff ? 0.0 : MA;
                    rec[1] = IA == 0x7fffffff ? 0.0 : sa;
                    rec[2] = (double)IA;
                }
                if (IB == 0x7fffffff ? t == 0 : (ib == IB && mb == MB)) {
                    rec[3] = IB == 0x7fffffff ? 0.0 : MB;
                    rec[4] = IB == 0x7fffffff ? 0.0 : sb;
                    rec[5] = (double)IB;
                }
                if (t == 0 && ch == 0) {
                    rec[6] = cc0a;
                    rec[7] = cc0b;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                const unsigned done = add(&sl->s2, 1u) + 1;
                if (done == (unsigned)(seq / NS + 1) * R1) {
                    // ---- the sequence's last chunk: combine the R1 records in chunk order (first index of the greatest |cc|:
                    // maxAbsIndex, xcorr.go:39-50), scale by 1 / sigma, write the pair's results, free the slot
                    double q[4] = {0.0, 0.0, 0.0, 0.0};
                    for (int c = 0; c < R1; c++)
                        for (int k = 0; k < 4; k++)
                            q[k] += ldf(&sl->stat[c][k]);
                    bool zero[2], nan[2];
                    double var[2];
                    var[0] = variance(Stat{q[0], q[1]}, invN, invNm1, zero[0], nan[0]);
                    var[1] = variance(Stat{q[2], q[3]}, invN, invNm1, zero[1], nan[1]);
                    for (int sidx = 0; sidx < (hasB ? 2 : 1); sidx++) {
                        double best = 0.0, bval = 0.0, bidx = 2147483647.0;
                        for (int c = 0; c < R1; c++) {
                            const double m = ldf(&sl->part[c][3 * sidx]), vv = ldf(&sl->part[c][3 * sidx + 1]),
                                         ix = ldf(&sl->part[c][3 * sidx + 2]);
                            if (m > best || (m == best && m > 0.0 && ix < bidx)) {
                                best = m;
                                bval = vv;
                                bidx = ix;
                            }
                        }
                        const bool none = !(best > 0.0);
                        double y = __builtin_amdgcn_rsq(var[sidx]);
                        y = y * fma(-0.5 * var[sidx] * y, y, 1.5);
                        y = y * fma(-0.5 * var[sidx] * y, y, 1.5);
                        const int ix = none ? 0 : (int)bidx;
                        double mv = (none ? ldf(&sl->part[0][6 + sidx]) : bval) * y;
                        int lag = ix > n / 2 ? ix - n : ix;
                        if (zero[sidx]) { mv = 0.0; lag = 0; }              // xcorr.go:166-167
                        if (nan[sidx]) { mv = __builtin_nan(""); lag = 0; } // placeholder: the pair is redone
                        p.mv[rA + sidx] = mv;
                        p.lag[rA + sidx] = lag;
                    }
                    // a NaN / Inf series poisons its partner through the shared transform, and sigmas too far apart cost the
                    // smaller series its precision: such pairs are redone by the kernel that isolates and rescales first
                    if (nan[0] || (hasB && (nan[1] || sigma_spread_too_wide(var[0], var[1])))) {
                        const int slot = atomicAdd(p.ovf_count, 1);
                        p.ovf_list[slot] = pair;
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    add(&sl->freed, 1u); // (everything of this sequence has been read: the slot may take sequence seq + NS)
                }
            }
        }
    }
}

// n = 32768, 65536, float64 rows, every pair.  p.team_ctl: a zeroed team::Ctl; p.team_slots slices per XCD in p.gscratch;
// wgs_per_cu resident workgroups per CU draw the tasks.
hipError_t launch_long_team(const FusedParams &p, int num_cus, int wgs_per_cu, hipStream_t stream)
{
    if (!p.rows || !p.gscratch || !p.twl || !p.xcp || !p.g2 || !p.g3a || !p.g3b || !p.ovf_list || !p.ovf_count || p.pair_list ||
        !p.team_ctl || p.team_dist < 1 || p.team_slots < 2 * p.team_dist + 1 || p.team_slots > team::MAX_SLOTS || (p.N < p.n && !p.c1) || wgs_per_cu < 1 || wgs_per_cu > 4 ||
        (long long)team::XCDS * p.team_slots > p.gscratch_slices)
        return hipErrorInvalidValue;
    const dim3 grid((unsigned)(num_cus * wgs_per_cu)), block(256);
    if (p.logn == 15) {
        if (p.N < p.n)
            hipLaunchKernelGGL((xcorr_long_team<15, true>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_long_team<15, false>), grid, block, 0, stream, p);
    } else if (p.logn == 16) {
        if (p.N < p.n)
            hipLaunchKernelGGL((xcorr_long_team<16, true>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((xcorr_long_team<16, false>), grid, block, 0, stream, p);
    } else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

size_t long_team_ctl_bytes() { return sizeof(team::Ctl); }
size_t long_team_error_offset() { return offsetof(team::Ctl, error); }

} // namespace muse
