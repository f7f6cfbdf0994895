// reduce_kernels.hip -- the reduction that follows the fused xcorr pass:
// per-label-group maximum (Batch.scoreSingle, muse_batch.go:74-89; Muse.Run,
// muse.go:72-88), the Results.passed filter (results.go:46-52) and a device
// side pre-selection of top-N candidates (results.go:55-72 keeps the N
// largest |score|).  HBM traffic here is 12-28 B per series: negligible next
// to the 8*N B per series of the fused pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xcorr_kernels.h"

namespace muse {

constexpr long long IDX_NONE = 0x7fffffffffffffffLL;

// muse_batch.go:74-77 (abs, clamp to 1) / muse.go:72-76 (signed clamp)
__device__ __forceinline__ double clamp_score(double mv, int abs_scores)
{
    if (abs_scores) {
        double v = fabs(mv);
        if (v > 1.0)
            v = 1.0;
        return v;
    }
    double v = mv;
    if (v > 1.0)
        v = 1.0;
    else if (v < -1.0)
        v = -1.0;
    return v;
}
__device__ __forceinline__ unsigned long long abs_bits(double v)
{
    return (unsigned long long)__double_as_longlong(fabs(v));
}

__global__ void group_init_kernel(GroupWork gw, int G)
{
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
        gw.key[g] = 0ull;
        gw.first[g] = IDX_NONE;
        gw.win[g] = IDX_NONE;
    }
}

// pass A: per group, lowest member index and max |score| bits over non-NaN members
__global__ void group_key_kernel(SelectParams sp, GroupWork gw)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < sp.M;
         i += (long long)gridDim.x * blockDim.x) {
        const int g = sp.group_id[i];
        if (g < 0 || g >= sp.G)
            continue;
        atomicMin(&gw.first[g], i);
        if (sp.include && sp.include[i] != 1)
            continue; // filter-and-refine Run: this row cannot be the group's record (its score is an estimate)
        const double v = clamp_score(sp.mv[i], sp.abs_scores);
        if (v == v)
            atomicMax(&gw.key[g], abs_bits(v));
    }
}
// pass B: lowest index attaining the group max ("first wins ties", muse_batch.go:87)
__global__ void group_win_kernel(SelectParams sp, GroupWork gw)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < sp.M;
         i += (long long)gridDim.x * blockDim.x) {
        const int g = sp.group_id[i];
        if (g < 0 || g >= sp.G)
            continue;
        if (sp.include && sp.include[i] != 1)
            continue;
        const double v = clamp_score(sp.mv[i], sp.abs_scores);
        if (v == v && abs_bits(v) == gw.key[g])
            atomicMin(&gw.win[g], i);
    }
}

// results.go:46-52
__device__ __forceinline__ bool passed(double s, int lag, const SelectParams &sp)
{
    return fabs((double)lag) <= (double)sp.max_lag && fabs(s) >= sp.threshold &&
           (sp.sign_filter == 0 || (s > 0 && sp.sign_filter == 1) || (s < 0 && sp.sign_filter == -1));
}

// one group's record + selection key from its first member f and its winner w (IDX_NONE: none)
__device__ __forceinline__ void final_record(const SelectParams &sp, int g, long long f, long long w, muse_record &r,
                                             unsigned long long &key)
{
    r.series = -1;
    r.score = 0.0;
    r.lag = 0;
    r.group = g;
    key = 0ull;
    if (sp.group_id) {
        if (f == IDX_NONE) // empty group: Score.Labels == nil, results.go:56-59
            return;
        const double vf = clamp_score(sp.mv[f], sp.abs_scores);
        if (sp.partial) { // one shard of the group: the merge over shards decides (muse_merge_group_records)
            key = vf != vf ? 2ull : 1ull;
            if (w != IDX_NONE) {
                r.series = w + sp.series_offset;
                r.score = clamp_score(sp.mv[w], sp.abs_scores);
                r.lag = sp.lag[w];
            }
            return;
        }
        if (w == IDX_NONE || vf != vf) // first member NaN is never replaced (x > NaN is false)
            w = f;
    } else {
        w = g;
    }
    const double s = clamp_score(sp.mv[w], sp.abs_scores);
    const int lg = sp.lag[w];
    r.series = w + sp.series_offset;
    r.score = s;
    r.lag = lg;
    // (filter-and-refine Run: only rows the fp64 kernel has re-evaluated may be selected)
    key = (!sp.include || sp.include[w] == 1) && passed(s, lg, sp) ? abs_bits(s) + 1ull : 0ull;
}

// pass C: one record + selection key per group
__global__ void group_final_kernel(SelectParams sp, GroupWork gw, muse_record *rec, unsigned long long *selkey)
{
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < sp.G; g += gridDim.x * blockDim.x) {
        muse_record r;
        unsigned long long key;
        final_record(sp, g, sp.group_id ? gw.first[g] : IDX_NONE, sp.group_id ? gw.win[g] : IDX_NONE, r, key);
        rec[g] = r;
        selkey[g] = key;
    }
}

// one 32-byte slot of pinned host memory: the record in ordinary stores (neighbouring lanes' slots leave the chip combined into a few
// large writes), a wait until the memory system has taken them, then the stamp into the same 32-byte block -- slots are 32-byte
// aligned, so a slot never straddles a 64-byte line, and writes to one line stay in order.  Measured alternatives
// (tools/time_small_runs.py, 10 000-series Run(nil) / 5 000 x 480 in 100 groups): __threadfence_system() between record and stamp
// 438 / 40 us (its L2 write-back costs ~2 us per wave and serialises); system-scope (sc0 sc1) stores 1 733 / 68 us (every 8-byte
// store its own PCIe write, ~43 ns each).
__device__ __forceinline__ void write_slot(SmallSlot *slot, const muse_record &r, unsigned long long k, unsigned long long token)
{
    slot->series = r.series;
    slot->score = r.score;
    slot->lag = r.lag;
    slot->key = k > 2ull ? 3u : (unsigned)k;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    slot->stamp = token; // (an ordinary store too: a volatile one is sc0 sc1)
    asm volatile("" ::: "memory");
}

// Small Runs (the reference's own benchmark shapes: 5 000 series in 100 label groups, muse_batch_test.go:134-162): the four passes
// above in ONE workgroup with the per-group work arrays in LDS, each group's record written straight into a 32-byte slot of pinned
// host memory -- one launch and a poll where the general path has four launches, two copies and a synchronisation.  Every slot
// carries its own stamp, stored by the thread that wrote the slot, behind the slot's own stores and into the same 32-byte block:
// writes to DIFFERENT blocks of host memory arrive in no particular order (one flag behind all records was seen to overtake them:
// tools/soak_round6.py), so the host waits for every slot's stamp.
__global__ __launch_bounds__(1024) void small_groups_kernel(SelectParams sp, SmallSlot *out, unsigned long long token)
{
    extern __shared__ unsigned long long sm[];
    unsigned long long *key = sm;
    long long *first = (long long *)(sm + sp.G), *win = (long long *)(sm + 2 * (size_t)sp.G);
    const int t = threadIdx.x;
    if (sp.group_id) {
        for (int g = t; g < sp.G; g += 1024) {
            key[g] = 0ull;
            first[g] = IDX_NONE;
            win[g] = IDX_NONE;
        }
        __syncthreads();
        // (one workgroup: what it waits for is the latency of its own loads -- eight series per thread are requested before the
        // first of them is used: 13.7 -> 9.3 us for 10 000 series in 100 groups, rocprofv3 kernel trace)
        constexpr int U = 8;
        for (long long i0 = t; i0 < sp.M; i0 += U * 1024) { // group_key_kernel
            int g[U];
            double v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long long i = i0 + u * 1024;
                g[u] = -1;
                v[u] = 0.0;
                if (i < sp.M) {
                    g[u] = sp.group_id[i];
                    v[u] = sp.mv[i];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long long i = i0 + u * 1024;
                if (g[u] < 0 || g[u] >= sp.G)
                    continue;
                atomicMin(&first[g[u]], i);
                if (sp.include && sp.include[i] != 1)
                    continue;
                const double c = clamp_score(v[u], sp.abs_scores);
                if (c == c)
                    atomicMax(&key[g[u]], abs_bits(c));
            }
        }
        __syncthreads();
        for (long long i0 = t; i0 < sp.M; i0 += U * 1024) { // group_win_kernel
            int g[U];
            double v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long long i = i0 + u * 1024;
                g[u] = -1;
                v[u] = 0.0;
                if (i < sp.M) {
                    g[u] = sp.group_id[i];
                    v[u] = sp.mv[i];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long long i = i0 + u * 1024;
                if (g[u] < 0 || g[u] >= sp.G)
                    continue;
                if (sp.include && sp.include[i] != 1)
                    continue;
                const double c = clamp_score(v[u], sp.abs_scores);
                if (c == c && abs_bits(c) == key[g[u]])
                    atomicMin(&win[g[u]], i);
            }
        }
        __syncthreads();
    }
    for (int g = t; g < sp.G; g += 1024) {
        muse_record r;
        unsigned long long k;
        final_record(sp, g, sp.group_id ? first[g] : IDX_NONE, sp.group_id ? win[g] : IDX_NONE, r, k);
        write_slot(out + g, r, k, token);
    }
}

// the same without a label map (Run(nil): every series its own group): no reduction at all, one thread per series on as many
// workgroups as it takes, each writing its slot and then the slot's stamp
__global__ __launch_bounds__(256) void ungrouped_slots_kernel(SelectParams sp, SmallSlot *out, unsigned long long token)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= sp.G)
        return;
    muse_record r;
    unsigned long long k;
    final_record(sp, g, IDX_NONE, IDX_NONE, r, k);
    write_slot(out + g, r, k, token);
}

hipError_t launch_small_groups(const SelectParams &sp, SmallSlot *out, unsigned long long token, hipStream_t stream)
{
    if (sp.G <= 0 || sp.G > (sp.group_id ? SMALL_GROUPS_MAX_G : SMALL_UNGROUPED_MAX))
        return hipErrorInvalidValue;
    if (sp.group_id)
        hipLaunchKernelGGL(small_groups_kernel, dim3(1), dim3(1024), (size_t)sp.G * 3 * sizeof(unsigned long long), stream, sp, out, token);
    else
        hipLaunchKernelGGL(ungrouped_slots_kernel, dim3((sp.G + 255) / 256), dim3(256), 0, stream, sp, out, token);
    return hipGetLastError();
}

hipError_t launch_group_reduce(const SelectParams &sp, const GroupWork &gw, muse_record *rec,
                               unsigned long long *selkey, hipStream_t stream)
{
    if (sp.G <= 0)
        return hipSuccess;
    const int gb = (int)((sp.G + 255) / 256 < 2048 ? (sp.G + 255) / 256 : 2048);
    if (sp.group_id) {
        long long mb = (sp.M + 255) / 256;
        if (mb > 4096)
            mb = 4096;
        if (mb < 1)
            mb = 1;
        hipLaunchKernelGGL(group_init_kernel, dim3(gb), dim3(256), 0, stream, gw, sp.G);
        hipLaunchKernelGGL(group_key_kernel, dim3((unsigned)mb), dim3(256), 0, stream, sp, gw);
        hipLaunchKernelGGL(group_win_kernel, dim3((unsigned)mb), dim3(256), 0, stream, sp, gw);
    }
    hipLaunchKernelGGL(group_final_kernel, dim3(gb), dim3(256), 0, stream, sp, gw, rec, selkey);
    return hipGetLastError();
}

// ---------------------------------------------------------------- top-N
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// Each workgroup owns TOPN_CHUNK consecutive groups and extracts its K best
// by (selkey desc, group id asc), one per round.
__global__ __launch_bounds__(256) void topn_kernel(const muse_record *__restrict__ rec,
                                                   const unsigned long long *__restrict__ selkey, int G, int K,
                                                   muse_record *cand, int *cnt)
{
    __shared__ unsigned long long sk[4];
    __shared__ int si[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int base = blockIdx.x * TOPN_CHUNK;
    unsigned long long k[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int g = base + t + 256 * m;
        k[m] = g < G ? selkey[g] : 0ull;
    }
    int r = 0;
    for (; r < K; r++) {
        unsigned long long bk = 0ull;
        int bm = 0;
#pragma unroll
        for (int m = 0; m < 16; m++)
            if (k[m] > bk) {
                bk = k[m];
                bm = m;
            }
        const unsigned long long wk = wave_max_u64(bk);
        if (lane == 0)
            sk[wave] = wk;
        __syncthreads();
        unsigned long long BK = sk[0];
        BK = sk[1] > BK ? sk[1] : BK;
        BK = sk[2] > BK ? sk[2] : BK;
        BK = sk[3] > BK ? sk[3] : BK;
        if (BK == 0ull)
            break; // uniform
        int ci = (bk == BK) ? (t + 256 * bm) : 0x7fffffff;
        ci = wave_min_i32(ci);
        if (lane == 0)
            si[wave] = ci;
        __syncthreads();
        const int CI = min(min(si[0], si[1]), min(si[2], si[3]));
        if (t == (CI & 255)) {
            const int mm = CI >> 8;
#pragma unroll
            for (int m = 0; m < 16; m++)
                if (m == mm)
                    k[m] = 0ull;
            cand[(long long)blockIdx.x * K + r] = rec[base + CI];
        }
        // sk/si are rewritten only after the next round's first barrier pair
        __syncthreads();
    }
    if (t == 0)
        cnt[blockIdx.x] = r;
}


// Run(nil) over more series than the exact feed takes: topn_kernel's selection with the keys computed here (what
// group_final_kernel writes for an ungrouped Run: the series' own clamped score through passed()) and the picked series' records
// written straight into pinned slots -- the M records and keys of the general path are never materialised.
__global__ __launch_bounds__(256) void topn_ungrouped_kernel(SelectParams sp, int K, SmallSlot *cand, CountSlot *cnt,
                                                             unsigned long long token)
{
    __shared__ unsigned long long sk[4];
    __shared__ int si[4];
    __shared__ int picked[TOPN_DEVICE_MAX];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long base = (long long)blockIdx.x * TOPN_CHUNK;
    unsigned long long k[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const long long g = base + t + 256 * m;
        k[m] = 0ull;
        if (g < sp.G) {
            const double s = clamp_score(sp.mv[g], sp.abs_scores);
            if (passed(s, sp.lag[g], sp))
                k[m] = abs_bits(s) + 1ull;
        }
    }
    int r = 0;
    for (; r < K; r++) {
        unsigned long long bk = 0ull;
        int bm = 0;
#pragma unroll
        for (int m = 0; m < 16; m++)
            if (k[m] > bk) {
                bk = k[m];
                bm = m;
            }
        const unsigned long long wk = wave_max_u64(bk);
        if (lane == 0)
            sk[wave] = wk;
        __syncthreads();
        unsigned long long BK = sk[0];
        BK = sk[1] > BK ? sk[1] : BK;
        BK = sk[2] > BK ? sk[2] : BK;
        BK = sk[3] > BK ? sk[3] : BK;
        if (BK == 0ull)
            break; // uniform
        int ci = (bk == BK) ? (t + 256 * bm) : 0x7fffffff;
        ci = wave_min_i32(ci);
        if (lane == 0)
            si[wave] = ci;
        __syncthreads();
        const int CI = min(min(si[0], si[1]), min(si[2], si[3]));
        if (t == (CI & 255)) {
            const int mm = CI >> 8;
#pragma unroll
            for (int m = 0; m < 16; m++)
                if (m == mm)
                    k[m] = 0ull;
            picked[r] = CI;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = t; i < r; i += 256) {
        const long long g = base + picked[i];
        muse_record rec;
        rec.series = g + sp.series_offset;
        rec.score = clamp_score(sp.mv[g], sp.abs_scores);
        rec.lag = sp.lag[g];
        rec.group = 0;
        write_slot(cand + (long long)blockIdx.x * K + i, rec, 1ull, token);
    }
    if (t == 0) {
        cnt[blockIdx.x].count = (unsigned long long)r;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cnt[blockIdx.x].stamp = token;
        asm volatile("" ::: "memory");
    }
}

hipError_t launch_topn_ungrouped(const SelectParams &sp, int K, SmallSlot *cand, CountSlot *cnt, unsigned long long token,
                                 hipStream_t stream)
{
    if (sp.G <= 0 || sp.group_id || sp.include || K < 1 || K > TOPN_DEVICE_MAX)
        return hipErrorInvalidValue;
    const int nb = (sp.G + TOPN_CHUNK - 1) / TOPN_CHUNK;
    hipLaunchKernelGGL(topn_ungrouped_kernel, dim3(nb), dim3(256), 0, stream, sp, K, cand, cnt, token);
    return hipGetLastError();
}

// ------------------------------------------------- filter-and-refine Run (ungrouped, n = 4096)
// Selection keys of one row from the screening pass's estimate `s` (|s - exact| <= E at every possible argmax) and
// its SCR_* flags: kmin <= exact key <= kplus, where the exact key is what group_final_kernel computes from the
// fp64 result (abs_bits(clamped |score|) + 1 if passed() else 0).
// what the screening pass knows about one row
struct RowBounds {
    bool nan;                 // the exact score is NaN
    double lo, hi;            // clamped |score| bounds: lo <= exact <= hi
    bool pass_may, pass_must; // Results.passed apart from the Threshold test: may be true / is certainly true
};
__device__ __forceinline__ RowBounds row_bounds(double sv, double var, unsigned f, const ScreenSelect &q)
{
    RowBounds r;
    r.nan = (f & SCR_NAN) != 0u;
    if (r.nan || (f & SCR_REFINE)) { // no estimate: may be anything (NaN rows are never selected)
        r.lo = 0.0;
        r.hi = 1.0;
        r.pass_may = !r.nan;
        r.pass_must = false;
        return r;
    }
    const double s = var > 0.0 ? sv * (1.0 / sqrt(var)) : 0.0;
    const double a = fabs(s);
    double lo = a - q.E, hi = a + q.E;
    r.lo = lo < 0.0 ? 0.0 : (lo > 1.0 ? 1.0 : lo);
    r.hi = hi > 1.0 ? 1.0 : hi;
    const bool lag_may = (f & SCR_IN) != 0u, lag_must = lag_may && !(f & SCR_OUT);
    bool sign_may = true, sign_must = true;
    if (q.sign_filter != 0 && q.abs_scores) { // Batch.Run filters the sign of |score| (muse_batch.go:74-77): > 0 unless the score is 0
        sign_may = q.sign_filter > 0;
        sign_must = q.sign_filter > 0 && r.lo > 0.0;
    } else if (q.sign_filter != 0) {
        const unsigned want = q.sign_filter > 0 ? SCR_POS : SCR_NEG, other = q.sign_filter > 0 ? SCR_NEG : SCR_POS;
        const bool small = a <= 4.0 * q.E; // the exact value at the exact argmax may be zero or of either sign
        sign_may = (f & want) != 0u || small;
        sign_must = (f & want) != 0u && !(f & other) && !small;
    }
    r.pass_may = lag_may && sign_may;
    r.pass_must = lag_must && sign_must;
    return r;
}
__device__ __forceinline__ void screen_keys(double sv, double var, unsigned f, const ScreenSelect &q,
                                            unsigned long long &kmin, unsigned long long &kplus)
{
    const RowBounds r = row_bounds(sv, var, f, q);
    const bool may = r.pass_may && r.hi >= q.threshold;
    const bool must = r.pass_must && r.lo >= q.threshold && r.lo > 0.0;
    kplus = may ? abs_bits(r.hi) + 1ull : 0ull;
    kmin = must ? abs_bits(r.lo) + 1ull : 0ull;
}

// pessimistic key of every row -> selkey
__global__ void screen_kmin_kernel(ScreenSelect q, unsigned long long *selkey)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < q.M; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long kmin, kplus;
        screen_keys(q.mv[i], q.var[i], q.flags[i], q, kmin, kplus);
        selkey[i] = kmin;
    }
}

// per chunk of TOPN_CHUNK rows: its K largest keys, descending (zero-filled)
__global__ __launch_bounds__(256) void topn_keys_kernel(const unsigned long long *__restrict__ selkey, long long G, int K,
                                                        unsigned long long *keys)
{
    __shared__ unsigned long long sk[4];
    __shared__ int si[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long base = (long long)blockIdx.x * TOPN_CHUNK;
    unsigned long long k[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const long long g = base + t + 256 * m;
        k[m] = g < G ? selkey[g] : 0ull;
    }
    for (int r = 0; r < K; r++) {
        unsigned long long bk = 0ull;
        int bm = 0;
#pragma unroll
        for (int m = 0; m < 16; m++)
            if (k[m] > bk) {
                bk = k[m];
                bm = m;
            }
        const unsigned long long wk = wave_max_u64(bk);
        if (lane == 0)
            sk[wave] = wk;
        __syncthreads();
        unsigned long long BK = sk[0];
        BK = sk[1] > BK ? sk[1] : BK;
        BK = sk[2] > BK ? sk[2] : BK;
        BK = sk[3] > BK ? sk[3] : BK;
        int ci = (bk == BK && BK != 0ull) ? (t + 256 * bm) : 0x7fffffff;
        ci = wave_min_i32(ci);
        if (lane == 0)
            si[wave] = ci;
        __syncthreads();
        const int CI = min(min(si[0], si[1]), min(si[2], si[3]));
        if (CI != 0x7fffffff && t == (CI & 255)) {
            const int mm = CI >> 8;
#pragma unroll
            for (int m = 0; m < 16; m++)
                if (m == mm)
                    k[m] = 0ull;
        }
        if (t == 0)
            keys[(long long)blockIdx.x * K + r] = BK;
        __syncthreads();
    }
}

// one thread per pair of rows: the pair is re-evaluated (listed for the fp64 kernel, both rows marked in `include`)
// when either row's optimistic key is non-zero and reaches the cut
__global__ void screen_compact_kernel(ScreenSelect q, const unsigned long long *cut, long long npairs,
                                      long long *pair_list, int *pair_count, unsigned char *include)
{
    const unsigned long long c = *cut;
    for (long long pr = blockIdx.x * (long long)blockDim.x + threadIdx.x; pr < npairs; pr += (long long)gridDim.x * blockDim.x) {
        const long long rA = 2 * pr, rB = rA + 1;
        unsigned long long kmin, kp;
        screen_keys(q.mv[rA], q.var[rA], q.flags[rA], q, kmin, kp);
        bool need = kp != 0ull && kp >= c;
        if (rB < q.M) {
            screen_keys(q.mv[rB], q.var[rB], q.flags[rB], q, kmin, kp);
            need = need || (kp != 0ull && kp >= c);
        }
        if (need) {
            const int slot = atomicAdd(pair_count, 1);
            pair_list[slot] = pr;
            include[rA] = 1;
            if (rB < q.M)
                include[rB] = 1;
        }
    }
}

// ---- label groups (Batch.Run(groupByLabels)): the record of a group is its member with the largest clamped |score|
// (first index among equals; a NaN first member is never replaced), and the filters apply to that record.
//   G1: first[g] = lowest member index, glo[g] = max over members of lo (the group's score is at least that)
//   G2: members with hi >= glo[g] are the only possible winners: the group certainly passes the filters iff all of them
//       certainly do (gcert), and may pass with a score up to the largest hi among those that may (gmay)
//   G3: pessimistic / optimistic key per group; the cut is the top_n-th largest pessimistic key
//   G4: the possible winners of every group whose optimistic key reaches the cut are re-evaluated
__global__ void screen_g1_kernel(ScreenSelect q, long long *first, unsigned long long *glo)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < q.M; i += (long long)gridDim.x * blockDim.x) {
        const int g = q.group_id[i];
        if (g < 0 || g >= q.G)
            continue;
        atomicMin(&first[g], i);
        const RowBounds r = row_bounds(q.mv[i], q.var[i], q.flags[i], q);
        if (!r.nan)
            atomicMax(&glo[g], abs_bits(r.lo));
    }
}
__global__ void screen_g2_kernel(ScreenSelect q, const unsigned long long *glo, int *gcert, unsigned long long *gmay)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < q.M; i += (long long)gridDim.x * blockDim.x) {
        const int g = q.group_id[i];
        if (g < 0 || g >= q.G)
            continue;
        const RowBounds r = row_bounds(q.mv[i], q.var[i], q.flags[i], q);
        if (r.nan || abs_bits(r.hi) < glo[g])
            continue; // cannot be the group's record
        if (!r.pass_must)
            gcert[g] = 0; // (benign race: every writer stores 0)
        if (r.pass_may && r.hi >= q.threshold)
            atomicMax(&gmay[g], abs_bits(r.hi) + 1ull);
    }
}
__global__ void screen_g3_kernel(ScreenSelect q, const long long *first, const unsigned long long *glo, const int *gcert,
                                 const unsigned long long *gmay, unsigned long long *gkmin, unsigned long long *gkplus)
{
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < q.G; g += gridDim.x * blockDim.x) {
        const long long f = first[g];
        unsigned long long kmin = 0ull, kplus = 0ull;
        if (f != IDX_NONE && !(q.flags[f] & SCR_NAN)) { // (empty group / NaN first member: never selected)
            const double lo = __longlong_as_double((long long)glo[g]);
            kplus = gmay[g];
            if (gcert[g] && lo >= q.threshold && lo > 0.0)
                kmin = glo[g] + 1ull;
        }
        gkmin[g] = kmin;
        gkplus[g] = kplus;
    }
}
__global__ void screen_g4_kernel(ScreenSelect q, const unsigned long long *glo, const unsigned long long *gkplus,
                                 const unsigned long long *cut, long long npairs, long long *pair_list, int *pair_count,
                                 unsigned char *include)
{
    const unsigned long long c = *cut;
    for (long long pr = blockIdx.x * (long long)blockDim.x + threadIdx.x; pr < npairs; pr += (long long)gridDim.x * blockDim.x) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const long long i = 2 * pr + k;
            if (i >= q.M)
                continue;
            const int g = q.group_id[i];
            if (g < 0 || g >= q.G)
                continue;
            const unsigned long long kp = gkplus[g];
            if (kp == 0ull || kp < c)
                continue;
            const RowBounds r = row_bounds(q.mv[i], q.var[i], q.flags[i], q);
            if (r.nan || abs_bits(r.hi) < glo[g])
                continue;
            include[i] = 1;
            any = true;
        }
        if (any) {
            const int slot = atomicAdd(pair_count, 1);
            pair_list[slot] = pr;
        }
    }
}
__global__ void screen_ginit_kernel(int G, long long *first, unsigned long long *glo, int *gcert, unsigned long long *gmay)
{
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
        first[g] = IDX_NONE;
        glo[g] = 0ull;
        gcert[g] = 1;
        gmay[g] = 0ull;
    }
}

// ---- run-time guard of the error bound: the rows that are re-evaluated have both an fp32 estimate and (afterwards) an
// fp64 score; their largest | |estimate| - |score| | must stay below E, or the Run falls back to the all-fp64 path
__global__ void screen_save_kernel(ScreenSelect q, const long long *pair_list, const int *pair_count, double *est_save)
{
    const int n = *pair_count;
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        const long long pr = pair_list[slot];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const long long i = 2 * pr + k;
            double e = -1.0; // no estimate
            if (i < q.M) {
                const unsigned f = q.flags[i];
                const double var = q.var[i];
                if (!(f & (SCR_NAN | SCR_REFINE)) && var > 0.0)
                    e = fabs(q.mv[i] * (1.0 / sqrt(var)));
            }
            est_save[2 * (long long)slot + k] = e;
        }
    }
}
__global__ void screen_check_kernel(const double *mv, long long M, const long long *pair_list, const int *pair_count,
                                    const double *est_save, unsigned long long *err_bits)
{
    const int n = *pair_count;
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        const long long pr = pair_list[slot];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const long long i = 2 * pr + k;
            const double e = est_save[2 * (long long)slot + k];
            if (i >= M || e < 0.0)
                continue;
            const double x = fabs(mv[i]);
            if (x == x) // (a NaN score has no estimate to compare with)
                atomicMax(err_bits, abs_bits(fabs(e - x)));
        }
    }
}
// Guard sample: one pair in 1024, chosen by a hash of (pair, salt), is appended to the re-evaluation list although the
// selection does not need it, so that the check below also sees rows the bound alone vouches for (a violated bound on
// an unrefined row would otherwise be invisible).  The sampled rows get exact scores but stay OUT of the selection
// (include = 2, unless the selection listed them itself): a label group's record is its member with the largest score and
// the filters apply to that record, so a sampled member of a group whose possible winners were not re-evaluated must
// not stand in for them.
__global__ void screen_sample_kernel(long long npairs, long long M, unsigned long long salt, long long *pair_list,
                                     int *pair_count, unsigned char *include)
{
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npairs; p += (long long)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)p * 0x9E3779B97F4A7C15ull + salt;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        if ((x & 1023ull) != 0ull)
            continue;
        const int slot = atomicAdd(pair_count, 1);
        pair_list[slot] = p;
        if (include[2 * p] == 0)
            include[2 * p] = 2;
        if (2 * p + 1 < M && include[2 * p + 1] == 0)
            include[2 * p + 1] = 2;
    }
}
hipError_t launch_screen_sample(long long npairs, long long M, unsigned long long salt, long long *pair_list, int *pair_count,
                                unsigned char *include, hipStream_t stream)
{
    if (npairs <= 0)
        return hipSuccess;
    const long long blocks = (npairs + 255) / 256;
    hipLaunchKernelGGL(screen_sample_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, npairs, M,
                       salt, pair_list, pair_count, include);
    return hipGetLastError();
}

hipError_t launch_screen_save(const ScreenSelect &q, const long long *pair_list, const int *pair_count, double *est_save,
                              hipStream_t stream)
{
    hipLaunchKernelGGL(screen_save_kernel, dim3(256), dim3(256), 0, stream, q, pair_list, pair_count, est_save);
    return hipGetLastError();
}
hipError_t launch_screen_check(const double *mv, long long M, const long long *pair_list, const int *pair_count,
                               const double *est_save, unsigned long long *err_bits, hipStream_t stream)
{
    hipLaunchKernelGGL(screen_check_kernel, dim3(256), dim3(256), 0, stream, mv, M, pair_list, pair_count, est_save, err_bits);
    return hipGetLastError();
}

long long screen_select_scratch(long long G, int top_n) // G = number of selection units (rows, or label groups)
{
    const long long nb = (G + TOPN_CHUNK - 1) / TOPN_CHUNK;
    return nb * top_n + ((nb * top_n + TOPN_CHUNK - 1) / TOPN_CHUNK + 1) * top_n;
}

hipError_t launch_screen_select(const ScreenSelect &q, int top_n, unsigned long long *selkey, unsigned long long *keys,
                                const ScreenGroupWork &gw, long long *pair_list, int *pair_count, unsigned char *include,
                                hipStream_t stream)
{
    const long long units = q.group_id ? (long long)q.G : q.M;
    const long long nb = (units + TOPN_CHUNK - 1) / TOPN_CHUNK;
    long long mb = (q.M + 255) / 256;
    mb = mb > 4096 ? 4096 : (mb < 1 ? 1 : mb);
    const int gb = (int)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048);
    if (q.group_id) {
        hipLaunchKernelGGL(screen_ginit_kernel, dim3(gb), dim3(256), 0, stream, q.G, gw.first, gw.glo, gw.gcert, gw.gmay);
        hipLaunchKernelGGL(screen_g1_kernel, dim3((unsigned)mb), dim3(256), 0, stream, q, gw.first, gw.glo);
        hipLaunchKernelGGL(screen_g2_kernel, dim3((unsigned)mb), dim3(256), 0, stream, q, gw.glo, gw.gcert, gw.gmay);
        hipLaunchKernelGGL(screen_g3_kernel, dim3(gb), dim3(256), 0, stream, q, gw.first, gw.glo, gw.gcert, gw.gmay, selkey,
                           gw.gkplus);
    } else {
        hipLaunchKernelGGL(screen_kmin_kernel, dim3((unsigned)mb), dim3(256), 0, stream, q, selkey);
    }
    // the top_n largest pessimistic keys: per-chunk top-K lists, reduced again until one chunk holds them in
    // descending order; its entry top_n - 1 is the cut (0 when fewer than top_n rows certainly pass)
    unsigned long long *src = keys, *dst = keys + nb * top_n;
    hipLaunchKernelGGL(topn_keys_kernel, dim3((unsigned)nb), dim3(256), 0, stream, selkey, units, top_n, src);
    long long count = nb * top_n;
    for (;;) {
        const long long cb = (count + TOPN_CHUNK - 1) / TOPN_CHUNK;
        hipLaunchKernelGGL(topn_keys_kernel, dim3((unsigned)cb), dim3(256), 0, stream, src, count, top_n, dst);
        count = cb * top_n;
        unsigned long long *tmp = src;
        src = dst;
        dst = tmp;
        if (cb == 1)
            break;
    }
    const unsigned long long *cut = src + (top_n - 1);
    const long long npairs = (q.M + 1) / 2;
    long long pb = (npairs + 255) / 256;
    pb = pb > 4096 ? 4096 : (pb < 1 ? 1 : pb);
    if (q.group_id)
        hipLaunchKernelGGL(screen_g4_kernel, dim3((unsigned)pb), dim3(256), 0, stream, q, gw.glo, gw.gkplus, cut, npairs,
                           pair_list, pair_count, include);
    else
        hipLaunchKernelGGL(screen_compact_kernel, dim3((unsigned)pb), dim3(256), 0, stream, q, cut, npairs, pair_list,
                           pair_count, include);
    return hipGetLastError();
}

// ---------------------------------------------------------------- one label group in one workgroup (Muse.Run, muse.go:52-90)
// The whole of a small group's post-processing in ONE launch: the winner among the members whose score is a number
// (maximum of |clamped score|, the lowest index on ties -- "strictly greater replaces", muse.go:86) and the group's state
// (1: the first member's score is a number, 2: it is NaN and never replaced; 0: no member), with the meaning of
// SelectParams::partial.  `out` may be pinned host memory: the record is written once, by one lane.
__global__ __launch_bounds__(256) void single_group_kernel(const double *__restrict__ mv, const int *__restrict__ lag,
                                                           long long M, int abs_scores, long long series_offset,
                                                           SingleGroupOut *out)
{
    __shared__ unsigned long long s_key[4];
    __shared__ long long s_win[4];
    const int t = threadIdx.x, w = t >> 6;
    unsigned long long key = 0ull;
    long long win = IDX_NONE;
    for (long long i = t; i < M; i += 256) {      // ascending i per lane: the first index attaining the lane's maximum is kept
        const double v = clamp_score(mv[i], abs_scores);
        if (v == v) {
            const unsigned long long k = abs_bits(v) + 1ull;   // (+1: a score of exactly 0 still beats "no member")
            if (k > key) {
                key = k;
                win = i;
            }
        }
    }
    const unsigned long long wk = wave_max_u64(key);
    long long cand = key == wk && key != 0ull ? win : IDX_NONE;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long u = __shfl_xor(cand, o, 64);
        cand = u < cand ? u : cand;
    }
    if ((t & 63) == 0) {
        s_key[w] = wk;
        s_win[w] = cand;
    }
    __syncthreads();
    if (t == 0) {
        unsigned long long bk = 0ull;
        long long bw = IDX_NONE;
        for (int q = 0; q < 4; q++)
            if (s_key[q] > bk || (s_key[q] == bk && bk != 0ull && s_win[q] < bw)) {
                bk = s_key[q];
                bw = s_win[q];
            }
        muse_record r;
        r.series = -1;
        r.score = 0.0;
        r.lag = 0;
        r.group = 0;
        unsigned long long state = 0ull;
        if (M > 0) {
            const double vf = clamp_score(mv[0], abs_scores);
            state = vf != vf ? 2ull : 1ull;
            if (bw != IDX_NONE) {
                r.series = bw + series_offset;
                r.score = clamp_score(mv[bw], abs_scores);
                r.lag = lag[bw];
            }
        }
        out->rec = r;
        __threadfence_system();
        out->state = state;
    }
}

hipError_t launch_single_group(const double *mv, const int *lag, long long M, int abs_scores, long long series_offset,
                               SingleGroupOut *out, hipStream_t stream)
{
    hipLaunchKernelGGL(single_group_kernel, dim3(1), dim3(256), 0, stream, mv, lag, M, abs_scores, series_offset, out);
    return hipGetLastError();
}

hipError_t launch_topn(const muse_record *rec, const unsigned long long *selkey, int G, int K, muse_record *cand,
                       int *cnt, hipStream_t stream)
{
    if (G <= 0 || K <= 0)
        return hipSuccess;
    const int nb = (G + TOPN_CHUNK - 1) / TOPN_CHUNK;
    hipLaunchKernelGGL(topn_kernel, dim3(nb), dim3(256), 0, stream, rec, selkey, G, K, cand, cnt);
    return hipGetLastError();
}

} // namespace muse
