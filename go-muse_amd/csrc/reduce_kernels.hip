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

// pass C: one record + selection key per group
__global__ void group_final_kernel(SelectParams sp, GroupWork gw, muse_record *rec, unsigned long long *selkey)
{
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < sp.G; g += gridDim.x * blockDim.x) {
        long long w;
        if (sp.group_id) {
            const long long f = gw.first[g];
            if (f == IDX_NONE) { // empty group: Score.Labels == nil, results.go:56-59
                rec[g].series = -1;
                rec[g].score = 0.0;
                rec[g].lag = 0;
                rec[g].group = g;
                selkey[g] = 0ull;
                continue;
            }
            w = gw.win[g];
            const double vf = clamp_score(sp.mv[f], sp.abs_scores);
            if (w == IDX_NONE || vf != vf) // first member NaN is never replaced (x > NaN is false)
                w = f;
        } else {
            w = g;
        }
        const double s = clamp_score(sp.mv[w], sp.abs_scores);
        const int lg = sp.lag[w];
        rec[g].series = w + sp.series_offset;
        rec[g].score = s;
        rec[g].lag = lg;
        rec[g].group = g;
        selkey[g] = passed(s, lg, sp) ? abs_bits(s) + 1ull : 0ull;
    }
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
