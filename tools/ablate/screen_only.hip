// EXPERIMENT (diagnostic only, not part of libmuse_hip.so): how fast is a PURE fp32 screening pass over
// 1 M x 4096 at four workgroups per CU (128 VGPRs, 37 KB LDS)?  Approximate scores only (fp32 transform error,
// ~1e-6 absolute): a product path would re-evaluate the rows that can reach the top-N / sit near a filter bound
// with the fp64 kernel (filter-and-refine).  See docs/HISTORY.md 4.6.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I go-muse_amd/csrc tools/ablate/screen_only.hip -o tools/ablate/screen_only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../go-muse_amd/csrc/xcorr_r16_screen.hip"
using namespace muse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

namespace so {
using namespace muse::scr;
// forward fp32 FFT; pass-1 factors from four per-thread base powers W^t, W^2t, W^4t, W^8t (products of <= 4 exact factors)
// ABL (ablation, garbage results): 1 = the second pass's butterflies removed (-170 VALU instructions per transform),
// 2 = the pass-1 twiddle products removed as well (every factor = w1)
template <bool MULXC, int ABL = 0>
__device__ __forceinline__ void fft(f2 (&v)[16], f2 *xbuf, const f2 *tw2s, const f2 w1, const f2 w2, const f2 w4, const f2 w8,
                                    const f2 (&xq)[16], const int t)
{
    dft16f(v);
    if (ABL >= 2) {
#pragma unroll
        for (int k = 1; k < 16; k++)
            v[P16(k)] = cmulf(v[P16(k)], w1);
    } else
    {
        const f2 w3 = cmulf(w1, w2), w5 = cmulf(w4, w1), w6 = cmulf(w4, w2), w7 = cmulf(w4, w3);
        v[P16(1)] = cmulf(v[P16(1)], w1); v[P16(2)] = cmulf(v[P16(2)], w2); v[P16(3)] = cmulf(v[P16(3)], w3);
        v[P16(4)] = cmulf(v[P16(4)], w4); v[P16(5)] = cmulf(v[P16(5)], w5); v[P16(6)] = cmulf(v[P16(6)], w6);
        v[P16(7)] = cmulf(v[P16(7)], w7); v[P16(8)] = cmulf(v[P16(8)], w8);
        v[P16(9)] = cmulf(v[P16(9)], cmulf(w8, w1)); v[P16(10)] = cmulf(v[P16(10)], cmulf(w8, w2));
        v[P16(11)] = cmulf(v[P16(11)], cmulf(w8, w3)); v[P16(12)] = cmulf(v[P16(12)], cmulf(w8, w4));
        v[P16(13)] = cmulf(v[P16(13)], cmulf(w8, w5)); v[P16(14)] = cmulf(v[P16(14)], cmulf(w8, w6));
        v[P16(15)] = cmulf(v[P16(15)], cmulf(w8, w7));
    }
    exchange<false>(v, xbuf, t);
    if (ABL < 1)
        dft16f(v);
    {
        const int lo = t & 15;
#pragma unroll
        for (int k = 1; k < 16; k++)
            v[P16(k)] = cmulf(v[P16(k)], tw2s[k * 16 + lo]);
    }
    exchange<true>(v, xbuf, t);
    dft16f(v);
    f2 w[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        w[k] = v[P16(k)];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = MULXC ? cmulf(w[k], xq[k]) : w[k];
}
} // namespace so

template <int WPC, bool TIMING, int ABL = 0>
__global__ __launch_bounds__(256, WPC) void screen_only(const FusedParams p)
{
    using namespace muse::scr;
    __shared__ f2 xbuf[SCR_XBUF];
    __shared__ f2 tw2s[256];
    __shared__ double red[32]; // [0,16) statistics; [16,32): per wave (max A, max B) fp32 + indices
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr double invN = 1.0 / 4096.0, invNm1 = 1.0 / 4095.0;
    float *redf = reinterpret_cast<float *>(red + 16);
    int *redi = reinterpret_cast<int *>(red + 24);
    f2 w1, w2, w4, w8;
    {
        const float2 tw = p.tw2f[t];
        tw2s[t] = mk2(tw.x, tw.y);
        const gptr<float2> tp = scalar_ptr(p.tw1f);
        w1 = ldg_f2(tp, 256 + t); w2 = ldg_f2(tp, 512 + t); w4 = ldg_f2(tp, 1024 + t); w8 = ldg_f2(tp, 2048 + t);
    }
    __syncthreads();
    PhaseClock<TIMING> clk;
    clk.start();
    long long pair = blockIdx.x;
    double ra[16], rb[16], kA, kB;
    issue_series(ra, kA, p.rows + 2 * pair * p.stride, t);
    issue_series(rb, kB, p.rows + (2 * pair + 1 < p.M ? 2 * pair + 1 : 2 * pair) * p.stride, t);
    // results of the previous pair: finalised by lanes 0 / 1 of wave 0 one barrier later
    for (; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        long long nxt = pair + gridDim.x;
        nxt = nxt < p.npairs ? nxt : p.npairs - 1;
        const long long nA = 2 * nxt, nB = (nA + 1 < p.M) ? nA + 1 : nA;
        if (TIMING)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        clk.template stamp<0>();
        float na[16], nb[16];
        double q[4];
        reduce_series(ra, kA, na, q[0], q[1]);
        reduce_series(rb, kB, nb, q[2], q[3]);
        fence();
        // the batch's spectrum factors for the first transform's last pass: issued while no HBM load is in flight
        f2 xq[16];
        {
            const Tw1FetchF fetch{p.xcf, t};
#pragma unroll
            for (int k = 0; k < 16; k++)
                xq[k] = fetch(k);
        }
        fence();
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
            q[k] = uniform((red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]));
        const double mA = uniform(q[0] * invN), mB = uniform(q[2] * invN);
        const double varA = uniform((q[1] - q[0] * q[0] * invN) * invNm1);
        const double varB = uniform((q[3] - q[2] * q[2] * invN) * invNm1);
        const bool deadA = !(varA > 0.0) || !__builtin_isfinite(varA), deadB = !(varB > 0.0) || !__builtin_isfinite(varB) || !hasB;
        const int eA = (int)((__double_as_longlong(varA) >> 52) & 0x7ff) - 1023;
        const int eB = (int)((__double_as_longlong(varB) >> 52) & 0x7ff) - 1023;
        const float sclA = deadA ? 0.f : __int_as_float((127 - (eA >> 1)) << 23);
        const float sclB = deadB ? 0.f : __int_as_float((127 - (eB >> 1)) << 23);
        const float mAf = deadA ? 0.f : (float)mA, mBf = deadB ? 0.f : (float)mB;
        f2 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = mk2((na[i] - mAf) * sclA, (nb[i] - mBf) * sclB);
        clk.template stamp<1>();
        so::fft<true, ABL>(v, xbuf, tw2s, w1, w2, w4, w8, xq, t);
        clk.template stamp<2>();
        // ---- the next pair streams in behind the second transform (no other global load until it is consumed)
        fence();
        issue_series(ra, kA, p.rows + nA * p.stride, t);
        issue_series(rb, kB, p.rows + nB * p.stride, t);
        fence();
        so::fft<false, ABL>(v, xbuf, tw2s, w1, w2, w4, w8, xq, t);
        clk.template stamp<3>();
        // ---- fp32 argmax |cc| per series: first index of the maximum (lag index 256 k + t)
        float ma = -1.f, mb = -1.f;
        int ia = 0, ib = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float xa = fabsf(v[k].x), xb = fabsf(v[k].y);
            if (xa > ma) { ma = xa; ia = k; }
            if (xb > mb) { mb = xb; ib = k; }
        }
        const float sa = v[0].x; // (sign recovery omitted in this experiment)
        (void)sa;
        const float MA = wave_max_f32_dpp(ma), MB = wave_max_f32_dpp(mb);
        int ca = (ma == MA) ? 256 * ia + t : 0x7fffffff, cb = (mb == MB) ? 256 * ib + t : 0x7fffffff;
        ca = wave_min_i_dpp(ca);
        cb = wave_min_i_dpp(cb);
        if (lane == 0) {
            redf[wave] = MA; redf[4 + wave] = MB;
            redi[wave] = ca; redi[4 + wave] = cb;
        }
        lds_barrier();
        if (t < 2) { // lane 0: series A, lane 1: series B
            float M = -1.f; int L = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float x = redf[4 * t + w]; const int l = redi[4 * t + w];
                if (x > M || (x == M && l < L)) { M = x; L = l; }
            }
            const double var = t ? varB : varA;
            const int e = t ? eB : eA;
            const double u = __longlong_as_double((long long)(1023 + (e >> 1)) << 52);
            const bool dead = t ? deadB : deadA;
            const double mv = dead ? 0.0 : (double)M * u * (1.0 / sqrt(var));
            const int lag = dead ? 0 : (L > 2048 ? L - 4096 : L);
            if (t == 0 || hasB) {
                p.mv[rA + t] = mv;
                p.lag[rA + t] = lag;
            }
        }
        clk.template stamp<4>();
    }
    if (TIMING && p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < NPHASE; i++)
            p.dbg[((long long)blockIdx.x * 4 + wave) * NPHASE + i] = clk.acc[i];
    }
}

__global__ void fill(double* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5; } }
template <typename K> void run(const char* title, K kern, FusedParams p, int grid, bool stamped)
{
    const char* names[5] = {"rows wait", "reduce+stats+convert", "FFT1 (+xc)", "issue next + FFT2", "argmax+store"};
    unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)grid * 4 * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8)); p.dbg = dbg;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int loops = getenv("LOOPS") ? atoi(getenv("LOOPS")) : 5;
    for (int l = 0; l < loops; l++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double pairs_per_wg = (double)p.npairs / grid;
    printf("%s grid=%d: %.3f ms, %.1f pairs per workgroup -> %.1f%% of 8 TB/s\n", title, grid, ms, pairs_per_wg, p.M * 32784.0 / (ms * 1e-3) / 8e12 * 100);
    if (stamped) {
        std::vector<unsigned long long> h((size_t)grid * 4 * 16);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double tot = 0; double s[16] = {0};
        for (int w = 0; w < grid * 4; w++) for (int i = 0; i < 16; i++) s[i] += (double)h[(size_t)w * 16 + i];
        for (int i = 0; i < 5; i++) tot += s[i];
        for (int i = 0; i < 5; i++) printf("  %-24s %9.0f cycles/pair/wave  %5.1f%%\n", names[i], s[i] / (grid * 4) / pairs_per_wg, 100.0 * s[i] / tot);
        printf("  total %.0f cycles/pair/wave\n", tot / (grid * 4) / pairs_per_wg);
    }
    CK(hipFree(dbg));
}
int main(int argc, char** argv)
{
    long long M = argc > 1 ? atoll(argv[1]) : 1000000;
    FusedParams p{}; p.M = M; p.stride = 4096; p.npairs = M / 2; p.N = 4096; p.n = 4096; p.logn = 12; p.normalize_y = 1;
    double* rows; CK(hipMalloc(&rows, M * 4096 * 8)); p.rows = rows;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, rows, M * 4096);
    std::vector<float2> t1f(4096), t2f(256), xcf(4096);
    for (int k = 0; k < 16; k++) for (int t = 0; t < 256; t++) { double a = -2 * M_PI * ((k * t) % 4096) / 4096.0; t1f[k * 256 + t] = make_float2((float)cos(a), (float)sin(a)); }
    for (int k = 0; k < 16; k++) for (int c = 0; c < 16; c++) { double a = -2 * M_PI * ((k * c) % 256) / 256.0; t2f[k * 16 + c] = make_float2((float)cos(a), (float)sin(a)); }
    for (int i = 0; i < 4096; i++) xcf[i] = make_float2((float)(cos(0.001 * i) / 4096), (float)(sin(0.002 * i) / 4096));
    float2 *f1, *f2_, *fx; CK(hipMalloc(&f1, 4096 * 8)); CK(hipMalloc(&f2_, 256 * 8)); CK(hipMalloc(&fx, 4096 * 8));
    CK(hipMemcpy(f1, t1f.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(f2_, t2f.data(), 256 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(fx, xcf.data(), 4096 * 8, hipMemcpyHostToDevice));
    p.tw1f = f1; p.tw2f = f2_; p.xcf = fx;
    CK(hipMalloc(&p.mv, M * 8)); CK(hipMalloc(&p.lag, M * 4));
    CK(hipDeviceSynchronize());
    unsigned* fl; CK(hipMalloc(&fl, M * 4)); CK(hipMemset(fl, 0, M * 4)); p.scr_flags = fl; double* sv; CK(hipMalloc(&sv, M * 8)); p.scr_var = sv; p.scr_max_lag = 15; p.screen_delta = 1e-3;
    const int which = argc > 2 ? atoi(argv[2]) : 127; // bit mask of the configurations to run
    if (which & 1) run("screen-only WPC=3", screen_only<3, false>, p, 256 * 3, false);
    if (which & 32) run("screen-only WPC=3, ablation 1 (-340 VALU instructions per pair)", screen_only<3, false, 1>, p, 256 * 3, false);
    if (which & 64) run("screen-only WPC=3, ablation 2 (-430 VALU instructions per pair)", screen_only<3, false, 2>, p, 256 * 3, false);
    if (which & 2) run("library screening pass (MaxLag < 256, sign-agnostic Run)", xcorr_screen_pass_n4096<3, false, false, true, false>, p, 256 * 3, false);
    if (which & 4) run("library screening pass (MaxLag < 256, signed Run)", xcorr_screen_pass_n4096<3, false, false, true, true>, p, 256 * 3, false);
    if (which & 8) run("library screening pass (sign-agnostic, stamped)", xcorr_screen_pass_n4096<3, true, false, true, false>, p, 256 * 3, true);
    if (which & 16) run("library screening pass (signed, stamped)", xcorr_screen_pass_n4096<3, true, false, true, true>, p, 256 * 3, true);
    return 0;
}
