// EXPERIMENT (diagnostic only): the screening pass at FOUR workgroups per CU (128 VGPRs): the next pair's rows are
// prefetched one series at a time into a single 32-register fp64 buffer (series A behind the first transform's end,
// reduced to fp32 + sums next to the second transform's last butterflies; series B from there to the next top).
// Timing only (no flags / estimates are checked here); compare with tools/ablate/screen_only.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I go-muse_amd/csrc tools/ablate/screen_w4.hip -o tools/ablate/screen_w4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../go-muse_amd/csrc/xcorr_r16_screen.hip"
using namespace muse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

namespace w4 {
using namespace muse::scr;
template <bool MULXC, typename F, typename L>
__device__ __forceinline__ void fft(f2 (&v)[16], f2 *xbuf, const f2 *tw2s, const f2 w1, const f2 w2, const f2 w4_, const f2 w8,
                                    const f2 (&xq)[16], const int t, F mid, L late)
{
    dft16f(v);
    {
        const f2 w3 = cmulf(w1, w2), w5 = cmulf(w4_, w1), w6 = cmulf(w4_, w2), w7 = cmulf(w4_, w3);
        v[P16(1)] = cmulf(v[P16(1)], w1); v[P16(2)] = cmulf(v[P16(2)], w2); v[P16(3)] = cmulf(v[P16(3)], w3);
        v[P16(4)] = cmulf(v[P16(4)], w4_); v[P16(5)] = cmulf(v[P16(5)], w5); v[P16(6)] = cmulf(v[P16(6)], w6);
        v[P16(7)] = cmulf(v[P16(7)], w7); v[P16(8)] = cmulf(v[P16(8)], w8);
        v[P16(9)] = cmulf(v[P16(9)], cmulf(w8, w1)); v[P16(10)] = cmulf(v[P16(10)], cmulf(w8, w2));
        v[P16(11)] = cmulf(v[P16(11)], cmulf(w8, w3)); v[P16(12)] = cmulf(v[P16(12)], cmulf(w8, w4_));
        v[P16(13)] = cmulf(v[P16(13)], cmulf(w8, w5)); v[P16(14)] = cmulf(v[P16(14)], cmulf(w8, w6));
        v[P16(15)] = cmulf(v[P16(15)], cmulf(w8, w7));
    }
    exchange<false>(v, xbuf, t);
    fence();
    mid();
    fence();
    dft16f(v);
    {
        const int lo = t & 15;
#pragma unroll
        for (int k = 1; k < 16; k++)
            v[P16(k)] = cmulf(v[P16(k)], tw2s[k * 16 + lo]);
    }
    exchange<true>(v, xbuf, t);
    fence();
    late();
    fence();
    dft16f(v);
    f2 w[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        w[k] = v[P16(k)];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = MULXC ? cmulf(w[k], xq[k]) : w[k];
}
} // namespace w4

// END: 0 = bare maxima; 1 = the full end phase (four folded maxima per series, single-writer finish through LDS);
// 2 = END 1 + the trust rules / dead-series handling at the top; 3 = END 2 with series B requested only after the fold
// (the transform's values are dead by then: fewer live registers at the tightest point, a shorter window for B)
template <int WPC, int END = 0>
__global__ __launch_bounds__(256, WPC) void screen_w4(const FusedParams p)
{
    using namespace muse::scr;
    __shared__ f2 xbuf[SCR_XBUF];
    __shared__ f2 tw2s[256];
    __shared__ double red[32];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr double invN = 1.0 / 4096.0, invNm1 = 1.0 / 4095.0;
    float *redf = reinterpret_cast<float *>(red + 16);
    const float window = (float)p.screen_delta;
    const int max_lag = p.scr_max_lag;
    f2 w1, w2, w4_, w8;
    {
        const float2 tw = p.tw2f[t];
        tw2s[t] = mk2(tw.x, tw.y);
        const gptr<float2> tp = scalar_ptr(p.tw1f);
        w1 = ldg_f2(tp, 256 + t); w2 = ldg_f2(tp, 512 + t); w4_ = ldg_f2(tp, 1024 + t); w8 = ldg_f2(tp, 2048 + t);
    }
    __syncthreads();
    long long pair = blockIdx.x;
    double raw[16], k0, sA1, sA2;
    float na[16];
    issue_series(raw, k0, p.rows + 2 * pair * p.stride, t);
    fence();
    reduce_series(raw, k0, na, sA1, sA2);
    fence();
    issue_series(raw, k0, p.rows + (2 * pair + 1 < p.M ? 2 * pair + 1 : 2 * pair) * p.stride, t);
    fence();
    for (; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        long long nxt = pair + gridDim.x;
        nxt = nxt < p.npairs ? nxt : p.npairs - 1;
        const long long nA = 2 * nxt, nB = (nA + 1 < p.M) ? nA + 1 : nA;
        float nb[16];
        double q[4];
        q[0] = sA1;
        q[1] = sA2;
        reduce_series(raw, k0, nb, q[2], q[3]);
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
        const int eA = (int)((__double_as_longlong(varA) >> 52) & 0x7ff) - 1023;
        const int eB = (int)((__double_as_longlong(varB) >> 52) & 0x7ff) - 1023;
        const bool nanA = !__builtin_isfinite(varA), nanB = !__builtin_isfinite(varB);
        const bool redoA = END >= 2 && !(!(varA > 0.0) || nanA) && (eA > 200 || eA < -200 || mA * mA > 64.0 * varA);
        const bool redoB = END >= 2 && !(!(varB > 0.0) || nanB) && (eB > 200 || eB < -200 || mB * mB > 64.0 * varB);
        const bool deadA = !(varA > 0.0) || nanA || redoA, deadB = !(varB > 0.0) || nanB || redoB || !hasB;
        const float sclA = deadA ? 0.f : __int_as_float((127 - (eA >> 1)) << 23);
        const float sclB = deadB ? 0.f : __int_as_float((127 - (eB >> 1)) << 23);
        const float mAf = deadA ? 0.f : (float)mA, mBf = deadB ? 0.f : (float)mB;
        f2 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = mk2((na[i] - mAf) * sclA, (nb[i] - mBf) * sclB);
        if (END >= 2 && (deadA || deadB)) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v[i].x = deadA ? 0.f : v[i].x;
                v[i].y = deadB ? 0.f : v[i].y;
            }
        }
        fence();
        f2 xq[16];
        {
            const Tw1FetchF fetch{p.xcf, t};
#pragma unroll
            for (int k = 0; k < 16; k++)
                xq[k] = fetch(k);
        }
        fence();
        w4::fft<true>(v, xbuf, tw2s, w1, w2, w4_, w8, xq, t, NoHook(), NoHook());
        fence();
        issue_series(raw, k0, p.rows + nA * p.stride, t); // series A of the next pair, behind the spectrum factors
        fence();
        const double *rowB = p.rows + nB * p.stride;
        w4::fft<false>(v, xbuf, tw2s, w1, w2, w4_, w8, xq, t, NoHook(), [&]() {
            reduce_series(raw, k0, na, sA1, sA2);
            fence();
            if (END != 3)
                issue_series(raw, k0, rowB, t);
        });
        if (END == 0) { // ---- bare maxima (timing experiment: estimate only)
            float ma = 0.f, mb = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                ma = fmaxf(ma, fabsf(v[k].x));
                mb = fmaxf(mb, fabsf(v[k].y));
            }
            ma = wave_max_f32_dpp(ma);
            mb = wave_max_f32_dpp(mb);
            if (lane == 0) {
                redf[wave] = ma;
                redf[4 + wave] = mb;
            }
            lds_barrier();
            if (t < 2) {
                const float M = fmaxf(fmaxf(redf[4 * t], redf[4 * t + 1]), fmaxf(redf[4 * t + 2], redf[4 * t + 3]));
                if (t == 0 || hasB)
                    p.mv[rA + t] = (double)M;
            }
        } else { // ---- the full end phase: in / out maxima per series folded before the barrier, one writer per row
            int to = t;
            asm volatile("" : "+v"(to));
            const int klo = max_lag >= to ? (max_lag - to) >> 8 : -1;
            const int khi = (4096 - max_lag - to + 255) >> 8;
            float inA, outA, inB, outB;
            {
                const float a0 = fabsf(v[0].x), a15 = fabsf(v[15].x), b0 = fabsf(v[0].y), b15 = fabsf(v[15].y);
                const bool i0 = klo >= 0, i15 = khi <= 15;
                inA = fmaxf(i0 ? a0 : -1.f, i15 ? a15 : -1.f);
                outA = fmaxf(i0 ? -1.f : a0, i15 ? -1.f : a15);
                inB = fmaxf(i0 ? b0 : -1.f, i15 ? b15 : -1.f);
                outB = fmaxf(i0 ? -1.f : b0, i15 ? -1.f : b15);
#pragma unroll
                for (int k = 1; k < 15; k++) {
                    outA = fmaxf(outA, fabsf(v[k].x));
                    outB = fmaxf(outB, fabsf(v[k].y));
                }
            }
            if (END == 3) {
                fence();
                issue_series(raw, k0, rowB, t);
                fence();
            }
            const float wiA = wave_max_f32_dpp(inA + 1.f), woA = wave_max_f32_dpp(outA + 1.f);
            const float wiB = wave_max_f32_dpp(inB + 1.f), woB = wave_max_f32_dpp(outB + 1.f);
            if (lane == 0) {
                redf[4 * wave + 0] = wiA;
                redf[4 * wave + 1] = woA;
                redf[16 + 4 * wave + 0] = wiB;
                redf[16 + 4 * wave + 1] = woB;
            }
            lds_barrier();
            if (t < 2 && (t == 0 || hasB)) {
                const float *r = redf + 16 * t;
                const float in = fmaxf(fmaxf(r[0], r[4]), fmaxf(r[8], r[12])) - 1.f, out = fmaxf(fmaxf(r[1], r[5]), fmaxf(r[9], r[13])) - 1.f;
                const float M = fmaxf(in, out), th = M - window;
                unsigned f = (in >= th ? SCR_IN : 0u) | (out >= th ? SCR_OUT : 0u) | SCR_POS | SCR_NEG;
                const int e = t ? eB : eA;
                double est = (double)M * __longlong_as_double((long long)(1023 + (e >> 1)) << 52);
                const bool off = t ? deadB : deadA;
                if (off) {
                    est = (t ? nanB : nanA) ? __builtin_nan("") : 0.0;
                    f = (t ? nanB : nanA) ? SCR_NAN : ((t ? redoB : redoA) ? SCR_REFINE : SCR_IN);
                }
                p.mv[rA + t] = est;
                p.scr_flags[rA + t] = f;
                p.scr_var[rA + t] = t ? varB : varA;
            }
        }
    }
}

__global__ void fill(double* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5; } }
template <typename K> void run(const char* title, K kern, FusedParams p, int grid)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int l = 0; l < 5; l++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s grid=%d: %.3f ms -> %.1f%% of 8 TB/s\n", title, grid, ms, p.M * 32784.0 / (ms * 1e-3) / 8e12 * 100);
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
    unsigned* fl; CK(hipMalloc(&fl, M * 4)); CK(hipMemset(fl, 0, M * 4)); p.scr_flags = fl; double* sv; CK(hipMalloc(&sv, M * 8)); p.scr_var = sv; p.scr_max_lag = 15; p.screen_delta = 1e-3;
    CK(hipDeviceSynchronize());
    run("library screening pass, 3 WG/CU", xcorr_screen_pass_n4096<3, false, false, true, false>, p, 256 * 3);
    run("one-series prefetch, 4 WG/CU (128 VGPRs), bare end", screen_w4<4, 0>, p, 256 * 4);
    run("one-series prefetch, 3 WG/CU (168 VGPRs), bare end", screen_w4<3, 0>, p, 256 * 3);
    run("one-series prefetch, 4 WG/CU, full end phase", screen_w4<4, 1>, p, 256 * 4);
    run("one-series prefetch, 3 WG/CU, full end phase", screen_w4<3, 1>, p, 256 * 3);
    run("one-series prefetch, 4 WG/CU, full end phase + trust rules", screen_w4<4, 2>, p, 256 * 4);
    run("one-series prefetch, 4 WG/CU, series B requested after the fold", screen_w4<4, 3>, p, 256 * 4);
    run("one-series prefetch, 3 WG/CU, series B requested after the fold", screen_w4<3, 3>, p, 256 * 3);
    run("one-series prefetch, 3 WG/CU, full end phase + trust rules", screen_w4<3, 2>, p, 256 * 3);
    return 0;
}
