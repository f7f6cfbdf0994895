// Ablation harness (diagnostic only, not part of the product): the first
// n = 4096 kernel with compile-time switches that remove one cost at a time,
// timed in one process on the same synthetic matrix.  Outputs are wrong under
// ablation by design; only the time matters (values are kept live).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../go-muse_amd/csrc/fft_device.h"

using namespace muse;

enum { NOLOAD = 1, NOTW = 2, NOXC = 4, NOXCHG = 8, NOBAR = 16, NOZN = 32, NOARG = 64, NODFT = 128, NTLOAD = 256, LOAD16 = 512 };

struct P { const double* rows; long long M, stride, npairs; const double2 *xc, *tw1, *tw2; double* mv; int* lag; };

template <int ABL> __device__ __forceinline__ void bar() { if (!(ABL & NOBAR)) __syncthreads(); }

template <int ABL>
__device__ __forceinline__ void fft(double2 (&v)[16], double2* lds, const double2* __restrict__ tw1, const double2* __restrict__ tw2, int t, double2 seed)
{
    const int hi = t >> 4, lo = t & 15;
    if (!(ABL & NODFT)) dft16(v);
#pragma unroll
    for (int k = 1; k < 16; k++) {
        double2 w = (ABL & NOTW) ? make_double2(seed.x + k * 1e-9, seed.y) : tw1[k * 256 + t];
        v[P16(k)] = cmul(v[P16(k)], w);
    }
    bar<ABL>();
    if (!(ABL & NOXCHG)) {
#pragma unroll
        for (int k = 0; k < 16; k++) lds[256 * k + t] = v[P16(k)];
    }
    bar<ABL>();
    if (!(ABL & NOXCHG)) {
#pragma unroll
        for (int b = 0; b < 16; b++) v[b] = lds[256 * hi + 16 * b + lo];
    }
    if (!(ABL & NODFT)) dft16(v);
#pragma unroll
    for (int k = 1; k < 16; k++) {
        double2 w = (ABL & NOTW) ? make_double2(seed.y + k * 1e-9, seed.x) : tw2[k * 16 + lo];
        v[P16(k)] = cmul(v[P16(k)], w);
    }
    bar<ABL>();
    if (!(ABL & NOXCHG)) {
#pragma unroll
        for (int k = 0; k < 16; k++) lds[272 * k + 17 * hi + lo] = v[P16(k)];
    }
    bar<ABL>();
    if (!(ABL & NOXCHG)) {
#pragma unroll
        for (int c = 0; c < 16; c++) v[c] = lds[272 * hi + 17 * lo + c];
    }
    if (!(ABL & NODFT)) dft16(v);
    double2 w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = v[P16(k)];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = w[k];
}

template <int ABL>
__global__ __launch_bounds__(256, 2) void kern(const P p)
{
    __shared__ double2 lds[16 * 272];
    __shared__ double red[64];
    __shared__ int redi[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int N = 4096;
    const double2 seed = p.tw1[256 + t];
    for (long long pair = blockIdx.x; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const double* __restrict__ ra = p.rows + rA * p.stride;
        const double* __restrict__ rb = p.rows + rB * p.stride;
        double2 v[16];
        if (ABL & LOAD16) {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const double2 xa = *reinterpret_cast<const double2*>(ra + 2 * t + 512 * a);
                const double2 xb = *reinterpret_cast<const double2*>(rb + 2 * t + 512 * a);
                v[2 * a] = make_double2(xa.x, xb.x);
                v[2 * a + 1] = make_double2(xa.y, xb.y);
            }
        } else
#pragma unroll
        for (int a = 0; a < 16; a++) {
            const int j = t + 256 * a;
            if (ABL & NOLOAD) v[a] = make_double2(seed.x * (a + 1) + (double)pair, seed.y * (a + 2));
            else if (ABL & NTLOAD) v[a] = make_double2(__builtin_nontemporal_load(ra + j), __builtin_nontemporal_load(rb + j));
            else v[a] = make_double2(ra[j], rb[j]);
        }
        double ia = 1.0, ib = 1.0;
        if (!(ABL & NOZN)) {
            double s[2] = {0.0, 0.0};
#pragma unroll
            for (int a = 0; a < 16; a++) { s[0] += v[a].x; s[1] += v[a].y; }
            block_sum<2>(s, red);
            const double ca = -s[0] / (double)N, cb = -s[1] / (double)N;
            double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int a = 0; a < 16; a++) {
                v[a].x += ca; v[a].y += cb;
                q[0] += v[a].x; q[1] = fma(v[a].x, v[a].x, q[1]);
                q[2] += v[a].y; q[3] = fma(v[a].y, v[a].y, q[3]);
            }
            block_sum<4>(q, red + 8);
            ZnFlags fa, fb;
            ia = zn_scale(q[0], q[1], N, fa);
            ib = zn_scale(q[2], q[3], N, fb);
        }
#pragma unroll
        for (int a = 0; a < 16; a++) { v[a].x *= ia; v[a].y *= ib; }
        fft<ABL>(v, lds, p.tw1, p.tw2, t, seed);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            double2 w = (ABL & NOXC) ? make_double2(seed.x - k * 1e-9, seed.y) : p.xc[t + 256 * k];
            v[k] = cmul(v[k], w);
        }
        fft<ABL>(v, lds, p.tw1, p.tw2, t, seed);
        double ma = 0.0, mb = 0.0, sa = 0.0, sb = 0.0;
        int ka = 0, kb = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const double aa = fabs(v[k].x), ab = fabs(v[k].y);
            if (aa > ma) { ma = aa; sa = v[k].x; ka = k; }
            if (ab > mb) { mb = ab; sb = v[k].y; kb = k; }
        }
        if (ABL & NOARG) {
            if (ma + mb == 12345.678) { p.mv[rA] = sa + sb; p.lag[rA] = ka + kb; }
        } else {
            double wa = wave_max(ma), wb = wave_max(mb);
            if (lane == 0) { red[32 + wave] = wa; red[36 + wave] = wb; }
            __syncthreads();
            const double MA = fmax(fmax(red[32], red[33]), fmax(red[34], red[35]));
            const double MB = fmax(fmax(red[36], red[37]), fmax(red[38], red[39]));
            int ca_i = (ma == MA && MA > 0.0) ? (t + 256 * ka) : 0x7fffffff;
            int cb_i = (mb == MB && MB > 0.0) ? (t + 256 * kb) : 0x7fffffff;
            ca_i = wave_min_i(ca_i); cb_i = wave_min_i(cb_i);
            if (lane == 0) { redi[wave] = ca_i; redi[4 + wave] = cb_i; }
            __syncthreads();
            const int IA = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
            const int IB = min(min(redi[4], redi[5]), min(redi[6], redi[7]));
            if (t == (IA & 255)) { p.mv[rA] = sa; p.lag[rA] = IA; }
            if (t == (IB & 255)) { p.mv[rB] = sb; p.lag[rB] = IB; }
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ABL> float run(const P& p, int grid, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern<ABL>, dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; i++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<ABL>, dim3(grid), dim3(256), 0, 0, p);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

__global__ void fill(double* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5; } }

int main(int argc, char** argv)
{
    long long M = argc > 1 ? atoll(argv[1]) : 1000000;
    P p{}; p.M = M; p.stride = 4096; p.npairs = M / 2;
    double* rows; CK(hipMalloc(&rows, M * 4096 * 8)); p.rows = rows;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, rows, M * 4096);
    std::vector<double2> t1(4096), t2(256), xc(4096);
    for (int k = 0; k < 16; k++) for (int t = 0; t < 256; t++) { double a = -2 * M_PI * ((k * t) % 4096) / 4096.0; t1[k * 256 + t] = make_double2(cos(a), sin(a)); }
    for (int k = 0; k < 16; k++) for (int c = 0; c < 16; c++) { double a = -2 * M_PI * ((k * c) % 256) / 256.0; t2[k * 16 + c] = make_double2(cos(a), sin(a)); }
    for (int f = 0; f < 4096; f++) xc[f] = make_double2(cos(0.001 * f) / 4096, sin(0.002 * f) / 4096);
    double2 *d1, *d2, *dx; CK(hipMalloc(&d1, 4096 * 16)); CK(hipMalloc(&d2, 256 * 16)); CK(hipMalloc(&dx, 4096 * 16));
    CK(hipMemcpy(d1, t1.data(), 4096 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, t2.data(), 256 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, xc.data(), 4096 * 16, hipMemcpyHostToDevice));
    p.tw1 = d1; p.tw2 = d2; p.xc = dx;
    CK(hipMalloc(&p.mv, M * 8)); CK(hipMalloc(&p.lag, M * 4));
    CK(hipDeviceSynchronize());
    const int grid = 4096, it = 5;
#define R(name, abl) printf("%-44s %8.3f ms\n", name, run<abl>(p, grid, it)); fflush(stdout);
    R("baseline", 0)
    R("nontemporal row loads", NTLOAD)
    R("-row loads", NOLOAD)
    R("-twiddle loads", NOTW)
    R("-xc loads", NOXC)
    R("-twiddle -xc loads", NOTW | NOXC)
    R("-LDS exchange (barriers kept)", NOXCHG)
    R("-FFT barriers", NOBAR)
    R("-LDS exchange -FFT barriers", NOXCHG | NOBAR)
    R("-znorm reductions", NOZN)
    R("-argmax reductions", NOARG)
    R("-znorm -argmax reductions", NOZN | NOARG)
    R("-dft16 math", NODFT)
    R("-loads -tw -xc (all global reads)", NOLOAD | NOTW | NOXC)
    R("-all global -zn -arg", NOLOAD | NOTW | NOXC | NOZN | NOARG)
    R("-all global -zn -arg -xchg -bar (VALU only)", NOLOAD | NOTW | NOXC | NOZN | NOARG | NOXCHG | NOBAR)
    R("only loads+zn+arg (no dft, tw, xc, xchg, bar)", NODFT | NOTW | NOXC | NOXCHG | NOBAR)
    R("only loads16+zn+arg", NODFT | NOTW | NOXC | NOXCHG | NOBAR | LOAD16)
    R("only loads (x2), nothing else", NODFT | NOTW | NOXC | NOXCHG | NOBAR | NOZN | NOARG)
    R("only loads16, nothing else", NODFT | NOTW | NOXC | NOXCHG | NOBAR | NOZN | NOARG | LOAD16)
    R("baseline with 16-B loads", LOAD16)
    return 0;
}
