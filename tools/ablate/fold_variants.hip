// A/B harness for measurement builds of xcorr_r16_fold.hip (the kernel's VAR template parameter): the variants run
// interleaved on ONE box over the same resident rows, and every variant's (lag, mv) output is compared with variant 0's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "../../go-muse_amd/csrc/xcorr_r16_fold.hip"
using namespace muse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(double* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5; } }
static double2 tw(long long num, long long den) { num %= den; double a = -2 * M_PI * (double)num / (double)den; return make_double2(cos(a), sin(a)); }
static void fill_g(std::vector<double2>& g, size_t stride, size_t idx, long long u)
{
    g[0 * stride + idx] = tw(u, 512); g[1 * stride + idx] = tw(u, 1024); g[2 * stride + idx] = tw(u, 2048); g[3 * stride + idx] = tw(u + 256, 2048);
    for (int q = 0; q < 4; q++) g[(4 + q) * stride + idx] = tw(u + 256 * q, 4096);
}
template <int VAR> static void launch(const FusedParams& p, int grid)
{
    CK(hipMemsetAsync(p.ovf_count, 0, 8));
    hipLaunchKernelGGL((xcorr_fused_n4096_fold<false, false, false, VAR>), dim3(grid), dim3(256), 0, 0, p);
}
typedef void (*launch_fn)(const FusedParams&, int);
int main(int argc, char** argv)
{
    long long M = argc > 1 ? atoll(argv[1]) : 1000000;
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 7;
    FusedParams p{}; p.M = M; p.stride = 4096; p.npairs = M / 2; p.N = 4096; p.n = 4096; p.logn = 12; p.normalize_y = 1;
    double* rows; CK(hipMalloc(&rows, M * 4096 * 8)); p.rows = rows;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, rows, M * 4096);
    std::vector<double2> g2(128), g3a(2048), g3b(2048), xc(4096);
    for (int j = 0; j < 16; j++) fill_g(g2, 16, j, 16 * j);
    for (int t = 0; t < 256; t++) { fill_g(g3a, 256, t, (t >> 4) + 16 * (t & 15)); fill_g(g3b, 256, t, t); }
    for (int f = 0; f < 4096; f++) xc[f] = make_double2(cos(0.001 * f) / 4096, sin(0.002 * f) / 4096);
    double2 *d2, *d3a, *d3b, *dx; CK(hipMalloc(&d2, 128 * 16)); CK(hipMalloc(&d3a, 2048 * 16)); CK(hipMalloc(&d3b, 2048 * 16)); CK(hipMalloc(&dx, 4096 * 16));
    CK(hipMemcpy(d2, g2.data(), 128 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d3a, g3a.data(), 2048 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d3b, g3b.data(), 2048 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, xc.data(), 4096 * 16, hipMemcpyHostToDevice));
    p.g2 = d2; p.g3a = d3a; p.g3b = d3b; p.xc = dx; p.xcp = dx;
    CK(hipMalloc(&p.mv, M * 8)); CK(hipMalloc(&p.lag, M * 4));
    CK(hipMalloc(&p.ovf_count, 8)); CK(hipMalloc(&p.ovf_list, p.npairs * 16)); p.work_counter = p.ovf_count + 1;
    CK(hipDeviceSynchronize());
    int cus = 256; { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount; }
    const int grid = cus * 4;
    const int vars[] = {0, 7, 8, 15, 2};
    const launch_fn fns[] = {launch<0>, launch<7>, launch<8>, launch<15>, launch<2>};
    const int NV = sizeof(vars) / sizeof(vars[0]);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> mv0(M), mv(M);
    std::vector<int> lag0(M), lag(M);
    std::vector<std::vector<float>> ts(NV);
    for (int v = 0; v < NV; v++) { // warm-up + output check
        fns[v](p, grid);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(mv.data(), p.mv, M * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(lag.data(), p.lag, M * 4, hipMemcpyDeviceToHost));
        if (v == 0) { mv0 = mv; lag0 = lag; continue; }
        double worst = 0; long long badlag = 0;
        for (long long i = 0; i < M; i++) {
            worst = std::max(worst, std::fabs(mv[i] - mv0[i]) / std::max(std::fabs(mv0[i]), 1e-300));
            badlag += lag[i] != lag0[i];
        }
        printf("VAR %d vs 0: max rel score diff %.3e, lag mismatches %lld of %lld\n", vars[v], worst, badlag, M);
    }
    for (int r = 0; r < reps; r++)
        for (int v = 0; v < NV; v++) {
            CK(hipMemsetAsync(p.ovf_count, 0, 8));
            CK(hipEventRecord(e0));
            fns[v](p, grid);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts[v].push_back(ms);
        }
    for (int v = 0; v < NV; v++) {
        std::sort(ts[v].begin(), ts[v].end());
        printf("VAR %d: median %.3f ms  min %.3f ms  (%.2f%% of 8 TB/s)\n", vars[v], ts[v][ts[v].size() / 2], ts[v][0], M * 32784.0 / (ts[v][ts[v].size() / 2] * 1e-3) / 8e12 * 100);
    }
    return 0;
}
