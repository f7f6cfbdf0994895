// Harness for xcorr_r16_fold.hip (diagnostic only): un-stamped timing (median of REPS launches) and in-kernel phase stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "../../go-muse_amd/csrc/xcorr_r16_fold.hip"
using namespace muse;
#ifndef HARNESS_F32
#define HARNESS_F32 0
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename T> __global__ void fill(T* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (T)((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5); } }
static double2 tw(long long num, long long den) { num %= den; double a = -2 * M_PI * (double)num / (double)den; return make_double2(cos(a), sin(a)); }
static void fill_g(std::vector<double2>& g, size_t stride, size_t idx, long long u)
{
    g[0 * stride + idx] = tw(u, 512); g[1 * stride + idx] = tw(u, 1024); g[2 * stride + idx] = tw(u, 2048); g[3 * stride + idx] = tw(u + 256, 2048);
    for (int q = 0; q < 4; q++) g[(4 + q) * stride + idx] = tw(u + 256 * q, 4096);
}
template <bool T> void launch(const FusedParams& p, int grid)
{
    CK(hipMemsetAsync(p.ovf_count, 0, 8));
    hipLaunchKernelGGL((xcorr_fused_n4096_fold<T, false, HARNESS_F32 != 0>), dim3(grid), dim3(256), 0, 0, p);
}
int main(int argc, char** argv)
{
    long long M = argc > 1 ? atoll(argv[1]) : 1000000;
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 9;
    FusedParams p{}; p.M = M; p.stride = 4096; p.npairs = M / 2; p.N = 4096; p.n = 4096; p.logn = 12; p.normalize_y = 1;
#if HARNESS_F32
    float* rows; CK(hipMalloc(&rows, M * 4096 * 4)); p.rows32 = rows;
    hipLaunchKernelGGL(fill<float>, dim3(4096), dim3(256), 0, 0, rows, M * 4096);
#else
    double* rows; CK(hipMalloc(&rows, M * 4096 * 8)); p.rows = rows;
    hipLaunchKernelGGL(fill<double>, dim3(4096), dim3(256), 0, 0, rows, M * 4096);
#endif
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
    const int wpc = getenv("WPC") ? atoi(getenv("WPC")) : 4;
    const int grid = cus * wpc;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch<false>(p, grid); launch<false>(p, grid); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) {
        CK(hipMemsetAsync(p.ovf_count, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((xcorr_fused_n4096_fold<false, false, HARNESS_F32 != 0>), dim3(grid), dim3(256), 0, 0, p);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("FOLD grid=%d (%d WG/CU): median %.3f ms  min %.3f ms  (%.1f%% of 8 TB/s on %d B per series)\n", grid, wpc, ts[ts.size() / 2], ts[0], M * (HARNESS_F32 ? 16400.0 : 32784.0) / (ts[ts.size() / 2] * 1e-3) / 8e12 * 100, HARNESS_F32 ? 16400 : 32784);
    if (getenv("NOSTAMP")) return 0;
    const char* names[16] = {"row load wait", "shift+sumsq", "F1 pass1", "F1 xchg A (cross)+finalize", "F1 pass2 (g)", "barrier + F1 xchg B (local)", "F1 pass3 (g) + DC", "F2 pass1 + xc", "F2 xchg A (local)", "F2 pass2 (g)", "F2 xchg B (cross)", "F2 pass3 (g)", "row request (after stamp 13)", "argmax+store", "", ""};
    unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)grid * 4 * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8)); p.dbg = dbg;
    launch<true>(p, grid); CK(hipDeviceSynchronize());
    CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8));
    CK(hipMemsetAsync(p.ovf_count, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((xcorr_fused_n4096_fold<true, false, HARNESS_F32 != 0>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4 * 16);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double pairs_per_wg = (double)p.npairs / grid;
    printf("stamped build: %.3f ms, %.1f pairs per workgroup\n", ms, pairs_per_wg);
    double tot = 0; double s[16] = {0};
    for (int w = 0; w < grid * 4; w++) for (int i = 0; i < 16; i++) s[i] += (double)h[(size_t)w * 16 + i];
    for (int i = 0; i < 14; i++) tot += s[i];
    for (int i = 0; i < 14; i++) printf("  %-30s %9.0f ticks/pair/wave  %5.1f%%\n", names[i], s[i] / (grid * 4) / pairs_per_wg, 100.0 * s[i] / tot);
    printf("  total %.0f ticks/pair/wave\n", tot / (grid * 4) / pairs_per_wg);
    return 0;
}
