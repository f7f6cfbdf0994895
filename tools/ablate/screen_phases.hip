// Phase-stamp harness for the fp32 screening kernels of xcorr_r16_screen.hip (diagnostic only).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I go-muse_amd/csrc tools/ablate/screen_phases.hip -o tools/ablate/screen_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../go-muse_amd/csrc/xcorr_r16_screen.hip"
using namespace muse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(double* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5; } }
template <typename K> void run(const char* title, K kern, FusedParams p, int grid, const char* const* names, int nph)
{
    unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)grid * 4 * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8)); p.dbg = dbg;
    CK(hipMemset(p.ovf_count, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    const int loops = getenv("LOOPS") ? atoi(getenv("LOOPS")) : 3;
    for (int l = 1; l < loops; l++) { CK(hipMemsetAsync(p.ovf_count, 0, 4)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p); }
    CK(hipDeviceSynchronize());
    CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8));
    CK(hipMemset(p.ovf_count, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4 * 16);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    int ovf = 0; CK(hipMemcpy(&ovf, p.ovf_count, 4, hipMemcpyDeviceToHost));
    double pairs_per_wg = (double)p.npairs / grid;
    printf("%s grid=%d: %.3f ms (stamped build), %.1f pairs per workgroup, overflow pairs %d\n", title, grid, ms, pairs_per_wg, ovf);
    double tot = 0; double s[16] = {0};
    for (int w = 0; w < grid * 4; w++) for (int i = 0; i < 16; i++) s[i] += (double)h[(size_t)w * 16 + i];
    for (int i = 0; i < nph; i++) tot += s[i];
    for (int i = 0; i < nph; i++) printf("  %-28s %9.0f ticks/pair/wave  %5.1f%%\n", names[i], s[i] / (grid * 4) / pairs_per_wg, 100.0 * s[i] / tot);
    printf("  total %.0f ticks/pair/wave (100 MHz ticks)\n", tot / (grid * 4) / pairs_per_wg);
    CK(hipFree(dbg));
}
int main(int argc, char** argv)
{
    long long M = argc > 1 ? atoll(argv[1]) : 1000000;
    FusedParams p{}; p.M = M; p.stride = 4096; p.npairs = M / 2; p.N = 4096; p.n = 4096; p.logn = 12; p.normalize_y = 1;
    double* rows; CK(hipMalloc(&rows, M * 4096 * 8)); p.rows = rows;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, rows, M * 4096);
    std::vector<float2> t1f(4096), t2f(256), xcf(4096); std::vector<double> xs(4096);
    for (int k = 0; k < 16; k++) for (int t = 0; t < 256; t++) { double a = -2 * M_PI * ((k * t) % 4096) / 4096.0; t1f[k * 256 + t] = make_float2((float)cos(a), (float)sin(a)); }
    for (int k = 0; k < 16; k++) for (int c = 0; c < 16; c++) { double a = -2 * M_PI * ((k * c) % 256) / 256.0; t2f[k * 16 + c] = make_float2((float)cos(a), (float)sin(a)); }
    for (int i = 0; i < 4096; i++) { xcf[i] = make_float2((float)(cos(0.001 * i) / 4096), (float)(sin(0.002 * i) / 4096)); xs[i] = sin(0.01 * i) / 64.0; }
    float2 *f1, *f2_, *fx; double* dxs; CK(hipMalloc(&f1, 4096 * 8)); CK(hipMalloc(&f2_, 256 * 8)); CK(hipMalloc(&fx, 4096 * 8)); CK(hipMalloc(&dxs, 4096 * 8));
    CK(hipMemcpy(f1, t1f.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(f2_, t2f.data(), 256 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(fx, xcf.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dxs, xs.data(), 4096 * 8, hipMemcpyHostToDevice));
    p.tw1f = f1; p.tw2f = f2_; p.xcf = fx; p.xs = dxs; p.screen_delta = 1e-4;
    CK(hipMalloc(&p.mv, M * 8)); CK(hipMalloc(&p.lag, M * 4));
    CK(hipMalloc(&p.ovf_count, 4)); CK(hipMalloc(&p.ovf_list, p.npairs * 8));
    CK(hipDeviceSynchronize());
    const char* n1[16] = {"row load wait", "stats+convert", "fp32 FFT1 (+xc)", "fp32 FFT2", "max+candidates", "fp64 re-eval", "result store"};
    const char* n2[16] = {"series B wait", "stats+convert", "issue A' + FFT1 (+ wait A')", "reduce A', issue B'", "FFT2", "max+candidates", "re-eval", "result store"};
    const int which = argc > 2 ? atoi(argv[2]) : 7;
    if (which & 1) run("SCREEN gen1", xcorr_fused_n4096_screen<false, true>, p, 256 * 3, n1, 7);
    if (which & 2) run("SCREEN gen2 WPC=2", xcorr_fused_n4096_screen2<2, true>, p, 256 * 2, n2, 8);

    return 0;
}
