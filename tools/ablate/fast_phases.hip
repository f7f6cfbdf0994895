// Phase-stamp harness for the xcorr_r16_fast.hip kernel (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../go-muse_amd/csrc/xcorr_r16_fast.hip"
using namespace muse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(double* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5; } }
template <int WPS> void go(FusedParams p, int grid)
{
    const char* names[16] = {"row load wait", "shift+sumsq", "F1 pass1+tw1", "F1 xchg A (cross)+finalize", "F1 pass2+tw2", "barrier + F1 xchg B (local)", "F1 pass3 + xc", "F2 pass1+tw1", "F2 xchg A (local)", "F2 pass2+tw2", "F2 xchg B (cross)", "prefetch + F2 pass3", "argmax+store", "", "", ""};
    unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)grid * 4 * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8)); p.dbg = dbg;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(p.ovf_count, 0, 4));
    hipLaunchKernelGGL((xcorr_fused_n4096_fast<WPS, true>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8));
    CK(hipMemset(p.ovf_count, 0, 4));
    const int loops = getenv("LOOPS") ? atoi(getenv("LOOPS")) : 1; // LOOPS=N: run long enough to sample clocks / power
    for (int l = 1; l < loops; l++) {
        CK(hipMemsetAsync(p.ovf_count, 0, 4));
        hipLaunchKernelGGL((xcorr_fused_n4096_fast<WPS, true>), dim3(grid), dim3(256), 0, 0, p);
    }
    if (loops > 1) { CK(hipDeviceSynchronize()); CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8)); CK(hipMemset(p.ovf_count, 0, 4)); }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((xcorr_fused_n4096_fast<WPS, true>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4 * 16);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double pairs_per_wg = (double)p.npairs / grid;
    printf("FAST WPS=%d grid=%d: %.3f ms (stamped build), %.1f pairs per workgroup\n", WPS, grid, ms, pairs_per_wg);
    double tot = 0; double s[16] = {0};
    for (int w = 0; w < grid * 4; w++) for (int i = 0; i < 16; i++) s[i] += (double)h[(size_t)w * 16 + i];
    for (int i = 0; i < 13; i++) tot += s[i];
    for (int i = 0; i < 13; i++) printf("  %-30s %9.0f ticks/pair/wave  %5.1f%%\n", names[i], s[i] / (grid * 4) / pairs_per_wg, 100.0 * s[i] / tot);
    printf("  total %.0f ticks/pair/wave\n", tot / (grid * 4) / pairs_per_wg);
}
int main(int argc, char** argv)
{
    long long M = argc > 1 ? atoll(argv[1]) : 1000000;
    FusedParams p{}; p.M = M; p.stride = 4096; p.npairs = M / 2; p.N = 4096; p.n = 4096; p.logn = 12; p.normalize_y = 1;
    double* rows; CK(hipMalloc(&rows, M * 4096 * 8)); p.rows = rows;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, rows, M * 4096);
    std::vector<double2> t1(4096), t2(256), xc(4096);
    for (int k = 0; k < 16; k++) for (int t = 0; t < 256; t++) { double a = -2 * M_PI * ((k * t) % 4096) / 4096.0; t1[k * 256 + t] = make_double2(cos(a), sin(a)); }
    for (int k = 0; k < 16; k++) for (int c = 0; c < 16; c++) { double a = -2 * M_PI * ((k * c) % 256) / 256.0; t2[k * 16 + c] = make_double2(cos(a), sin(a)); }
    for (int f = 0; f < 4096; f++) xc[f] = make_double2(cos(0.001 * f) / 4096, sin(0.002 * f) / 4096);
    double2 *d1, *d2, *dx; CK(hipMalloc(&d1, 4096 * 16)); CK(hipMalloc(&d2, 256 * 16)); CK(hipMalloc(&dx, 4096 * 16));
    CK(hipMemcpy(d1, t1.data(), 4096 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, t2.data(), 256 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, xc.data(), 4096 * 16, hipMemcpyHostToDevice));
    p.tw1 = d1; p.tw1p = d1; p.tw2 = d2; p.xc = dx; p.xcp = dx;
    CK(hipMalloc(&p.mv, M * 8)); CK(hipMalloc(&p.lag, M * 4));
    CK(hipMalloc(&p.ovf_count, 4)); CK(hipMalloc(&p.ovf_list, p.npairs * 16));
    CK(hipDeviceSynchronize());
    go<3>(p, 256 * 3 * 16);
    go<4>(p, 256 * 4 * 16);
    go<4>(p, 256 * 4);
    return 0;
}
