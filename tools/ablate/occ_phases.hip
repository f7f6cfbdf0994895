// Phase-stamp harness for the shipped occ4 kernel (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../go-muse_amd/csrc/xcorr_r16_occ4.hip"
#include "../../go-muse_amd/csrc/xcorr_r16_screen.hip"
using namespace muse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(double* r, long long n) { for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; r[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5; } }
template <int WPS> void go(FusedParams p, int grid)
{
    const char* names[16] = {"row load wait", "znorm+barrier", "F1 pass1+tw1", "F1 exchange A", "F1 pass2+tw2", "F1 exchange B", "F1 pass3", "xc multiply", "F2 pass1+tw1", "F2 exchange A", "F2 pass2+tw2", "F2 exchange B", "F2 pass3", "argmax+store", "", ""};
    unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)grid * 4 * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8)); p.dbg = dbg;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((xcorr_fused_n4096_occ4<false, WPS, true>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((xcorr_fused_n4096_occ4<false, WPS, true>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4 * 16);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double pairs_per_wg = (double)p.npairs / grid;
    printf("WPS=%d grid=%d: %.3f ms (stamped build), %.1f pairs per workgroup\n", WPS, grid, ms, pairs_per_wg);
    double tot = 0; double s[16] = {0};
    for (int w = 0; w < grid * 4; w++) for (int i = 0; i < 16; i++) s[i] += (double)h[(size_t)w * 16 + i];
    for (int i = 0; i < 14; i++) tot += s[i];
    for (int i = 0; i < 14; i++) printf("  %-16s %9.0f ticks/pair/wave  %5.1f%%\n", names[i], s[i] / (grid * 4) / pairs_per_wg, 100.0 * s[i] / tot);
    printf("  total %.0f ticks/pair/wave\n", tot / (grid * 4) / pairs_per_wg);
}
void go_screen(FusedParams p, int grid)
{
    const char* names[16] = {"row load wait", "stats+convert", "fp32 FFT1 (+xc)", "fp32 FFT2", "max+candidates", "fp64 re-eval", "result store", "", "", "", "", "", "", "", "", ""};
    unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)grid * 4 * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 4 * 16 * 8)); p.dbg = dbg;
    CK(hipMemset(p.ovf_count, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((xcorr_fused_n4096_screen<false, true>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipMemset(p.ovf_count, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((xcorr_fused_n4096_screen<false, true>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4 * 16);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    int ovf = 0; CK(hipMemcpy(&ovf, p.ovf_count, 4, hipMemcpyDeviceToHost));
    double pairs_per_wg = (double)p.npairs / grid;
    printf("SCREEN grid=%d: %.3f ms (stamped build), overflow pairs %d\n", grid, ms, ovf);
    double tot = 0; double s[16] = {0};
    for (int w = 0; w < grid * 4; w++) for (int i = 0; i < 16; i++) s[i] += (double)h[(size_t)w * 16 + i];
    for (int i = 0; i < 7; i++) tot += s[i];
    for (int i = 0; i < 7; i++) printf("  %-16s %9.0f ticks/pair/wave  %5.1f%%\n", names[i], s[i] / (grid * 4) / pairs_per_wg, 100.0 * s[i] / tot);
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
    p.tw1 = d1; p.tw2 = d2; p.xc = dx;
    CK(hipMalloc(&p.mv, M * 8)); CK(hipMalloc(&p.lag, M * 4));
    CK(hipDeviceSynchronize());
    std::vector<float2> t1f(4096), t2f(256), xcf(4096); std::vector<double> xs(4096);
    for (int i = 0; i < 4096; i++) { t1f[i] = make_float2((float)t1[i].x, (float)t1[i].y); xcf[i] = make_float2((float)xc[i].x, (float)xc[i].y); xs[i] = sin(0.01 * i) / 64.0; }
    for (int i = 0; i < 256; i++) t2f[i] = make_float2((float)t2[i].x, (float)t2[i].y);
    float2 *f1, *f2_, *fx; double* dxs; CK(hipMalloc(&f1, 4096 * 8)); CK(hipMalloc(&f2_, 256 * 8)); CK(hipMalloc(&fx, 4096 * 8)); CK(hipMalloc(&dxs, 4096 * 8));
    CK(hipMemcpy(f1, t1f.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(f2_, t2f.data(), 256 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(fx, xcf.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dxs, xs.data(), 4096 * 8, hipMemcpyHostToDevice));
    p.tw1f = f1; p.tw2f = f2_; p.xcf = fx; p.xs = dxs; p.screen_delta = 1e-4;
    CK(hipMalloc(&p.ovf_count, 4)); CK(hipMalloc(&p.ovf_list, p.npairs * 8));
    go_screen(p, 256 * 3);
    go_screen(p, 256 * 1);
    go<3>(p, 256 * 3);
    return 0;
}
