// Ablation harness for the SHIPPED pipelined kernel (diagnostic only): includes
// the product source and instantiates it with its AB_* switches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../go-muse_amd/csrc/xcorr_r16_pipe.hip"

using namespace muse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ABL> float run(const FusedParams& p, int grid, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((xcorr_fused_n4096_pipe<false, ABL>), dim3(grid), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; i++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((xcorr_fused_n4096_pipe<false, ABL>), dim3(grid), dim3(256), 0, 0, p);
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
    int grid = argc > 2 ? atoi(argv[2]) : 512;
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
    const int it = 5;
#define R(name, abl) printf("%-52s %8.3f ms\n", name, run<abl>(p, grid, it)); fflush(stdout);
    R("pipe baseline", 0)
    R("no stagger", AB_NOSTAGGER)
    R("-row loads", AB_NOLOAD)
    R("-xc loads", AB_NOXC)
    R("-tw1 loads", AB_NOTW1)
    R("-tw2 LDS reads", AB_NOTW2)
    R("-xc -tw1 -tw2", AB_NOXC | AB_NOTW1 | AB_NOTW2)
    R("-znorm reduction", AB_NOZN)
    R("-argmax reduction", AB_NOARG)
    R("-znorm -argmax", AB_NOZN | AB_NOARG)
    R("-LDS exchange (barriers kept)", AB_NOXCHG)
    R("-FFT barriers", AB_NOBAR)
    R("-exchange -barriers", AB_NOXCHG | AB_NOBAR)
    R("-dft16 math", AB_NODFT)
    R("-all global (loads, xc, tw1)", AB_NOLOAD | AB_NOXC | AB_NOTW1)
    R("-all global -zn -arg", AB_NOLOAD | AB_NOXC | AB_NOTW1 | AB_NOZN | AB_NOARG)
    R("VALU only", AB_NOLOAD | AB_NOXC | AB_NOTW1 | AB_NOTW2 | AB_NOZN | AB_NOARG | AB_NOXCHG | AB_NOBAR)
    R("loads only (no dft, tw, xc, xchg, bar, zn, arg)", AB_NODFT | AB_NOXC | AB_NOTW1 | AB_NOTW2 | AB_NOXCHG | AB_NOBAR | AB_NOZN | AB_NOARG)
    R("loads + zn + arg", AB_NODFT | AB_NOXC | AB_NOTW1 | AB_NOTW2 | AB_NOXCHG | AB_NOBAR)
    return 0;
}
