// upload_probe.cpp -- what bounds the cold upload of a 19.2 MB Group (5 000 Series of 480 samples, separate heap blocks):
// (1) packing the rows into pinned memory with 1 .. 8 host threads, (2) one H2D copy of the packed block from pinned memory,
// (3) the same in pieces of 256 KB .. 4 MB on one stream, (4) pack and piecewise copy overlapped.  hipcc -O2 -pthread.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a) { return std::chrono::duration<double, std::micro>(Clock::now() - a).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const int M = 5000, N = 480;
    const size_t row = N * sizeof(double), total = (size_t)M * row;
    std::vector<std::vector<double>> series(M, std::vector<double>(N, 1.0));
    double *pin = nullptr, *dev = nullptr;
    CK(hipHostMalloc((void **)&pin, total, hipHostMallocDefault));
    CK(hipMalloc(&dev, total));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int T : {1, 2, 4, 8, 16}) {
        double best = 1e30;
        for (int rep = 0; rep < 20; rep++) {
            auto t0 = Clock::now();
            std::vector<std::thread> th;
            for (int w = 0; w < T; w++)
                th.emplace_back([&, w] { for (int r = w * M / T; r < (w + 1) * M / T; r++) memcpy(pin + (size_t)r * N, series[r].data(), row); });
            for (auto &t : th) t.join();
            best = std::min(best, us(t0));
        }
        printf("pack %2d threads (spawned per run): %7.1f us  (%.1f GB/s)\n", T, best, total / best / 1e3);
    }
    for (size_t piece : {total, (size_t)4 << 20, (size_t)1 << 20, (size_t)256 << 10}) {
        double best = 1e30;
        for (int rep = 0; rep < 20; rep++) {
            auto t0 = Clock::now();
            for (size_t o = 0; o < total; o += piece)
                CK(hipMemcpyAsync((char *)dev + o, (char *)pin + o, std::min(piece, total - o), hipMemcpyHostToDevice, st));
            double enq = us(t0);
            CK(hipStreamSynchronize(st));
            double t = us(t0);
            if (t < best) { best = t; (void)enq; }
        }
        printf("H2D from pinned in pieces of %8zu B: %7.1f us  (%.1f GB/s)\n", piece, best, total / best / 1e3);
    }
    { // pageable source, one call (what muse_group_append's slab path does)
        std::vector<double> slab((size_t)M * N, 2.0);
        double best = 1e30;
        for (int rep = 0; rep < 20; rep++) {
            auto t0 = Clock::now();
            CK(hipMemcpyAsync(dev, slab.data(), total, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            best = std::min(best, us(t0));
        }
        printf("H2D from pageable, one call: %7.1f us  (%.1f GB/s)\n", best, total / best / 1e3);
    }
    // a kernel reading pinned host memory directly (zero copy) into HBM
    return 0;
}
