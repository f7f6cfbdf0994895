// diag_kernels.hip -- measurement hook (include/muse_hip_test.h), not on the product path: ONE wave that samples the shader
// clock while other kernels run.  In-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md,
// "DVFS give-back" item 6: s_memtime ticks at the shader clock, s_memrealtime at a constant 100 MHz); bench.py starts the
// probe on its own stream next to the timed launches and prices the fp64-VALU and LDS ceilings of its roofline object at the
// clock the chip actually held under that load, not at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xcorr_kernels.h"
#include "foldk_device.h"

namespace muse {

// Unit-test hook (muse_test_wave_argmax): one 256-thread workgroup holds 2 x 4096 given values exactly as the n = 4096
// kernels hold a pair's correlations behind their last pass -- lag index t + 256 m of series A / B in v[BR16(m)].x / .y of
// thread t -- and runs foldk::wave_argmax_store on them: out[6 w ... 6 w + 5] = wave w's {max |cc|, signed value, index}
// of A, then of B.  The tie rules (lowest index among equal |values|, across registers, lanes and waves) are what the
// parity test feeds it.
__global__ __launch_bounds__(256) void wave_argmax_probe_kernel(const double *ccA, const double *ccB, double *out)
{
    using namespace fold;
    __shared__ double rec[24];
    const int t = threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int m = 0; m < 16; m++)
        v[BR16(m)] = make_double2(ccA[t + 256 * m], ccB[t + 256 * m]);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    foldk::wave_argmax_store(v, wave, t & 63, rec + 6 * wave);
    __syncthreads();
    if (t < 24)
        out[t] = rec[t];
}

hipError_t launch_wave_argmax_probe(const double *ccA, const double *ccB, double *out24, hipStream_t stream)
{
    hipLaunchKernelGGL(wave_argmax_probe_kernel, dim3(1), dim3(256), 0, stream, ccA, ccB, out24);
    return hipGetLastError();
}

// windows of ~window_ticks of the 100 MHz clock until total_ticks have passed (or the buffer is full): out[2 w] = shader
// ticks, out[2 w + 1] = 100 MHz ticks of window w; *count = windows written.  One wave, a handful of registers: it fits
// next to the resident grids of the product kernels and sleeps between samples.
// count[1] is a stop flag the host may set (pinned memory): the probe then ends within one sleep (~4 us), so that a
// device-wide synchronisation behind the measured launches does not wait out the rest of total_ticks.
__global__ __launch_bounds__(64, 1) void clock_probe_kernel(unsigned long long *out, int *count, int max_windows,
                                                            unsigned long long window_ticks, unsigned long long total_ticks)
{
    if (threadIdx.x != 0)
        return;
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();
    const volatile int *stop = count + 1;
    int w = 0;
    for (; w < max_windows && !*stop; w++) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long r1 = r0;
        while (r1 - r0 < window_ticks && !*stop) {
            __builtin_amdgcn_s_sleep(127);
            r1 = __builtin_amdgcn_s_memrealtime();
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
        __threadfence_system();
        *(volatile int *)count = w + 1; // (the host polls this to know the probe is resident before it starts its launches)
        if (r1 - r_begin >= total_ticks) {
            w++;
            break;
        }
    }
    __threadfence_system();
    *(volatile int *)count = w;
}

hipError_t launch_clock_probe(unsigned long long *out, int *count, int max_windows, double window_ms, double total_ms,
                              hipStream_t stream)
{
    const unsigned long long wt = (unsigned long long)(window_ms * 1e5), tt = (unsigned long long)(total_ms * 1e5);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out, count, max_windows, wt, tt);
    return hipGetLastError();
}

} // namespace muse
