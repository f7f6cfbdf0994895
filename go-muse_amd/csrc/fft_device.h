// fft_device.h -- device-side building blocks shared by the fused xcorr kernels
// (complex helpers, wave/block reductions, z-normalisation constants, the
// in-register 16-point DFT).  gfx950 only; wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace muse {

// ------------------------------------------------------------ small helpers
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sums of K doubles for a 256-thread block (4 waves).  Every
// thread returns the same bits (fixed summation order).  `scratch` must hold
// 4*K doubles that no other reduction is using concurrently.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = wave_sum(v[k]);
        if (lane == 0)
            scratch[wave * K + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++)
        v[k] = (scratch[0 * K + k] + scratch[1 * K + k]) + (scratch[2 * K + k] + scratch[3 * K + k]);
}

// z-normalisation constants from block totals (xcorr.go:84-95 with the
// centred second pass of gonum stat.StdDev): returns 1/sigma, sets flags.
struct ZnFlags {
    bool zero; // sigma == 0  -> (nil,0,0)
    bool nan;  // sigma is NaN -> every cc is NaN -> (lag 0, mv NaN)
};
__device__ __forceinline__ double zn_scale(double s1, double s2, int N, ZnFlags &f)
{
    const double n = (double)N;
    double var = (s2 - s1 * s1 / n) / (double)(N - 1);
    double sd = sqrt(var);
    f.zero = (sd == 0.0);
    f.nan = (sd != sd);
    return (f.zero || f.nan) ? 0.0 : 1.0 / sd;
}

// Two series share one complex transform (z = yA + i yB) and 1/sigma is applied to the winning
// value only, so the transform's rounding noise is relative to the LARGER series: a pair whose
// sigmas differ by 2^k loses k bits on the smaller one.  Kernels that know the statistics before
// the transform bring both series to O(1) with an EXACT power of two close to 1/sigma; kernels that
// defer the statistics list pairs beyond SIGMA_EXP_SPREAD for the kernel that rescales.
__device__ __forceinline__ int var_exp(double var) { return (int)((__double_as_longlong(var) >> 52) & 0x7ff) - 1023; }
__device__ __forceinline__ double pow2_inv_sigma(double var) // 2^-(floor(log2 var) / 2), var finite and > 0
{
    return __longlong_as_double((long long)(1023 - (var_exp(var) >> 1)) << 52);
}
constexpr int SIGMA_EXP_SPREAD = 32; // exponent spread of the VARIANCES tolerated without rescaling: sigma ratio
                                     // 2^16, i.e. at most 16 of the 53 bits lost on the smaller series (1e-11 relative)
__device__ __forceinline__ bool sigma_spread_too_wide(double varA, double varB)
{
    const int d = var_exp(varA) - var_exp(varB);
    return varA > 0.0 && varB > 0.0 && (d > SIGMA_EXP_SPREAD || d < -SIGMA_EXP_SPREAD);
}

// ===================================================== tuned n = 4096 kernel
// 16-point DFT in registers: two radix-4 layers.  Input x[a] at v[a]; output
// X[k] at v[P16(k)], P16(k) = 4*(k&3) + (k>>2) (an involution).
#define P16(k) ((((k)&3) << 2) | ((k) >> 2))

__device__ __forceinline__ void radix4(double2 &a, double2 &b, double2 &c, double2 &d)
{
    double2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = make_double2(t1.x + t3.y, t1.y - t3.x); // t1 - i*t3
    d = make_double2(t1.x - t3.y, t1.y + t3.x); // t1 + i*t3
}

__device__ __forceinline__ void dft16(double2 (&v)[16])
{
    constexpr double C1 = 0.92387953251128675613; // cos(pi/8)
    constexpr double S1 = 0.38268343236508977173; // sin(pi/8)
    constexpr double H = 0.70710678118654752440;  // sqrt(1/2)
    // layer 1: over a1 (stride 4): v[a0 + 4*k1] = y[a0][k1]
#pragma unroll
    for (int a0 = 0; a0 < 4; a0++)
        radix4(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12]);
    // internal twiddles W16^(a0*k1)
    double2 u;
    // a0 = 1: k1 = 1,2,3 -> W1, W2, W3
    u = v[1 + 4];  v[1 + 4]  = make_double2(u.x * C1 + u.y * S1, u.y * C1 - u.x * S1);
    u = v[1 + 8];  v[1 + 8]  = make_double2((u.x + u.y) * H, (u.y - u.x) * H);
    u = v[1 + 12]; v[1 + 12] = make_double2(u.x * S1 + u.y * C1, u.y * S1 - u.x * C1);
    // a0 = 2: W2, W4, W6
    u = v[2 + 4];  v[2 + 4]  = make_double2((u.x + u.y) * H, (u.y - u.x) * H);
    u = v[2 + 8];  v[2 + 8]  = make_double2(u.y, -u.x);
    u = v[2 + 12]; v[2 + 12] = make_double2((u.y - u.x) * H, -(u.x + u.y) * H);
    // a0 = 3: W3, W6, W9
    u = v[3 + 4];  v[3 + 4]  = make_double2(u.x * S1 + u.y * C1, u.y * S1 - u.x * C1);
    u = v[3 + 8];  v[3 + 8]  = make_double2((u.y - u.x) * H, -(u.x + u.y) * H);
    u = v[3 + 12]; v[3 + 12] = make_double2(-u.x * C1 - u.y * S1, u.x * S1 - u.y * C1);
    // layer 2: over a0: v[4*k1 + k0] = X[4*k0 + k1]
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
        radix4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}


// ------------------------------------------------------- DPP wave reductions
// All-lanes reductions without LDS traffic: xor-1, xor-2 (quad_perm), then
// row_half_mirror and row_mirror complete each 16-lane row; the four row
// results are combined through v_readlane.  Every lane returns the same bits.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (old = 0 with bound_ctrl and full masks: every lane is written, so the destination is not tied to the source and
    // the compiler needs no copy of it -- with old = src each 64-bit exchange cost two extra v_mov)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v)
{
    v += dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v); // row_half_mirror
    v += dpp_f64<0x140>(v); // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max_dpp(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
// Wave maximum of NON-NEGATIVE doubles: their bit patterns order like the values, so the maximum is an unsigned maximum of
// the high words followed by one of the low words among the lanes that hold the winning high word -- v_max_u32 takes the
// DPP operand itself (one instruction per step instead of two moves and a v_max_f64 behind a canonicalising copy), and
// row_bcast:15 / :31 (gfx9 DPP) carry the rows' results to lane 63.  A NaN anywhere wins (its pattern is above Inf's);
// callers only use the result of such a series to flag it.
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true)); // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true)); // row_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1, 3 (0 = identity elsewhere)
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_max_nonneg(double x)
{
    const unsigned hi = (unsigned)__double2hiint(x), lo = (unsigned)__double2loint(x);
    const unsigned mh = wave_max_u32_dpp(hi);
    const unsigned ml = wave_max_u32_dpp(hi == mh ? lo : 0u);
    return __hiloint2double((int)mh, (int)ml);
}
__device__ __forceinline__ int wave_min_i_dpp(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}


// ------------------------------------------------------------- fp32 complex
// A plain struct of two floats on purpose: every operation is a scalar fp32 VALU
// op (2-cycle issue).  With a native 2-vector type (or SLP vectorisation) the
// compiler emits v_pk_*_f32, which issue no faster per flop and cost v_mov
// shuffles to pair registers (measured: 400 v_mov per pair of series).
struct alignas(8) f2 {
    float x, y;
};
__device__ __forceinline__ f2 mk2(float x, float y)
{
    f2 r;
    r.x = x;
    r.y = y;
    return r;
}
__device__ __forceinline__ f2 caddf(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ f2 csubf(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ f2 cmulf(f2 a, f2 w) // a * w
{
    return mk2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

__device__ __forceinline__ void radix4f(f2 &a, f2 &b, f2 &c, f2 &d)
{
    const f2 t0 = caddf(a, c), t1 = csubf(a, c), t2 = caddf(b, d), t3 = csubf(b, d);
    a = caddf(t0, t2);
    c = csubf(t0, t2);
    b = mk2(t1.x + t3.y, t1.y - t3.x); // t1 - i*t3
    d = mk2(t1.x - t3.y, t1.y + t3.x); // t1 + i*t3
}

// 16-point DFT in registers, fp32: same dataflow as dft16 (output X[k] at v[P16(k)])
__device__ __forceinline__ void dft16f(f2 (&v)[16])
{
    constexpr float C1 = 0.92387953251128675613f; // cos(pi/8)
    constexpr float S1 = 0.38268343236508977173f; // sin(pi/8)
    constexpr float H = 0.70710678118654752440f;  // sqrt(1/2)
#pragma unroll
    for (int a0 = 0; a0 < 4; a0++)
        radix4f(v[a0], v[a0 + 4], v[a0 + 8], v[a0 + 12]);
    f2 u;
    u = v[1 + 4];  v[1 + 4]  = mk2(u.x * C1 + u.y * S1, u.y * C1 - u.x * S1);
    u = v[1 + 8];  v[1 + 8]  = mk2((u.x + u.y) * H, (u.y - u.x) * H);
    u = v[1 + 12]; v[1 + 12] = mk2(u.x * S1 + u.y * C1, u.y * S1 - u.x * C1);
    u = v[2 + 4];  v[2 + 4]  = mk2((u.x + u.y) * H, (u.y - u.x) * H);
    u = v[2 + 8];  v[2 + 8]  = mk2(u.y, -u.x);
    u = v[2 + 12]; v[2 + 12] = mk2((u.y - u.x) * H, -(u.x + u.y) * H);
    u = v[3 + 4];  v[3 + 4]  = mk2(u.x * S1 + u.y * C1, u.y * S1 - u.x * C1);
    u = v[3 + 8];  v[3 + 8]  = mk2((u.y - u.x) * H, -(u.x + u.y) * H);
    u = v[3 + 12]; v[3 + 12] = mk2(-u.x * C1 - u.y * S1, u.x * S1 - u.y * C1);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
        radix4f(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

// fp32 DPP all-lanes max of non-negative values
__device__ __forceinline__ float wave_max_f32_dpp(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}


// In-kernel phase stamps (diagnostic builds only: TIMING = false in every
// shipped instantiation, where all of this compiles to nothing).
constexpr int NPHASE = 16;
template <bool TIMING>
struct PhaseClock {
    unsigned long long acc[NPHASE];
    unsigned long long last;
    __device__ __forceinline__ void start()
    {
        if (TIMING) {
#pragma unroll
            for (int i = 0; i < NPHASE; i++)
                acc[i] = 0;
            last = __builtin_amdgcn_s_memtime();
        }
    }
    template <int I>
    __device__ __forceinline__ void stamp()
    {
        if (TIMING) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            acc[I] += now - last;
            last = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};


} // namespace muse
