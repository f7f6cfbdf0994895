// fold_device.h -- 16-point transforms with every twiddle multiplication folded into the
// butterflies' fused multiply-adds (xcorr_r16_fold.hip).  gfx950 only.
//
// Why: the fp64 kernels are VALU-bound (profiles/r01_fast_rocprof_summary.txt: 1 540 VALU
// instructions per wave per pair, v_*_f64 issue at 4 cycles).  A radix-16 pass written as
// "DFT16, then 15 complex twiddle multiplies" costs 160 + 60 = 220 instructions; the same pass
// as a GENERALISED DFT
//        X[m] = sum_b x[b] W_16^(b (m + delta)),     delta = the pass's per-thread twiddle phase
// in four radix-2 stages costs 32 butterflies x 6 FMAs = 192, because
//        (a, b) -> (a + w b, a - w b)   =   o1 = a + w b (4 FMAs),  o2 = 2 a - o1 (2 FMAs)
// needs no separate complex multiply.  (All radix-2/4/split-radix FMA factorizations of a
// 16-point transform with general twiddles cost 192: Linzer & Feig's bound of 6 per butterfly.)
// The plain transform (delta = 0) costs 148 instead of 160 with the same butterflies.
//
// Twiddles of stage L in {2, 4, 8, 16}: exp(-2 pi i (m + delta) / L), m < L/2; all of them are
// one of EIGHT per-thread constants times 1 or -i:
//     G[0] = w2, G[1] = w4, G[2] = w8, G[3] = w8 W_8, G[4 + q] = w16 W_16^q (q < 4),  wL = exp(-2 pi i delta / L)
// so a pass loads 8 complex factors per thread instead of 15.
//
// Two in-place register orders (BR = 4-bit reversal):
//   NR: natural in (x[b] at v[b]) -> bit-reversed out (X[m] at v[BR16(m)]); strides 8, 4, 2, 1
//   RN: bit-reversed in (x[b] at v[BR16(b)]) -> natural out (X[m] at v[m]);   strides 1, 2, 4, 8
#pragma once
#include <hip/hip_runtime.h>

#include "fft_device.h"

namespace muse {
namespace fold {

#define BR16(k) (((((k)&1) << 3) | (((k)&2) << 1) | (((k)&4) >> 1) | (((k)&8) >> 3)))

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// (a, b) <- (a + w b, a - w b)
__device__ __forceinline__ void bf_gen(double2 &a, double2 &b, const double2 w)
{
    const double px = fma_(w.x, b.x, a.x), py = fma_(w.x, b.y, a.y);
    const double ox = fma_(-w.y, b.y, px), oy = fma_(w.y, b.x, py);
    b = make_double2(fma_(2.0, a.x, -ox), fma_(2.0, a.y, -oy));
    a = make_double2(ox, oy);
}
// (a, b) <- (a + (-i w) b, a - (-i w) b):  -i (w b) = ((w b).y, -(w b).x)
__device__ __forceinline__ void bf_gen_mi(double2 &a, double2 &b, const double2 w)
{
    const double px = fma_(w.x, b.y, a.x), py = fma_(-w.x, b.x, a.y);
    const double ox = fma_(w.y, b.x, px), oy = fma_(w.y, b.y, py);
    b = make_double2(fma_(2.0, a.x, -ox), fma_(2.0, a.y, -oy));
    a = make_double2(ox, oy);
}
// w = 1
__device__ __forceinline__ void bf_one(double2 &a, double2 &b)
{
    const double2 o = make_double2(a.x + b.x, a.y + b.y);
    b = make_double2(a.x - b.x, a.y - b.y);
    a = o;
}
// w = -i
__device__ __forceinline__ void bf_mi(double2 &a, double2 &b)
{
    const double2 o = make_double2(a.x + b.y, a.y - b.x);
    b = make_double2(a.x - b.y, a.y + b.x);
    a = o;
}
// w = W_8 = (1 - i) / sqrt 2:  W_8 b = H (b.x + b.y, b.y - b.x)
__device__ __forceinline__ void bf_w8(double2 &a, double2 &b)
{
    constexpr double H = 0.70710678118654752440;
    const double tx = b.x + b.y, ty = b.y - b.x;
    const double2 o = make_double2(fma_(H, tx, a.x), fma_(H, ty, a.y));
    b = make_double2(fma_(-H, tx, a.x), fma_(-H, ty, a.y));
    a = o;
}
// w = -i W_8 = W_8^3 = (-1 - i) / sqrt 2:  W_8^3 b = H (b.y - b.x, -(b.x + b.y))
__device__ __forceinline__ void bf_w8_mi(double2 &a, double2 &b)
{
    constexpr double H = 0.70710678118654752440;
    const double tx = b.y - b.x, ty = b.x + b.y;
    const double2 o = make_double2(fma_(H, tx, a.x), fma_(-H, ty, a.y));
    b = make_double2(fma_(-H, tx, a.x), fma_(H, ty, a.y));
    a = o;
}

// ---- generalised 16-point transform, natural in -> bit-reversed out; g[0..7] as in the header
__device__ __forceinline__ void gdft16_nr_s12(double2 (&v)[16], const double2 g0, const double2 g1)
{
#pragma unroll
    for (int j = 0; j < 8; j++) // stage L = 2: (j, j + 8)
        bf_gen(v[j], v[j + 8], g0);
#pragma unroll
    for (int j = 0; j < 4; j++) { // stage L = 4: m1 = 0: w4, m1 = 1: -i w4
        bf_gen(v[j], v[j + 4], g1);
        bf_gen_mi(v[j + 8], v[j + 12], g1);
    }
}
__device__ __forceinline__ void gdft16_nr_s3(double2 (&v)[16], const double2 g2, const double2 g3)
{
#pragma unroll
    for (int j = 0; j < 2; j++) { // stage L = 8: position j + 8 m1 + 4 m2 (+2), m' = m1 + 2 m2: w8 {1, W8, -i, -i W8}
        bf_gen(v[j], v[j + 2], g2);             // m' = 0
        bf_gen(v[j + 8], v[j + 10], g3);        // m' = 1
        bf_gen_mi(v[j + 4], v[j + 6], g2);      // m' = 2
        bf_gen_mi(v[j + 12], v[j + 14], g3);    // m' = 3
    }
}
__device__ __forceinline__ void gdft16_nr_s4(double2 (&v)[16], const double2 g4, const double2 g5, const double2 g6,
                                             const double2 g7)
{
    // stage L = 16: position 8 m1 + 4 m2 + 2 m3 (+1), m'' = m1 + 2 m2 + 4 m3: w16 W16^m'', m'' >= 4: -i times m'' - 4
    bf_gen(v[0], v[1], g4);       // m'' = 0
    bf_gen(v[8], v[9], g5);       // 1
    bf_gen(v[4], v[5], g6);       // 2
    bf_gen(v[12], v[13], g7);     // 3
    bf_gen_mi(v[2], v[3], g4);    // 4
    bf_gen_mi(v[10], v[11], g5);  // 5
    bf_gen_mi(v[6], v[7], g6);    // 6
    bf_gen_mi(v[14], v[15], g7);  // 7
}

// fetch(s), s = 0..7 returns g[s]; two batches of four in flight at most (32 VGPRs)
template <typename F>
__device__ __forceinline__ void gdft16_nr(double2 (&v)[16], F fetch)
{
    double2 ga[4], gb[4];
#pragma unroll
    for (int s = 0; s < 4; s++)
        ga[s] = fetch(s);
    __builtin_amdgcn_sched_barrier(0);
    gdft16_nr_s12(v, ga[0], ga[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; s++)
        gb[s] = fetch(4 + s);
    __builtin_amdgcn_sched_barrier(0);
    gdft16_nr_s3(v, ga[2], ga[3]);
    gdft16_nr_s4(v, gb[0], gb[1], gb[2], gb[3]);
}

// ---- plain 16-point DFT (delta = 0), natural in -> bit-reversed out: 32 + 32 + 40 + 44 = 148 instructions
__device__ __forceinline__ void bf_c(double2 &a, double2 &b, const double c, const double s) // w = c - i s
{
    bf_gen(a, b, make_double2(c, -s));
}
__device__ __forceinline__ void bf_c_mi(double2 &a, double2 &b, const double c, const double s) // w = -i (c - i s)
{
    bf_gen_mi(a, b, make_double2(c, -s));
}
constexpr double C16_1 = 0.92387953251128675613; // cos(pi/8)
constexpr double S16_1 = 0.38268343236508977173; // sin(pi/8)

__device__ __forceinline__ void dft16_nr(double2 (&v)[16])
{
#pragma unroll
    for (int j = 0; j < 8; j++)
        bf_one(v[j], v[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        bf_one(v[j], v[j + 4]);
        bf_mi(v[j + 8], v[j + 12]);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        bf_one(v[j], v[j + 2]);
        bf_w8(v[j + 8], v[j + 10]);
        bf_mi(v[j + 4], v[j + 6]);
        bf_w8_mi(v[j + 12], v[j + 14]);
    }
    bf_one(v[0], v[1]);
    bf_c(v[8], v[9], C16_1, S16_1);        // W16^1
    bf_w8(v[4], v[5]);                     // W16^2
    bf_c(v[12], v[13], S16_1, C16_1);      // W16^3 = cos(3pi/8) - i sin(3pi/8)
    bf_mi(v[2], v[3]);                     // W16^4
    bf_c_mi(v[10], v[11], C16_1, S16_1);   // W16^5 = -i W16^1
    bf_w8_mi(v[6], v[7]);                  // W16^6
    bf_c_mi(v[14], v[15], S16_1, C16_1);   // W16^7 = -i W16^3
}

// ---- plain 16-point DFT, bit-reversed in -> natural out, with the input multiplied by a per-element
// factor folded into the first stage: x[b] = z[b] * xc[b], z[b] at v[BR16(b)], xc[b] = fetch(b).
//   stage 1 pairs b and b + 8 = registers (r, r + 1), r = BR16(b) even:  p = z_b xc_b (4),
//   o1 = p + z_(b+8) xc_(b+8) (4 FMAs), o2 = 2 p - o1 (2): 10 per butterfly instead of 4 + 4 + 4.
__device__ __forceinline__ void bf_xc(double2 &a, double2 &b, const double2 xa, const double2 xb)
{
    const double px = fma_(-a.y, xa.y, a.x * xa.x), py = fma_(a.y, xa.x, a.x * xa.y);
    const double ox = fma_(b.x, xb.x, fma_(-b.y, xb.y, px)), oy = fma_(b.x, xb.y, fma_(b.y, xb.x, py));
    b = make_double2(fma_(2.0, px, -ox), fma_(2.0, py, -oy));
    a = make_double2(ox, oy);
}
// the same first stage on the element-wise SQUARE of the input (two-sided xCorr, xcorr_two_sided.hip):
//   p = z_b^2 (4), o1 = p + z_(b+8)^2 (4), o2 = 2 p - o1 (2)
__device__ __forceinline__ void bf_sq(double2 &a, double2 &b)
{
    const double px = fma_(-a.y, a.y, a.x * a.x), py = (a.x + a.x) * a.y;
    const double ox = fma_(b.x, b.x, fma_(-b.y, b.y, px)), oy = fma_(b.x + b.x, b.y, py);
    b = make_double2(fma_(2.0, px, -ox), fma_(2.0, py, -oy));
    a = make_double2(ox, oy);
}
// stages 2..4 of the RN order: strides 2, 4, 8; the stage-L butterfly at sub-index m uses W_L^m
__device__ __forceinline__ void dft16_rn_s234(double2 (&v)[16])
{
#pragma unroll
    for (int q = 0; q < 4; q++) { // stage 2: positions 4 q + m1 (+2)
        bf_one(v[4 * q], v[4 * q + 2]);
        bf_mi(v[4 * q + 1], v[4 * q + 3]);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) { // stage 3: positions 8 q + m' (+4), m' = 0..3: 1, W8, -i, -i W8
        bf_one(v[8 * q], v[8 * q + 4]);
        bf_w8(v[8 * q + 1], v[8 * q + 5]);
        bf_mi(v[8 * q + 2], v[8 * q + 6]);
        bf_w8_mi(v[8 * q + 3], v[8 * q + 7]);
    }
    bf_one(v[0], v[8]); // stage 4: positions m'' (+8), W16^m''
    bf_c(v[1], v[9], C16_1, S16_1);
    bf_w8(v[2], v[10]);
    bf_c(v[3], v[11], S16_1, C16_1);
    bf_mi(v[4], v[12]);
    bf_c_mi(v[5], v[13], C16_1, S16_1);
    bf_w8_mi(v[6], v[14]);
    bf_c_mi(v[7], v[15], S16_1, C16_1);
}

} // namespace fold
} // namespace muse
