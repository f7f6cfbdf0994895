// two_device.h -- the batched two-sided xCorr's per-pair statistics (xcorr.go:108-128, 139-143), shared by
// xcorr_two_sided.hip (n = 4096, n >= 32768) and xcorr_small.hip (n = 512 ... 2048, 8192, 16384).
#pragma once
#include "r16_device.h"

namespace muse {
namespace two {

using namespace occ4;

// what the statistics of one pair decide (uniform over the pair's threads)
struct PairScale {
    double sA, sB, mA, mB; // v = x * s - m at valid positions
    double fac;            // cc = transform output * fac
    bool nil, nan;
};
// the launch's reciprocals (uniform; formed on the host where the kernel wants them in scalar registers)
struct PairInv {
    double invNx, invNxm1, invNy, invNym1, invnm1;
};
__host__ __device__ __forceinline__ PairInv pair_inv(const int Nx, const int Ny, const int n)
{
    return PairInv{1.0 / (double)Nx, 1.0 / (double)(Nx - 1), 1.0 / (double)Ny, 1.0 / (double)(Ny - 1), 1.0 / (double)(n - 1)};
}
// q = {sum dx, sum dx^2, sum dy, sum dy^2} with d = sample - first sample (normalized) or the sample itself (raw)
__device__ __forceinline__ PairScale pair_scale(const double (&q)[4], const PairInv &iv, const bool normalize)
{
    PairScale s;
    if (normalize) {
        bool zA, nA, zB, nB;
        const double invNx = iv.invNx, invNy = iv.invNy;
        const double vA0 = variance(Stat{q[0], q[1]}, invNx, iv.invNxm1, zA, nA);
        const double vB0 = variance(Stat{q[2], q[3]}, invNy, iv.invNym1, zB, nB);
        s.nil = zA || zB; // xcorr.go:110-127: either sigma == 0 -> (nil, 0, 0) (x is checked first; a NaN x with a constant y is nil too)
        s.nan = !s.nil && (nA || nB);
        const bool dead = s.nil || s.nan;
        s.sA = dead ? 1.0 : pow2_inv_sigma(vA0);
        s.sB = dead ? 1.0 : pow2_inv_sigma(vB0);
        s.mA = q[0] * invNx * s.sA;
        s.mB = q[2] * invNy * s.sB;
        const double vA = vA0 * s.sA * s.sA, vB = vB0 * s.sB * s.sB;
        double ya = __builtin_amdgcn_rsq(vA), yb = __builtin_amdgcn_rsq(vB);
        ya = ya * fma(-0.5 * vA * ya, ya, 1.5);
        ya = ya * fma(-0.5 * vA * ya, ya, 1.5);
        yb = yb * fma(-0.5 * vB * yb, yb, 1.5);
        yb = yb * fma(-0.5 * vB * yb, yb, 1.5);
        s.fac = ya * yb * iv.invnm1; // xcorr.go:140: 1 / (n (n - 1)); the 1 / n rides in the untangled spectrum
    } else {
        const double eA = q[1] * iv.invNx, eB = q[3] * iv.invNy; // mean squares (only their binary exponents are used)
        s.nil = false;
        // A mean square that is not a positive number gives no scale: Inf / NaN (a NaN or Inf sample, or finite samples whose
        // squares overflow), or 0 (an all-zero series, or samples whose squares underflow) -- beside a series of any magnitude the
        // cross term of Z^2 would drown in the rounding of the other series' square.  Such a pair leaves as a NaN pair; the
        // host has every NaN pair looked at again (capi_xcorr.hip, rescue_overflowed_pairs): all-zero series -> every cc is 0,
        // finite samples -> recomputed on copies balanced by exact powers of two, NaN / Inf samples -> NaN stands.
        s.nan = !__builtin_isfinite(eA) || !__builtin_isfinite(eB) || !(eA > 0.0) || !(eB > 0.0);
        s.sA = s.nan ? 1.0 : pow2_inv_sigma(eA);
        s.sB = s.nan ? 1.0 : pow2_inv_sigma(eB);
        s.mA = s.mB = 0.0;
        s.fac = (1.0 / s.sA) * (1.0 / s.sB); // exact
    }
    return s;
}
__device__ __forceinline__ PairScale pair_scale(const double (&q)[4], const int Nx, const int Ny, const int n, const bool normalize)
{
    return pair_scale(q, pair_inv(Nx, Ny, n), normalize);
}

} // namespace two
} // namespace muse
