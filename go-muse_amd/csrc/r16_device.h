// r16_device.h -- device helpers shared by the tuned n = 4096 kernels
// (xcorr_r16_occ4.hip, xcorr_r16_fast.hip): LDS-only barrier, scalar global
// pointers, the half-round LDS transposes, twiddle fetchers and the register
// prefetch of the next pair's rows.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_device.h"
#include "xcorr_kernels.h"

namespace muse {

constexpr int OCC_THREADS = 256;
constexpr int OCC_XBUF = 8 * 272; // double2 elements: 34,816 B

namespace occ4 {

__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }
// LDS-only barrier (does not drain outstanding global loads)
__device__ __forceinline__ void lds_barrier()
{
    fence();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    fence();
}
// a wave-uniform double moved to SGPRs (frees two VGPRs per value)
__device__ __forceinline__ double uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// The result is an address_space(1) (global) pointer on purpose: laundering a
// generic pointer through the asm loses the address space and every load through
// it becomes flat_load, which counts on lgkmcnt as well -- the LDS-only barrier
// (s_waitcnt lgkmcnt(0)) would then drain the prefetch.
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
using gptr = const T __attribute__((address_space(1))) *;
#else
template <typename T>
using gptr = const T *; // host pass only parses this file
#endif
typedef double d2v __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ gptr<T> scalar_ptr(const T *p)
{
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+s"(u));
    return (gptr<T>)u;
}
// p + off as a scalar computed where it is used: the base is made opaque first, so
// the s_add cannot be hoisted out of the pair loop (hoisted bases get spilled to VGPR
// lanes and cost a v_readlane per use)
template <typename T>
__device__ __forceinline__ gptr<T> scalar_ptr_at(const T *p, long long off)
{
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+s"(u));
    u += (unsigned long long)(off * (long long)sizeof(T));
    asm volatile("" : "+s"(u));
    return (gptr<T>)u;
}
// a wave-uniform pointer fetched from a device table, forced into SGPRs (the compiler cannot prove uniformity)
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *ptr)
{
    const unsigned long long u = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

// 16-byte LDS accesses as ONE native vector access: a double2 struct copy is scalarised and re-paired by the compiler, which
// may then pair neighbours across element boundaries (ds_read2_b64 at offset 8: xcorr_two_sided_fold's second workgroup-wide
// transpose did -- SQ_LDS_BANK_CONFLICT 10 % of its LDS cycles on a layout that is conflict-free for ds_read_b128)
__device__ __forceinline__ double2 lds_ld2(const double2 *p)
{
    const d2v x = *reinterpret_cast<const d2v *>(p);
    return make_double2(x.x, x.y);
}
__device__ __forceinline__ void lds_st2(double2 *p, const double2 v)
{
    *reinterpret_cast<d2v *>(p) = d2v{v.x, v.y};
}
// 16-byte global load of one complex value (native vector type: HIP's double2
// struct cannot be copied out of an address_space(1) reference)
__device__ __forceinline__ double2 ldg2(gptr<double2> p, int i)
{
    const d2v x = ((gptr<d2v>)p)[i];
    return make_double2(x.x, x.y);
}
// the same with an UNSIGNED 32-bit lane offset (scalar base + voffset, no 64-bit address arithmetic)
__device__ __forceinline__ double2 ldg2u(gptr<double2> p, unsigned i)
{
    const d2v x = ((gptr<d2v>)p)[i];
    return make_double2(x.x, x.y);
}

// One LDS transpose in two half rounds through the 8 x 272 buffer (positions in
// double2 units).  Layouts (same bank analysis as xcorr_kernels.hip):
//   A: writer (b = hi, c = lo) output k1 -> 272*(k1&7) + t
//      reader (k1 = hi, c = lo) input b  <- 272*(hi&7) + 16*b + lo
//   B: writer (k1 = hi, c = lo) output k2 -> 272*(k2&7) + 17*hi + lo
//      reader (k1 = lo, k2 = hi) input c <- 272*(hi&7) + 17*lo + c
// Round 0 moves outputs 0..7 (read by waves 0-1, whose hi is 0..7), round 1
// outputs 8..15 (waves 2-3).  `wave` is an SGPR, so the two paths are scalar
// branches with disjoint live ranges; both execute the same four barriers.
template <bool B>
__device__ __forceinline__ void exchange(double2 (&v)[16], double2 *xbuf, const int wave, const int t)
{
    const int hi = t >> 4, lo = t & 15;
    const int wbase = B ? 17 * hi + lo : t;
    const int rbase = 272 * (hi & 7) + (B ? 17 * lo : lo);
    lds_barrier(); // buffer free: the previous transpose's last readers are done
#pragma unroll
    for (int k = 0; k < 8; k++)
        xbuf[272 * k + wbase] = v[P16(k)];
    lds_barrier();
    if (wave < 2) {
        double2 w[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            w[e] = xbuf[rbase + (B ? e : 16 * e)];
        lds_barrier();
#pragma unroll
        for (int k = 8; k < 16; k++)
            xbuf[272 * (k - 8) + wbase] = v[P16(k)];
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = w[e];
    } else {
        lds_barrier();
#pragma unroll
        for (int k = 8; k < 16; k++)
            xbuf[272 * (k - 8) + wbase] = v[P16(k)];
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = xbuf[rbase + (B ? e : 16 * e)];
    }
}

// 16-point DFT followed by 15 twiddle multiplies whose factors are fetched by
// `fetch(k)` (k = 1..15) in two batches; the first batch is issued BEFORE the
// butterflies and the second before the first is consumed, so the fetch latency
// (L2 or LDS) overlaps arithmetic instead of adding to the dependent chain.
template <typename F>
__device__ __forceinline__ void dft16_twiddle(double2 (&v)[16], F fetch)
{
    double2 ta[8], tb[7];
#pragma unroll
    for (int j = 0; j < 8; j++)
        ta[j] = fetch(1 + j);
    fence();
    dft16(v);
    fence();
#pragma unroll
    for (int j = 0; j < 7; j++)
        tb[j] = fetch(9 + j);
#pragma unroll
    for (int j = 0; j < 8; j++)
        v[P16(1 + j)] = cmul(v[P16(1 + j)], ta[j]);
    fence();
#pragma unroll
    for (int j = 0; j < 7; j++)
        v[P16(9 + j)] = cmul(v[P16(9 + j)], tb[j]);
}

// The same with four batches of four factors (two in flight: 32 VGPRs instead of 60), for the
// 128-register builds.
template <typename F>
__device__ __forceinline__ void dft16_twiddle_small(double2 (&v)[16], F fetch)
{
    double2 ta[4], tb[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        ta[j] = fetch(1 + j);
    fence();
    dft16(v);
    fence();
#pragma unroll
    for (int j = 0; j < 4; j++)
        tb[j] = fetch(5 + j);
#pragma unroll
    for (int j = 0; j < 4; j++)
        v[P16(1 + j)] = cmul(v[P16(1 + j)], ta[j]);
    fence();
#pragma unroll
    for (int j = 0; j < 4; j++)
        ta[j] = fetch(9 + j);
#pragma unroll
    for (int j = 0; j < 4; j++)
        v[P16(5 + j)] = cmul(v[P16(5 + j)], tb[j]);
    fence();
#pragma unroll
    for (int j = 0; j < 3; j++)
        tb[j] = fetch(13 + j);
#pragma unroll
    for (int j = 0; j < 4; j++)
        v[P16(9 + j)] = cmul(v[P16(9 + j)], ta[j]);
    fence();
#pragma unroll
    for (int j = 0; j < 3; j++)
        v[P16(13 + j)] = cmul(v[P16(13 + j)], tb[j]);
}

// each factor's row base is a scalar (s_add on the table pointer): the load is
// saddr + the shared VGPR offset 16 t, no 64-bit VALU address arithmetic
struct Tw1Fetch {
    const double2 *p;
    int t;
    __device__ __forceinline__ double2 operator()(int k) const
    {   // one scalar base per two rows: the odd row sits at immediate offset -4096 B
        return ldg2(scalar_ptr_at(p, ((k + 1) & ~1) * 256), t - 256 * (k & 1));
    }
};
struct Tw2Fetch {
    const double2 *p;
    int lo;
    __device__ __forceinline__ double2 operator()(int k) const
    {
        return p[k * 16 + lo];
    }
};

// The next pair's rows, prefetched into registers: element t + 256*i of the two
// (zero-padded) rows plus each row's first sample.
struct RawPair {
    double a[16], b[16];
    double ka, kb;
};
// Unconditional coalesced nontemporal loads (a conditional prefetch parks `raw`
// in scratch; a per-element `if` serialises the loads): the caller clamps `pair`.
// F32: the group stores float32 rows (muse_group_create_f32, opt-in: half the HBM bytes); every sample is widened to
// float64 exactly when it is consumed, the arithmetic is the float64 arithmetic of the float64 groups.
template <bool PADDED, bool F32 = false>
__device__ __forceinline__ void issue_row_loads(RawPair &r, const FusedParams &p, long long pair, int t, int pad)
{
    const long long rA = 2 * pair;
    const long long rB = (rA + 1 < p.M) ? rA + 1 : rA;
    // PADDED: the clamped per-lane offsets of the lower half (i < 8) are formed HERE, per call -- left visible to the compiler they
    // are loop-invariant, get hoisted out of the pair loop into sixteen registers, are parked in scratch, and every row request then
    // waits (vmcnt(0): scratch reloads share the counter) for the HBM request in front of it: sixteen serialised round trips per
    // pair.  The upper half (always data: pad < 2048) needs no clamp: one scalar base per request + the shared lane offset.
    if (PADDED)
        asm volatile("" : "+v"(t));
    if (F32) {
        const gptr<float> ra = scalar_ptr(p.rows32 + rA * p.stride);
        const gptr<float> rb = scalar_ptr(p.rows32 + rB * p.stride);
        r.ka = (double)ra[0];
        r.kb = (double)rb[0];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (PADDED && i >= 8) {
                r.a[i] = (double)__builtin_nontemporal_load(scalar_ptr_at(p.rows32 + rA * p.stride, 256 * i - pad) + (unsigned)(t & 255));
                r.b[i] = (double)__builtin_nontemporal_load(scalar_ptr_at(p.rows32 + rB * p.stride, 256 * i - pad) + (unsigned)(t & 255));
            } else if (PADDED) {
                int j = t + 256 * i - pad;
                j = j < 0 ? 0 : j;
                const unsigned ju = (unsigned)j & 4095u;
                r.a[i] = (double)__builtin_nontemporal_load(ra + ju);
                r.b[i] = (double)__builtin_nontemporal_load(rb + ju);
            } else { // one scalar base per four 1 KB slices (immediate offsets -2048 .. +1024 B) + the shared VGPR offset 4 t
                const int c = (i & ~3) * 256 + 512;
                r.a[i] = (double)__builtin_nontemporal_load(scalar_ptr_at(p.rows32 + rA * p.stride, c) + (256 * i - c) + t);
                r.b[i] = (double)__builtin_nontemporal_load(scalar_ptr_at(p.rows32 + rB * p.stride, c) + (256 * i - c) + t);
            }
        }
        return;
    }
    const gptr<double> ra = scalar_ptr(p.rows + rA * p.stride);
    const gptr<double> rb = scalar_ptr(p.rows + rB * p.stride);
    r.ka = ra[0];
    r.kb = rb[0];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (PADDED && i >= 8) { // pad < 2048 (n = nextPowOf2(N)): elements 2048.. are always data
            r.a[i] = __builtin_nontemporal_load(scalar_ptr_at(p.rows + rA * p.stride, 256 * i - pad) + (unsigned)(t & 255));
            r.b[i] = __builtin_nontemporal_load(scalar_ptr_at(p.rows + rB * p.stride, 256 * i - pad) + (unsigned)(t & 255));
        } else if (PADDED) {
            int j = t + 256 * i - pad;
            j = j < 0 ? 0 : j; // clamped: a pad position loads the row's first sample K, so d = x - K = 0 there without a mask
            // an UNSIGNED 12-bit index: saddr + 32-bit voffset addressing, no 64-bit sign extension per load
            const unsigned ju = (unsigned)j & 4095u;
            r.a[i] = __builtin_nontemporal_load(ra + ju);
            r.b[i] = __builtin_nontemporal_load(rb + ju);
        } else { // one scalar base per four 2 KB slices (immediate offsets -4096 .. +2048 B)
                 // + the shared VGPR offset 8 t: no 64-bit VALU address arithmetic
            const int c = (i & ~3) * 256 + 512;
            r.a[i] = __builtin_nontemporal_load(scalar_ptr_at(p.rows + rA * p.stride, c) + (256 * i - c) + t);
            r.b[i] = __builtin_nontemporal_load(scalar_ptr_at(p.rows + rB * p.stride, c) + (256 * i - c) + t);
        }
    }
}

// 16-byte row loads (N == n == 4096, float64 rows): lane (m = lane & 31, h = lane >> 5) of wave w requests, for i' = 0 .. 7, the
// two CONSECUTIVE samples 256 (i' + 8 h) + 64 w + 2 m (+1) of each row -- 16 global_load_dwordx4 per pair instead of 32
// global_load_dwordx2 (the request costs per instruction, profiles/r02_fold_f32_and_tail_experiments.txt); the pair lands in
// (r.a[i'], r.a[i' + 8]).  widen_rows() then trades the upper half-wave's first sample for the lower half-wave's second one
// (v_permlane32_swap, gfx950: 32 VALU per pair), after which lane (m, h) holds COLUMN 64 w + 2 m + h of both rows:
// r.a[i] = A[256 i + col], r.b[i] = B[256 i + col], i = 0 .. 15 -- the layout pass 1 wants, on a relabelled column.
__device__ __forceinline__ int wide_column(int t) { return (t & ~63) | ((t & 31) << 1) | ((t >> 5) & 1); }
__device__ __forceinline__ void issue_row_loads_wide(RawPair &r, const FusedParams &p, long long pair, int t)
{
    const long long rA = 2 * pair;
    const long long rB = (rA + 1 < p.M) ? rA + 1 : rA;
    // the rows' first samples through the scalar cache (uniform address): no vector memory instruction
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const double __attribute__((address_space(4))) *cptr;
    r.ka = *(cptr)(unsigned long long)(p.rows + rA * p.stride);
    r.kb = *(cptr)(unsigned long long)(p.rows + rB * p.stride);
#endif
    const int lo16 = ((t >> 5) & 1) * 1024 + (t >> 6) * 32 + (t & 31); // in 16-byte units: (2048 h + 64 w + 2 m) / 2
#pragma unroll
    for (int h = 0; h < 2; h++) { // one scalar base per row and four 2 KB slices (immediate offsets -4096 .. +2048 B), formed ONCE
        const int c = 1024 * h + 512;
        const gptr<d2v> ba = (gptr<d2v>)scalar_ptr_at(p.rows + rA * p.stride, c);
        const gptr<d2v> bb = (gptr<d2v>)scalar_ptr_at(p.rows + rB * p.stride, c);
#pragma unroll
        for (int i = 4 * h; i < 4 * h + 4; i++) {
            const d2v xa = __builtin_nontemporal_load(ba + (128 * i - c / 2) + lo16);
            const d2v xb = __builtin_nontemporal_load(bb + (128 * i - c / 2) + lo16);
            r.a[i] = xa.x;
            r.a[i + 8] = xa.y;
            r.b[i] = xb.x;
            r.b[i + 8] = xb.y;
        }
    }
}
__device__ __forceinline__ void swap_halves(double &first, double &second)
{
    // v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(first), (unsigned)__double2loint(second), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(first), (unsigned)__double2hiint(second), false, false);
    first = __hiloint2double((int)hi[0], (int)lo[0]);
    second = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void widen_rows(RawPair &r)
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        swap_halves(r.a[i], r.a[i + 8]);
        swap_halves(r.b[i], r.b[i + 8]);
    }
}

// shifted sums of one series: sum d, sum d^2  (d = x - x[0])
struct Stat {
    double s1, s2;
};

// zNormalize constants (xcorr.go:84-95 via the centred sample variance):
// variance from the shifted sums; flags for the (nil,0,0) and NaN outcomes.
__device__ __forceinline__ double variance(const Stat &s, double invN, double invNm1, bool &zero, bool &nan)
{
    const double var = (s.s2 - s.s1 * s.s1 * invN) * invNm1;
    // NaN or +-Inf statistics: every cc is NaN in the reference.  (Not `var - var != 0`:
    // under fp-contract the compiler fuses var's multiply into the subtraction and
    // the rounding residual makes it true for finite values.)
    nan = !__builtin_isfinite(var);
    zero = !nan && !(var > 0.0);  // sigma == 0 (rounding may leave -0 / a tiny negative)
    return var;
}

} // namespace occ4

} // namespace muse
