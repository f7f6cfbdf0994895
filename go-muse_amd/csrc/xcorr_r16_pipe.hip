// xcorr_r16_pipe.hip -- software-pipelined tuned kernel for n = 4096.
//
// Mathematics: identical to xcorr_fused_n4096 (xcorr_kernels.hip header; the
// reference path is xCorrWithX, /root/reference/xcorr.go:160-197).  Structure
// follows the round-1 ablation of that kernel (tools/ablate, profiles/):
// pure FFT arithmetic is ~4.5 ms per 1 M series, HBM streaming ~5.2 ms, but
// they ran one after the other.  Here, per workgroup (256 threads, one pair of
// series per pass, 2 workgroups per CU):
//   * the NEXT pair's rows are prefetched into registers (32 x 8-byte coalesced
//     nontemporal loads per thread) after pass 1 of the second FFT and stay in
//     flight through its two LDS transposes, passes 2-3 and the argmax;
//   * every barrier inside that window is a raw s_barrier behind
//     s_waitcnt lgkmcnt(0) only, so the prefetch is never drained early
//     (__syncthreads would wait vmcnt(0));
//   * nothing else touches global memory inside the window: the window opens
//     after pass 1 of the second FFT (whose W_4096^(k t) twiddles and the 16
//     spectrum values xc[t + 256 k] are L2 hits loaded just before), pass-2
//     twiddles W_256^(k c) come from a 4 KB LDS table;
//   * all row loads are unconditional (clamped index + select): a per-element
//     `if (j >= 0) load` made hipcc serialise the loads behind vmcnt(0) waits;
//   * wave reductions use DPP, z-normalisation needs ONE block reduction
//     (shifted sums) and argmax ONE; 1/sigma is applied to the single winning
//     value instead of to every sample (the argmax is scale invariant).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fft_device.h"
#include "xcorr_kernels.h"

namespace muse {

constexpr int PIPE_THREADS = 256;

// Diagnostic switches (tools/ablate only; the shipped instantiations use 0):
// each removes one cost so its share of the launch time can be measured.
enum { AB_NOLOAD = 1, AB_NOXC = 2, AB_NOTW1 = 4, AB_NOZN = 8, AB_NOARG = 16, AB_NOXCHG = 32, AB_NOBAR = 64,
       AB_NODFT = 128, AB_NOTW2 = 256, AB_NOSTAGGER = 512 };
constexpr int PIPE_XBUF = 16 * 272; // double2 exchange buffer: 69,632 B

// LDS-only barrier: this wave's LDS traffic is complete, then rendezvous.
// Outstanding global loads (the prefetch) stay in flight.
template <int ABL = 0>
__device__ __forceinline__ void lds_barrier()
{
    if (ABL & AB_NOBAR)
        return;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// Re-materialises a uniform pointer in SGPRs at this program point.  Loads
// through it use the saddr + (shared) VGPR-offset form, and the compiler can
// no longer hoist one 64-bit VGPR address per load out of the pair loop (that
// hoisting cost ~60 VGPRs and pushed loop invariants into scratch).
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
// 16-byte global load of one complex value (native vector type: HIP's double2
// struct cannot be copied out of an address_space(1) reference)
__device__ __forceinline__ double2 ldg2(gptr<double2> p, int i)
{
    const d2v x = ((gptr<d2v>)p)[i];
    return make_double2(x.x, x.y);
}
// compiler-only fence: nothing is scheduled across this point (bounds the
// live ranges the scheduler may create by hoisting loads)
__device__ __forceinline__ void pin()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// forward FFT, v[a] = x[t + 256 a] -> v[a] = X[t + 256 a], in two pieces so
// the caller can place the prefetch between them; exchange layouts as in
// xcorr_kernels.hip (conflict-free ds_write_b128 / ds_read_b128).
// piece 1: pass 1 (DFT over a) + twiddles W_4096^(k1 t) read from global (L2)
template <int ABL = 0>
__device__ __forceinline__ void fft4096_pass1(double2 (&v)[16], const double2 *__restrict__ tw1g, const int t)
{
    pin();
    const gptr<double2> tw1p = scalar_ptr(tw1g);
    double2 tw1[15];
#pragma unroll
    for (int k = 1; k < 16; k++)
        if (ABL & AB_NOTW1)
            tw1[k - 1] = make_double2(1.0 - 1e-9 * k * t, 1e-9 * k);
        else
            tw1[k - 1] = ldg2(tw1p, k * 256 + t);
    if (!(ABL & AB_NODFT))
        dft16(v);
#pragma unroll
    for (int k = 1; k < 16; k++)
        v[P16(k)] = cmul(v[P16(k)], tw1[k - 1]);
}
// piece 2: both LDS transposes, passes 2 and 3; touches LDS only
template <int ABL = 0>
__device__ __forceinline__ void fft4096_rest(double2 (&v)[16], double2 *xbuf, const double2 *tw2s, const int t)
{
    const int hi = t >> 4, lo = t & 15;
    lds_barrier<ABL>(); // previous readers of xbuf are done
    if (!(ABL & AB_NOXCHG)) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            xbuf[256 * k + t] = v[P16(k)];
    }
    lds_barrier<ABL>();
    if (!(ABL & AB_NOXCHG)) {
#pragma unroll
        for (int b = 0; b < 16; b++)
            v[b] = xbuf[256 * hi + 16 * b + lo];
    }
    if (!(ABL & AB_NODFT))
        dft16(v);
    lds_barrier<ABL>(); // every wave has consumed its exchange-A reads
    // pass-2 twiddles W_256^(k2 c) from the LDS table, applied in groups of four
    // and written straight into the exchange-B layout: keeps only 4 twiddles
    // (16 VGPRs) live at a time so the prefetch registers are never spilled.
#pragma unroll
    for (int g = 0; g < 4; g++) {
        double2 tw[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            tw[j] = (ABL & AB_NOTW2) ? make_double2(1.0 - 1e-9 * (g + j) * t, 1e-9 * j) : tw2s[(4 * g + j) * 16 + lo];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 4 * g + j;
            const double2 e = (k == 0) ? v[P16(k)] : cmul(v[P16(k)], tw[j]);
            if (ABL & AB_NOXCHG)
                v[P16(k)] = e;
            else
                xbuf[272 * k + 17 * hi + lo] = e;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier<ABL>();
    if (!(ABL & AB_NOXCHG)) {
#pragma unroll
        for (int c = 0; c < 16; c++)
            v[c] = xbuf[272 * hi + 17 * lo + c];
    }
    if (!(ABL & AB_NODFT))
        dft16(v);
    double2 w[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        w[k] = v[P16(k)];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = w[k];
}

struct RawPair {
    double a[16], b[16]; // element t + 256*i of the two (zero-padded) rows
    double ka, kb;       // first sample of each row (shift of the one-pass statistics)
};

template <bool PADDED, int ABL = 0>
__device__ __forceinline__ void issue_row_loads(RawPair &r, const FusedParams &p, long long pair, int t, int pad)
{
    if (ABL & AB_NOLOAD) {
        r.ka = 0.25;
        r.kb = 0.5;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            r.a[i] = 1e-3 * (double)(t * 16 + i) + (double)pair;
            r.b[i] = 2e-3 * (double)(t * 16 + i) - (double)pair;
        }
        return;
    }
    const long long rA = 2 * pair;
    const long long rB = (rA + 1 < p.M) ? rA + 1 : rA;
    const gptr<double> ra = scalar_ptr(p.rows + rA * p.stride);
    const gptr<double> rb = scalar_ptr(p.rows + rB * p.stride);
    r.ka = ra[0];
    r.kb = rb[0];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (PADDED) {
            int j = t + 256 * i - pad;
            j = j < 0 ? 0 : j; // clamped: the load is always issued, the value is masked later
            r.a[i] = __builtin_nontemporal_load(ra + j);
            r.b[i] = __builtin_nontemporal_load(rb + j);
        } else {
            r.a[i] = __builtin_nontemporal_load(ra + 256 * i + t);
            r.b[i] = __builtin_nontemporal_load(rb + 256 * i + t);
        }
    }
}

template <bool PADDED, int ABL = 0>
__global__ __launch_bounds__(PIPE_THREADS, 2) void xcorr_fused_n4096_pipe(const FusedParams p)
{
    __shared__ double2 xbuf[PIPE_XBUF];
    __shared__ double2 tw2s[256];
    __shared__ double red[64];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int N = p.N;
    const int pad = 4096 - N;
    const double dN = (double)N;

    tw2s[t] = p.tw2[t]; // workgroup-lifetime pass-2 twiddle table

    // De-phase the persistent workgroups.  Started together they run in
    // lockstep: every workgroup of the chip bursts its 64 KB prefetch at the same
    // moment and the two workgroups of a CU want the VALU, the LDS and the
    // barrier at the same time (ablation: the costs simply added up).  A one-off
    // start delay of hash(block) / 64 of a pair period spreads the phases.
    if (!(ABL & AB_NOSTAGGER)) {
        const unsigned slots = (blockIdx.x * 2654435761u) >> 26; // 0..63
        for (unsigned i = 0; i < slots; i++)
            __builtin_amdgcn_s_sleep(8); // ~512 cycles each; 64 slots ~ one pair period
    }

    long long pair = blockIdx.x;
    RawPair raw;
    if (pair < p.npairs)
        issue_row_loads<PADDED, ABL>(raw, p, pair, t, pad);
    __syncthreads();

    for (; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;

        // ---- consume the prefetched rows: d = x - K (pads -> 0)
        double2 v[16];
        const double KA = raw.ka, KB = raw.kb;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double da = raw.a[i] - KA, db = raw.b[i] - KB;
            if (PADDED) {
                const bool valid = t + 256 * i - pad >= 0;
                da = valid ? da : 0.0;
                db = valid ? db : 0.0;
            }
            v[i] = make_double2(da, db);
        }
        // ---- zNormalize (xcorr.go:84-95), one block reduction of shifted sums:
        // mean = K + S1/N, (N-1) var = S2 - S1^2/N.  K is a sample of the series,
        // so (mean-K)^2 <= (N-1) var and the cancellation is bounded by ~N ulp.
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            q[0] += v[i].x;
            q[1] = fma(v[i].x, v[i].x, q[1]);
            q[2] += v[i].y;
            q[3] = fma(v[i].y, v[i].y, q[3]);
        }
        if (!(ABL & AB_NOZN)) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                q[k] = wave_sum_dpp(q[k]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    red[wave * 4 + k] = q[k];
            }
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++)
                q[k] = (red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]);
        }
        ZnFlags fa, fb;
        double isa = zn_scale(q[0], q[1], N, fa); // 1/sigma, applied to the winner only
        double isb = zn_scale(q[2], q[3], N, fb);
        const double mA = q[0] / dN, mB = q[2] / dN;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (PADDED) {
                const bool valid = t + 256 * i - pad >= 0;
                v[i].x = valid ? v[i].x - mA : 0.0;
                v[i].y = valid ? v[i].y - mB : 0.0;
            } else {
                v[i].x -= mA;
                v[i].y -= mB;
            }
        }
        // a sigma == 0 / NaN series (or the missing partner of an odd last row)
        // must contribute exact zeros to the shared complex transform
        const bool deadA = fa.zero || fa.nan, deadB = fb.zero || fb.nan || !hasB;
        if (!deadA && !deadB) { // sigmas more than 2^16 apart (block-uniform, rare): see r16_device.h, pow2_inv_sigma
            const long long ea = (__double_as_longlong(isa) >> 52) & 0x7ff, eb = (__double_as_longlong(isb) >> 52) & 0x7ff;
            if (ea - eb > 16 || eb - ea > 16) {
                const double sA = __longlong_as_double(ea << 52), sB = __longlong_as_double(eb << 52); // exact powers of two <= 1/sigma
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    v[i].x *= sA;
                    v[i].y *= sB;
                }
                isa /= sA;
                isb /= sB;
            }
        }
        if (deadA || deadB) { // block-uniform, rare
#pragma unroll
            for (int i = 0; i < 16; i++) {
                v[i].x = deadA ? 0.0 : v[i].x;
                v[i].y = deadB ? 0.0 : v[i].y;
            }
        }

        // ---- Z = FFT(yA + i yB)
        fft4096_pass1<ABL>(v, p.tw1, t);
        fft4096_rest<ABL>(v, xbuf, tw2s, t);
        // ---- V[f] = Z[f] * conj(X[f]) / n   (f = t + 256 k)
        {
            pin();
            const gptr<double2> xcp = scalar_ptr(p.xc);
            double2 xc[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (ABL & AB_NOXC)
                    xc[k] = make_double2(1.0 - 1e-9 * k * t, 1e-9 * k);
                else
                    xc[k] = ldg2(xcp, 256 * k + t);
#pragma unroll
            for (int k = 0; k < 16; k++)
                v[k] = cmul(v[k], xc[k]);
        }
        // ---- ccA + i ccB = FFT(V)   (unscaled by 1/sigma)
        fft4096_pass1<ABL>(v, p.tw1, t);
        // ---- open the prefetch window: next pair's rows stay in flight until
        // the top of the next iteration; nothing else reads global memory in it.
        pin();
        {
            // unconditional (the last iteration re-reads its own pair): a
            // conditional prefetch made the register allocator park all of
            // `raw` in scratch memory
            long long nxt = pair + gridDim.x;
            nxt = nxt < p.npairs ? nxt : pair;
            issue_row_loads<PADDED, ABL>(raw, p, nxt, t, pad);
        }
        pin();
        fft4096_rest<ABL>(v, xbuf, tw2s, t);

        // ---- maxAbsIndex (xcorr.go:39-50), index = t + 256 k; one barrier
        double ma = 0.0, mb = 0.0, sa = 0.0, sb = 0.0;
        int ka = 0, kb = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const double aa = fabs(v[k].x), ab = fabs(v[k].y);
            if (aa > ma) { ma = aa; sa = v[k].x; ka = k; }
            if (ab > mb) { mb = ab; sb = v[k].y; kb = k; }
        }
        if (ABL & AB_NOARG) { // diagnostic: keep the values live, skip reductions + barrier
            if (ma + mb == 12345.678) {
                p.mv[rA] = sa + sb;
                p.lag[rA] = ka + kb;
            }
            continue;
        }
        const double wa = wave_max_dpp(ma), wb = wave_max_dpp(mb);
        const int ia_ = wave_min_i_dpp((ma == wa && wa > 0.0) ? (t + 256 * ka) : 0x7fffffff);
        const int ib_ = wave_min_i_dpp((mb == wb && wb > 0.0) ? (t + 256 * kb) : 0x7fffffff);
        // per wave {max |cc|, signed value, index}: red[16 + 3w ..] (A), red[28 + 3w ..] (B)
        if (ia_ == 0x7fffffff) {
            if (lane == 0) {
                red[16 + 3 * wave] = 0.0;
                red[17 + 3 * wave] = (wave == 0) ? v[0].x : 0.0; // cc[0] lives in thread 0
                red[18 + 3 * wave] = (double)0x7fffffff;
            }
        } else if (t + 256 * ka == ia_ && ma == wa) {
            red[16 + 3 * wave] = wa;
            red[17 + 3 * wave] = sa;
            red[18 + 3 * wave] = (double)ia_;
        }
        if (ib_ == 0x7fffffff) {
            if (lane == 0) {
                red[28 + 3 * wave] = 0.0;
                red[29 + 3 * wave] = (wave == 0) ? v[0].y : 0.0;
                red[30 + 3 * wave] = (double)0x7fffffff;
            }
        } else if (t + 256 * kb == ib_ && mb == wb) {
            red[28 + 3 * wave] = wb;
            red[29 + 3 * wave] = sb;
            red[30 + 3 * wave] = (double)ib_;
        }
        lds_barrier();
        if (t < 2 && (t == 0 || hasB)) {
            const int base = t == 0 ? 16 : 28;
            double best = red[base], bsv = red[base + 1], bidx = red[base + 2];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const double m = red[base + 3 * w], s = red[base + 3 * w + 1], ix = red[base + 3 * w + 2];
                if (m > best || (m == best && ix < bidx)) {
                    best = m;
                    bsv = s;
                    bidx = ix;
                }
            }
            const int idx = (best > 0.0) ? (int)bidx : 0; // nothing above 0: index 0, mv = cc[0]
            double mv = ((best > 0.0) ? bsv : red[base + 1]) * (t == 0 ? isa : isb);
            int lag = idx > 2048 ? idx - 4096 : idx;
            const ZnFlags f = t == 0 ? fa : fb;
            if (f.zero) { mv = 0.0; lag = 0; }
            if (f.nan) { mv = __builtin_nan(""); lag = 0; }
            const long long r = t == 0 ? rA : rB;
            p.mv[r] = mv;
            p.lag[r] = lag;
        }
    }
}

hipError_t launch_fused_pipe(const FusedParams &p, int num_cus, hipStream_t stream)
{
    long long grid = p.npairs;
    int mult = 1; // resident workgroups per CU x mult (MUSE_HIP_GRID_MULT: tuning aid)
    if (const char *m = getenv("MUSE_HIP_GRID_MULT"))
        mult = atoi(m) > 0 ? atoi(m) : mult;
    const long long cap = (long long)num_cus * 2 * mult;
    if (grid > cap)
        grid = cap;
    if (p.N < 4096)
        hipLaunchKernelGGL(xcorr_fused_n4096_pipe<true>, dim3((unsigned)grid), dim3(PIPE_THREADS), 0, stream, p);
    else
        hipLaunchKernelGGL(xcorr_fused_n4096_pipe<false>, dim3((unsigned)grid), dim3(PIPE_THREADS), 0, stream, p);
    return hipGetLastError();
}

} // namespace muse
