// xcorr_real.hip -- n = 32768 WITHOUT a trip through memory: ONE real series per 1024-thread workgroup as a real transform on the
// 16384-point complex machinery of xcorr_small.hip (round 5).
//
// Mathematics: xCorrWithX, /root/reference/xcorr.go:160-197 (z-normalise, leading zero pad, forward real transform,
// multiply by the conjugate reference spectrum, inverse real transform, 1/n, first greatest |cc|, lag unwrap), for n = 2 M,
// M = 16384.  The long-series kernel (xcorr_long.hip) packs two series into one complex transform of n points, which fits
// neither the LDS nor the registers of a CU: a four-step transform whose slice crosses memory four times per transform pair
// (5 x the algorithmic bytes, the memory side saturated at 5.8 TB/s: profiles/r04_counters.json).  Here the two halves of
// ONE series share a complex transform of M points, which does fit (the n = 16384 kernel's geometry: 1024 threads x 16
// points, every transpose in two half rounds through 139 KB of LDS):
//
//   z[m] = d[2m] + i d[2m+1]                      (d = the centred, scaled, zero-padded series; one 16-byte request per point)
//   Z = FFT_M(z)                                  (small::forward<14>)
//   E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / 2i      (spectra of the even / odd samples)
//   Y[k] = E + W O,   Y[M-k] = conj(E - W O),     W = W_n^k            (the series' spectrum, bins 0 .. M)
//   P[k] = Y[k] xc[k],   P[M-k] = Y[M-k] xc[M-k],   xc = conj(X) / n   (FusedParams::xc, all n bins)
//   A = P[k] + conj P[M-k],   B = (P[k] - conj P[M-k]) W                (cc = FFT_n(P) split into even / odd lags:
//   C[k] = A + i B,   C[M-k] = conj A + i conj B                         cc[2m] + i cc[2m+1] = FFT_M(C)[m])
//   c = FFT_M(C):   cc[2m] = Re c[m],  cc[2m+1] = Im c[m]               (small::forward<14> again)
//
// A thread holds Z[j + r S] (S = 1024, r = 0 .. 15); bin k's partner M - k = (S - j) + (15 - r) S lives in the thread of column
// S - j.  The pair (k, M - k) shares E, O, W O and both products, so the threads of columns j and S - j split the sixteen pairs
// between them: each puts its upper eight bins (r >= 8) into the half buffer, forms C[k] AND C[M-k] for its lower eight (32 fp64
// instructions per pair) from its own bin and the partner's, and writes C[M-k] back into the very slot the partner's bin came out
// of -- the volume of one transpose, three workgroup barriers, no second set of registers.
// Column 0 pairs inside itself (k = r S with (16 - r) S; k = 0 with the Nyquist bin, k = M / 2 with itself): the same code with
// one more slot of offset and a ninth evaluation in that thread's wave.
//
// One series per transform: nothing to isolate (no partner a NaN could poison, no sigma spread inside a transform), so no
// redo list.  Per series 8 N bytes in and 12 out -- the row never leaves the CU between them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "foldk_device.h"
#include "small_device.h"
#include "two_device.h"

// compile-time switches of tools/ablate/ab_real.sh (one box, profiles/r05_real_transform.txt): the table window (2 ... 5 pairs ahead:
// inside the noise) and the padded rows' 16-byte requests at 8-byte alignment (- 5 ... 9 % at N = 20000 / 24001)
#ifndef MUSE_REAL_AHEAD
#define MUSE_REAL_AHEAD 3
#endif
#ifndef MUSE_REAL64_PRE
#define MUSE_REAL64_PRE 0
#endif
#ifndef MUSE_REAL64_BATCH
#define MUSE_REAL64_BATCH 4
#endif
#ifndef MUSE_REAL_WIDE
#define MUSE_REAL_WIDE 1
#endif
// timing ablations (results are WRONG with any bit set; tools/ablate/ab_real_phases.sh): 1 no row requests inside the loop, 2 no mirror
// stage, 4 no second transform, 8 no first transform, 16 no statistics reduction
#ifndef MUSE_REAL_ABL
#define MUSE_REAL_ABL 0
#endif

namespace muse {

namespace real {

using namespace occ4;
using namespace fold;

// the real series' spectrum at a mirror pair of bins: Z = Z[k], Zm = Z[M-k], W = W_n^k  ->  k = 2 Y[k], m = 2 Y[M-k]
struct TwoBins {
    double2 k, m;
};
__device__ __forceinline__ TwoBins spectrum_pair(const double2 Z, const double2 Zm, const double2 W)
{
    const double2 E2 = make_double2(Z.x + Zm.x, Z.y - Zm.y);                 // Z + conj Zm = 2 E
    const double2 O2 = make_double2(Z.y + Zm.y, Zm.x - Z.x);                 // (Z - conj Zm) / i = 2 O
    const double2 T = cmul(W, O2);
    return TwoBins{make_double2(E2.x + T.x, E2.y + T.y),                     // 2 Y[k] = 2 E + T
                   make_double2(E2.x - T.x, T.y - E2.y)};                    // 2 Y[M-k] = conj(2 E - T)
}
// one mirror pair: Z = Z[k], Zm = Z[M-k], W = W_n^k, xk / xm = the factors of bins k / M - k  ->  k = 2 C[k], m = 2 C[M-k]
__device__ __forceinline__ TwoBins mirror_pair(const double2 Z, const double2 Zm, const double2 W, const double2 xk, const double2 xm)
{
    const TwoBins Y = spectrum_pair(Z, Zm, W);
    const double2 P = cmul(Y.k, xk), Pm = cmul(Y.m, xm);
    const double2 A = make_double2(P.x + Pm.x, P.y - Pm.y);                  // P + conj Pm
    const double2 B = cmul(make_double2(P.x - Pm.x, P.y + Pm.y), W);         // (P - conj Pm) W
    return TwoBins{make_double2(A.x - B.y, A.y + B.x),                       // A + i B
                   make_double2(A.x + B.y, B.x - A.y)};                      // conj A + i conj B
}

// cos / sin of 2 pi r / 32: W_32^r = C32[r] - i C32[8 - r] for r = 0 .. 8
constexpr double C32[9] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                           0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785, 0.0};
__device__ __forceinline__ double2 w32(const int r) // W_32^r, r = 0 .. 15 (compile-time r)
{
    return r <= 8 ? make_double2(C32[r], -C32[8 - r]) : make_double2(-C32[16 - r], -C32[r - 8]);
}

// The mirror stage of a 16 S-point spectrum held as Z[j + r S] at v[BR16(r)] (S threads, column j; S = 1024 or 256): for every mirror
// pair of bins the two values C (mirror_pair) from the thread's own bin and the partner thread's, in place -- v[BR16(r)] <- 2 C of
// the thread's bin j + r S for all sixteen r.  Each thread puts its upper eight bins (r >= 8, slot r - 8 of its column) into the
// half buffer b, evaluates its lower eight pairs and writes the partner's C back into the slot the partner's bin came out of.
//   MODE 0: bin k pairs with (H - k) mod H, H = 16 S: the partner is column S - j, register 15 - r; column 0 pairs inside itself
//           (r with 16 - r: one slot further), its pair r = 0 is bin 0 with itself (the real transform's DC and Nyquist bins) and
//           its bin H / 2 (register 8) pairs with itself: a ninth evaluation, ninth(v8), in that thread's wave.
//   MODE 1: bin k pairs with H - 1 - k: column S - 1 - j, register 15 - r, no special column.
// W(r) = the twiddle of bin j + r S = Wj W_32^r; req(r) requests the pair's two table values (AHEAD pairs in front of their use),
// fac(raw, W) turns them into the factors of bins k and its mirror; dc0: column 0's bin 0 gets the factor 0 (a series that is not
// centred leaves its mean in that bin alone).
template <int MODE, int AHEAD, typename RAW, int S = 1024, typename REQ, typename FAC, typename NINTH>
__device__ __forceinline__ void mirror_stage(double2 (&v)[16], double2 *b, const int j_, const int wave, const double2 Wj, const bool dc0,
                                             REQ req, FAC fac, NINTH ninth)
{
    using namespace small;
    constexpr int PK = padk(S); // S columns per slot, padded 17 / 16 (S = 1024: the 16384-point machinery; S = 256: the 4096-point one)
    int jm = j_;
    asm volatile("" : "+v"(jm)); // (addresses derived here, not hoisted out of the row loop)
    jm &= S - 1;
    const bool col0 = MODE == 0 && jm == 0;
    const int cm = MODE == 0 ? ((S - jm) & (S - 1)) : (S - 1 - jm); // the partner's column (MODE 0, column 0: itself)
    const int wbase = jm + (jm >> 4);                               // own column, slot 0
    const int rbase = cm + (cm >> 4);                               // partner's column, slot 0
    const int rbm = rbase + (col0 ? PK : 0);                        // column 0 pairs bin r S with bin (16 - r) S: one slot further
    lds_barrier(); // (the transform's last readers of the buffer are done)
#pragma unroll
    for (int s = 0; s < 8; s++)
        lds_st2(b + wbase + s * PK, v[BR16(8 + s)]);
    RAW raw[8];
#pragma unroll
    for (int r = 0; r < AHEAD; r++)
        raw[r] = req(r);
    const double2 v8 = v[BR16(8)]; // (column 0's ninth pair needs its bin H / 2 once more)
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        fence();
        if (r + AHEAD < 8)
            raw[r + AHEAD] = req(r + AHEAD);
        double2 *const slot = b + ((MODE == 0 && r == 0) ? rbase + 7 * PK : rbm + (7 - r) * PK); // the partner's mirror bin (its register 15 - r)
        double2 zm = lds_ld2(slot);
        if (MODE == 0 && r == 0) { // column 0: bin 0 pairs with itself
            // (component by component: a ?: between two double2 lvalues is a select of ADDRESSES, and an array whose element's address
            // escapes into one is never split into registers)
            zm.x = col0 ? v[BR16(0)].x : zm.x;
            zm.y = col0 ? v[BR16(0)].y : zm.y;
        }
        fence();
        const double2 W = r == 0 ? Wj : cmul(Wj, w32(r));
        TwoBins f = fac(raw[r], W);
        if (MODE == 0 && r == 0) {
            f.k.x = (dc0 && col0) ? 0.0 : f.k.x;
            f.k.y = (dc0 && col0) ? 0.0 : f.k.y;
        }
        const TwoBins o = mirror_pair(v[BR16(r)], zm, W, f.k, f.m);
        v[BR16(r)] = o.k;
        // the mirror bin's C goes back INTO THE SLOT its Z came out of: nobody else reads or writes that slot, so no barrier between
        // the read and the write and no registers held for a second exchange -- the partner finds the C of its register 8 + s in the
        // slot s of its own column.  (Column 0, r = 0: the mirror bin has no register; that slot belongs to the pair r = 1.)
        if (MODE == 1 || r > 0 || !col0)
            lds_st2(slot, o.m);
    }
    double2 c8 = make_double2(0.0, 0.0);
    if (MODE == 0 && wave == 0)
        c8 = ninth(v8);
    lds_barrier(); // (every partner has written back)
#pragma unroll
    for (int s = 0; s < 8; s++)
        v[BR16(8 + s)] = lds_ld2(b + wbase + s * PK);
    if (MODE == 0) {
        const bool mine = wave == 0 && col0;
        v[BR16(8)] = make_double2(mine ? c8.x : v[BR16(8)].x, mine ? c8.y : v[BR16(8)].y);
    }
}
// The same stage on the output of small::forward_split_dif (M = 16384 as 16 x 1024): thread j = 64 w + c holds bin w + 16 (c + 64 r) at
// v[BR16(r)].  Bin k's mirror M - k sits in wave (16 - w) mod 16: for w >= 1 at column 63 - c, register 15 - r (MODE 1 above, across two
// waves; wave 8 pairs inside itself); wave 0 pairs inside itself like MODE 0 with S = 64 (column 64 - c, register 15 - r; column 0: r with
// 16 - r, bin 0 with the Nyquist bin, bin M / 2 with itself).  Which of the two a wave does is a scalar; the buffer is addressed as sixteen
// wave images of eight slots x 68 (the wave-local transforms' padding).  W(r) = W_n^(w + 16 c) W_32^r.
template <int AHEAD, typename RAW, typename REQ, typename FAC, typename NINTH>
__device__ __forceinline__ void mirror_stage_split(double2 (&v)[16], double2 *b, const int j_, const int wave, const double2 Wj, const bool dc0,
                                                   REQ req, FAC fac, NINTH ninth)
{
    constexpr int PK = 68, WR = 8 * PK;
    int jm = j_;
    asm volatile("" : "+v"(jm)); // (addresses derived here, not hoisted out of the row loop)
    jm &= 63;
    const bool wave0 = wave == 0;
    const bool col0 = wave0 && jm == 0;
    const int cm = wave0 ? ((64 - jm) & 63) : (63 - jm); // the partner's column (wave 0, column 0: itself)
    const int pw = (16 - wave) & 15;                     // the partner's wave
    const int wbase = wave * WR + jm + (jm >> 4);
    const int rbase = pw * WR + cm + (cm >> 4);
    const int rbm = rbase + (col0 ? PK : 0);
    lds_barrier(); // (the transform's last readers of the buffer are done)
#pragma unroll
    for (int s = 0; s < 8; s++)
        lds_st2(b + wbase + s * PK, v[BR16(8 + s)]);
    RAW raw[8];
#pragma unroll
    for (int r = 0; r < AHEAD; r++)
        raw[r] = req(r);
    const double2 v8 = v[BR16(8)];
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        fence();
        if (r + AHEAD < 8)
            raw[r + AHEAD] = req(r + AHEAD);
        double2 *const slot = b + (r == 0 ? rbase + 7 * PK : rbm + (7 - r) * PK);
        double2 zm = lds_ld2(slot);
        if (r == 0) { // (component by component: see mirror_stage)
            zm.x = col0 ? v[BR16(0)].x : zm.x;
            zm.y = col0 ? v[BR16(0)].y : zm.y;
        }
        fence();
        const double2 W = r == 0 ? Wj : cmul(Wj, w32(r));
        TwoBins f = fac(raw[r], W);
        if (r == 0) {
            f.k.x = (dc0 && col0) ? 0.0 : f.k.x;
            f.k.y = (dc0 && col0) ? 0.0 : f.k.y;
        }
        const TwoBins o = mirror_pair(v[BR16(r)], zm, W, f.k, f.m);
        v[BR16(r)] = o.k;
        if (r > 0 || !col0)
            lds_st2(slot, o.m);
    }
    double2 c8 = make_double2(0.0, 0.0);
    if (wave == 0)
        c8 = ninth(v8);
    lds_barrier(); // (every partner has written back)
#pragma unroll
    for (int s = 0; s < 8; s++)
        v[BR16(8 + s)] = lds_ld2(b + wbase + s * PK);
    v[BR16(8)] = make_double2(col0 ? c8.x : v[BR16(8)].x, col0 ? c8.y : v[BR16(8)].y);
}
// v[BR16(r)] -> v[r]: the natural order the next transform takes its input in
__device__ __forceinline__ void natural_order(double2 (&v)[16])
{
    double2 w[16];
#pragma unroll
    for (int r = 0; r < 16; r++)
        w[r] = v[BR16(r)];
#pragma unroll
    for (int r = 0; r < 16; r++)
        v[r] = w[r];
}
struct RawPairXC {
    double2 a, b;
};

} // namespace real

// One real series of n = 2 M samples per workgroup iteration on the M = 2^LM-point complex transform: LM = 14 (n = 32768, 1024 threads,
// one workgroup per CU) and LM = 13 (n = 16384, 512 threads, 70 KB of LDS: TWO workgroups per CU, one's barriers under the other's
// arithmetic -- where the pair-packed 16384-point kernel, xcorr_fused_small<14>, has one workgroup of 16 waves per CU in lockstep).
// PADDED: N < n (leading zero pad, n / 2 < N)
// SPLIT (LM = 14): the transforms as 16 x 1024 (small::forward_split_dif / _dit: two of a transform's three transposes inside a wave)
// MULTI (SPLIT only; SURVEY 8f-2, README.md:10-13 of the reference): R references against the group in ONE pass -- the row is read,
// normalised and first-transformed once, Z is parked in the workgroup's 256 KB scratch slice (every thread re-reads what it wrote),
// and every reference takes mirror stage, second transform and argmax from there: per reference ~half a pass, 1 x the row bytes.
template <int LM, bool PADDED, bool SPLIT = false, bool MULTI = false>
__device__ __forceinline__ void real_one_series(const FusedParams &p)
{
    static_assert(!MULTI || SPLIT, "many references run on the split form");
    using namespace occ4;
    using namespace fold;
    using namespace small;
    using namespace real;
    constexpr int M = 1 << LM, n = 2 * M, S = M / 16, R1 = LM == 14 ? 4 : 2; // S threads, 16 complex points each
    static_assert(LM == 13 || LM == 14, "the 8192- and 16384-point transforms of small_device.h");
    static_assert(!SPLIT || LM == 14, "the split is built for 16384 = 16 x 1024");
    __shared__ double red[112];
    __shared__ double2 g2l[8 * R1];
    __shared__ double2 xbuf[(S / 64) * 544]; // the half buffer of the transposes: 8 S padded points = 139 KB (LM = 14), 70 KB (LM = 13)
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j = SPLIT ? column_of_lane<10>(t) : column_of_lane<LM>(t); // (SPLIT: j = 64 wave + the lane's column of the 1024-point transform)
    double2 *const b = xbuf;
    const int N = PADDED ? p.N : n, pad = PADDED ? n - N : 0;
    const double invN = PADDED ? p.invN : 1.0 / (double)n, invNm1 = PADDED ? p.invNm1 : 1.0 / (double)(n - 1);
    const double2 *__restrict__ twm = p.twm;
    const double2 *__restrict__ gs = p.gsmall; // the M-point transform's lane-ordered pass tables
    const double2 *__restrict__ xc = p.xc;
    if (t < 8 * R1)
        g2l[t] = tw_factor<R1>(twm, t % R1, t / R1);
    __syncthreads();
    const long long total = p.M;

    // point m = j + i S of z holds the samples 2m - pad and 2m + 1 - pad of the row (a pad position: 0): one 16-byte request per
    // point (PADDED: at an 8-byte aligned address when the pad is odd -- global loads only need dword alignment); a request that lies in
    // the pad entirely (2 (i + 1) S <= pad: wave-uniform) is pointed at the row's own first samples -- an L2 hit instead of the end
    // of the previous row streamed from HBM only to be masked -- and one that straddles the pad's end reads at most 2 S samples in
    // front of the row (the allocation's guard: capi_group.hip, GROUP_GUARD >= 2 S).
    double x0[16], x1[16], K; // (plain doubles: an array of HIP's double2 struct filled under a branch stays in scratch)
    const auto request = [&](long long row) __attribute__((always_inline)) {
        if (row >= total)
            row = total - 1; // (nothing left: an L2-hot dummy)
        const double *const r = p.rows + row * p.stride;
        int jr = j;
        asm volatile("" : "+v"(jr)); // (offsets derived per request, not hoisted)
        jr &= S - 1;
        K = scalar_ptr(r)[0];
        if (!PADDED) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const d2v s = __builtin_nontemporal_load((gptr<d2v>)scalar_ptr_at(r, 2 * i * S) + (unsigned)jr);
                x0[i] = s.x;
                x1[i] = s.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool all_pad = i < 8 && 2 * (i + 1) * S <= pad;
                const long long off = all_pad ? 0ll : 2ll * i * S - pad;
#if MUSE_REAL_WIDE
                // (a 16-byte request at an 8-byte aligned address when the pad is odd: global loads only need dword alignment)
                typedef d2v __attribute__((aligned(8))) d2u;
                const d2u s = __builtin_nontemporal_load((gptr<d2u>)scalar_ptr_at(r, off) + (unsigned)jr);
                x0[i] = s.x;
                x1[i] = s.y;
#else
                x0[i] = __builtin_nontemporal_load(scalar_ptr_at(r, off) + (unsigned)(2 * jr));
                x1[i] = __builtin_nontemporal_load(scalar_ptr_at(r, off + 1) + (unsigned)(2 * jr));
#endif
            }
        }
    };
    if (blockIdx.x < total)
        request(blockIdx.x);
    for (long long row = blockIdx.x; row < total; row += gridDim.x) {
        // ---- d = x - K with K the first sample, shifted statistics (xcorr.go:84-95)
        double2 v[16];
        int js = j;
        asm volatile("" : "+v"(js)); // (validity masks derived per iteration, not hoisted)
        js &= S - 1;
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = 2 * (js + i * S) - pad; // sample index of the point's real part
            const bool v0 = !PADDED || i >= 8 || e >= 0, v1 = !PADDED || i >= 8 || e + 1 >= 0; // (pad < n / 2: the upper half is data)
            const double d0 = v0 ? x0[i] - K : 0.0, d1 = v1 ? x1[i] - K : 0.0;
            v[i] = make_double2(d0, d1);
            q0 += d0;
            q1 = fma(d0, d0, q1);
            q2 += d1;
            q3 = fma(d1, d1, q3);
        }
        if (!(MUSE_REAL_ABL & 16))
            pair_sum4<S>(q0, q1, q2, q3, red, wave);
        bool zero, nan;
        const double var0 = variance(Stat{q0 + q2, q1 + q3}, invN, invNm1, zero, nan);
        const bool dead = zero || nan;
        // the series goes into the transform at O(1): an exact power-of-two scale close to 1 / sigma, folded into the mean removal
        const double sc = dead ? 1.0 : pow2_inv_sigma(var0);
        const double var = uniform(var0 * sc * sc);
        const double mean = (q0 + q2) * invN * sc;
        asm volatile("" : "+v"(js));
        js &= S - 1;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = 2 * (js + i * S) - pad;
            const bool v0 = !PADDED || i >= 8 || e >= 0, v1 = !PADDED || i >= 8 || e + 1 >= 0;
            v[i].x = (v0 && !dead) ? fma(v[i].x, sc, -mean) : 0.0;
            v[i].y = (v1 && !dead) ? fma(v[i].y, sc, -mean) : 0.0;
        }
        if constexpr (SPLIT) {
            // ---- Z = FFT_M(z): Z[w + 16 (c + 64 r)] at v[BR16(r)], j = 64 w + c
            int jm = j;
            asm volatile("" : "+v"(jm));
            jm &= S - 1;
            const double2 *__restrict__ ws = p.wsplit;
            if (!(MUSE_REAL_ABL & 8))
            forward_split_dif(v, b, g2l, p.gsmall_b, j, wave,
                              [&](const int k1) __attribute__((always_inline)) { return ldg2u(scalar_ptr_at(ws, (k1 - 1) * 1024), (unsigned)jm); });
        } else {
            // ---- Z = FFT_M(z): Z[j + r S] at v[BR16(r)]
            if (!(MUSE_REAL_ABL & 8))
            forward<LM>(v, b, g2l, gs, j);
        }
        typedef d2v __attribute__((address_space(1))) *gd2p;
        double2 *const zpark = MULTI ? p.gscratch + (size_t)blockIdx.x * (size_t)M : nullptr; // the workgroup's slice: Z [i][t]
        if constexpr (MULTI) {
#pragma unroll
            for (int i = 0; i < 16; i++)
                *((gd2p)scalar_ptr_at(zpark, i * S) + (unsigned)(t & (S - 1))) = d2v{v[i].x, v[i].y};
        }
        const int R = MULTI ? p.R : 1;
#pragma clang loop unroll(disable)
        for (int ref = 0; ref < R; ref++) {
        if constexpr (MULTI) {
            if (ref > 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const d2v z = *((gd2p)scalar_ptr_at(zpark, i * S) + (unsigned)(t & (S - 1)));
                    v[i] = make_double2(z.x, z.y);
                }
            }
        }
        double *const mv_out = MULTI ? uniform_ptr(p.mv_many[ref]) : p.mv;
        int *const lag_out = MULTI ? uniform_ptr(p.lag_many[ref]) : p.lag;
        if constexpr (SPLIT) {
            // ---- mirror pairs in that order; the reference's spectrum at the thread's bins and their mirrors from FusedParams::xcw
            // ([16384]: the reference's bin M / 2, which pairs with itself)
            int jm = j;
            asm volatile("" : "+v"(jm));
            jm &= S - 1;
            const double2 *__restrict__ xw = MULTI ? uniform_ptr(p.xcp_many[ref]) : p.xcw;
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)(2 * ((jm >> 6) + 16 * (jm & 63)))); // W_n^(w + 16 c)
            if (!(MUSE_REAL_ABL & 2))
            mirror_stage_split<MUSE_REAL_AHEAD, RawPairXC>(
                v, b, j, wave, Wj, false,
                [&](const int r) __attribute__((always_inline)) {
                    return RawPairXC{ldg2u(scalar_ptr_at(xw, r * 1024), (unsigned)jm), ldg2u(scalar_ptr_at(xw, 8192 + r * 1024), (unsigned)jm)};
                },
                [&](const RawPairXC &x, const double2) __attribute__((always_inline)) { return TwoBins{x.a, x.b}; },
                [&](const double2 v8) __attribute__((always_inline)) { // bin M / 2 pairs with itself, W = -i
                    const double2 xh = ldg2u(scalar_ptr_at(xw, 16384), 0u);
                    return mirror_pair(v8, v8, make_double2(0.0, -1.0), xh, xh).k;
                });
            natural_order(v);
            // ---- c = FFT_M(C): 2 cc[2m] + 2 i cc[2m+1] with m = j + r S at v[BR16(r)]
            if (!(MUSE_REAL_ABL & 4))
            forward_split_dit(v, b, g2l, p.gsmall_b, gs + 8 * 16 * R1, j, wave);
        } else {
        // ---- mirror pairs: Y, the product with the reference's spectrum, re-tangled for the second transform (mirror_stage above)
        {
            int jm = j;
            asm volatile("" : "+v"(jm)); // (addresses derived here, not hoisted out of the row loop)
            jm &= S - 1;
            // W_n^(j + r S) = W_n^j W_32^r: one table entry (W_65536^(j 65536 / n)) and seven constant factors; the reference's spectrum
            // at the two bins of a pair, xc[j + r S] and xc[M - j - r S]
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)((65536 / n) * jm));
            if (!(MUSE_REAL_ABL & 2))
            mirror_stage<0, MUSE_REAL_AHEAD, RawPairXC, S>(
                v, b, j, wave, Wj, false,
                [&](const int r) __attribute__((always_inline)) {
                    return RawPairXC{ldg2u(scalar_ptr_at(xc, r * S), (unsigned)jm), ldg2u(scalar_ptr_at(xc, M - r * S - S), (unsigned)(S - jm))};
                },
                [&](const RawPairXC &x, const double2) __attribute__((always_inline)) { return TwoBins{x.a, x.b}; },
                [&](const double2 v8) __attribute__((always_inline)) { // bin M / 2 pairs with itself, W = -i
                    const double2 xh = ldg2u(scalar_ptr_at(xc, M / 2), 0u);
                    return mirror_pair(v8, v8, make_double2(0.0, -1.0), xh, xh).k;
                });
        }
        natural_order(v); // (2 C[j + r S] sits at v[BR16(r)]; the transform takes its input in natural order)
        // ---- c = FFT_M(C): 2 cc[2m] + 2 i cc[2m+1] with m = j + r S at v[BR16(r)]
        if (!(MUSE_REAL_ABL & 4))
        forward<LM>(v, b, g2l, gs, j);
        }
        // ---- maxAbsIndex (xcorr.go:39-50): ascending r, real part before imaginary part = ascending lag index for this thread
        double sv = 0.0;
        int code = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const double a0 = v[BR16(r)].x, a1 = v[BR16(r)].y;
            const bool g0 = fabs(a0) > fabs(sv);
            sv = g0 ? a0 : sv;
            code = g0 ? 2 * r : code;
            const bool g1 = fabs(a1) > fabs(sv);
            sv = g1 ? a1 : sv;
            code = g1 ? 2 * r + 1 : code;
        }
        const double ma = fabs(sv);
        const int ia = 2 * (j + (code >> 1) * S) + (code & 1);
        const double cc0 = v[0].x; // (column 0: cc[0], reported when nothing is above 0)
        fence();
        if (!MULTI && !(MUSE_REAL_ABL & 1))
            request(row + gridDim.x); // the next row: in flight during the reductions and the result write-out
        fence();
        double pa = ma, pb = 0.0;
        pair_max2<S>(pa, pb, red, wave);
        int ca = (ma == pa && pa > 0.0) ? ia : 0x7fffffff, cb = 0x7fffffff;
        pair_min_i2<S>(ca, cb, red, wave);
        const bool own = ca == 0x7fffffff ? j == 0 : (ia == ca && ma == pa);
        if (own) {
            double y = __builtin_amdgcn_rsq(var);
            y = y * fma(-0.5 * var * y, y, 1.5);
            y = y * fma(-0.5 * var * y, y, 1.5);
            double mv = (ca == 0x7fffffff ? cc0 : sv) * (0.5 * y); // (the re-tangled spectrum carries 2 C)
            const int idx = ca == 0x7fffffff ? 0 : ca;
            int lag = idx > n / 2 ? idx - n : idx;
            if (zero) { mv = 0.0; lag = 0; }               // xcorr.go:166-167
            if (nan) { mv = __builtin_nan(""); lag = 0; }
            mv_out[row] = mv;
            lag_out[row] = lag;
        }
        } // (references)
        // (many references: the next row is requested behind the loop -- requested inside it, the 64 registers of the requests
        // are live across every reference's transforms and 140 registers go to scratch)
        if (MULTI && !(MUSE_REAL_ABL & 1))
            request(row + gridDim.x);
    }
}
template <bool PADDED>
__global__ __launch_bounds__(1024, 4) void xcorr_fused_real32k(const FusedParams p)
{
    real_one_series<14, PADDED>(p);
}
template <bool PADDED>
__global__ __launch_bounds__(512, 4) void xcorr_fused_real16k(const FusedParams p)
{
    real_one_series<13, PADDED>(p);
}
template <bool PADDED>
__global__ __launch_bounds__(1024, 4) void xcorr_fused_real32k_split(const FusedParams p)
{
    real_one_series<14, PADDED, true>(p);
}
template <bool PADDED>
__global__ __launch_bounds__(1024, 4) void xcorr_fused_real32k_multi(const FusedParams p)
{
    real_one_series<14, PADDED, true, true>(p);
}

// The batched two-sided xCorr (xcorr.go:102-153) at n = 32768 in the same form: pair i = (x_i, y_i), each zero-padded
// in front on its own (any Nx, Ny <= n), ONE pair per workgroup iteration and three 16384-point transforms per pair:
//   ZX = FFT_M(zx) is parked in the workgroup's slice of the context's scratch buffer (256 KB, natural bin order: written and read
//   back by the same CU within one iteration), ZY = FFT_M(zy) stays in registers; at a mirror pair of bins the thread rebuilds
//   2 X[k], 2 X[M-k] from the parked ZX[k], ZX[M-k] (its own bin and the bin at (M - k) mod M: two coalesced 16-byte reads where
//   the xCorrWithX kernel reads the reference's table) and goes on as above with xc := conj X: cc = FFT_n(Y conj X) / n.
// Every series is centred and scaled to O(1) by an exact power of two on its own before its own transform (two_device.h,
// pair_scale): no pair has to be listed and redone.  Round 4's four-step kernel (xcorr_two_sided_long<15>) crossed a 512 KB slice
// four times per pair and read the rows twice (or once, with a redo list): 5 - 6 x the algorithmic bytes; this one 2 x.
template <int LM, bool PADDED>
__device__ __forceinline__ void real_two_sided(const FusedParams &p, const two::PairInv &iv)
{
    using namespace occ4;
    using namespace fold;
    using namespace small;
    using namespace real;
    constexpr int M = 1 << LM, n = 2 * M, S = M / 16, R1 = LM == 14 ? 4 : 2;
    static_assert(LM == 13 || LM == 14, "the 8192- and 16384-point transforms of small_device.h");
    __shared__ double red[112];
    __shared__ double2 g2l[8 * R1];
    __shared__ double2 xbuf[(S / 64) * 544];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j = column_of_lane<LM>(t);
    double2 *const b = xbuf;
    const int padx = PADDED ? n - p.Nx : 0, pady = PADDED ? n - p.N : 0;
    const bool normalize = p.normalize_y != 0;
    const double2 *__restrict__ twm = p.twm;
    const double2 *__restrict__ gs = p.gsmall;
    double2 *const park = p.gscratch + (size_t)blockIdx.x * (size_t)M; // the workgroup's slice: ZX in natural bin order
    typedef d2v __attribute__((address_space(1))) *gd2;
    if (t < 8 * R1)
        g2l[t] = tw_factor<R1>(twm, t % R1, t / R1);
    __syncthreads();
    const long long total = p.npairs;
    for (long long pair = blockIdx.x; pair < total; pair += gridDim.x) {
        double2 v[16];
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        // one series into v: point m = j + i S holds the samples 2m - pad, 2m + 1 - pad (a pad position: 0), d = sample - K; the
        // sums of d and d^2 over the series land in q[qo], q[qo + 1]
        const auto load_series = [&](const double *const r, const int pad, const int qo) __attribute__((always_inline)) {
            const double K = normalize ? scalar_ptr(r)[0] : 0.0;
            int jr = j;
            asm volatile("" : "+v"(jr));
            jr &= S - 1;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int h = 0; h < 2; h++) { // two batches of eight requests
                d2v s8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = 8 * h + k;
                    // (wave-uniform: a request inside the pad is pointed at the row's own first samples, an L2 hit -- or, when the series
                    // is shorter than the 2 S samples such a request spans, at the 2 S samples in FRONT of the row: the allocation's guard
                    // or earlier rows, never past the end of the group's last row)
                    const bool all_pad = PADDED && 2 * (i + 1) * S <= pad;
                    const long long off = all_pad ? (n - pad >= 2 * S ? 0ll : -2ll * S) : 2ll * i * S - pad;
                    typedef d2v __attribute__((aligned(8))) d2u;
                    const d2u s = __builtin_nontemporal_load((gptr<d2u>)scalar_ptr_at(r, off) + (unsigned)jr);
                    s8[k] = d2v{s.x, s.y};
                }
                fence();
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = 8 * h + k;
                    const int e = 2 * (jr + i * S) - pad;
                    const bool v0 = !PADDED || e >= 0, v1 = !PADDED || e + 1 >= 0;
                    const double d0 = v0 ? s8[k].x - K : 0.0, d1 = v1 ? s8[k].y - K : 0.0;
                    v[i] = make_double2(d0, d1);
                    a0 += d0;
                    a1 = fma(d0, d0, a1);
                    a2 += d1;
                    a3 = fma(d1, d1, a3);
                }
            }
            pair_sum4<S>(a0, a1, a2, a3, red, wave);
            q[qo] = uniform(a0 + a2);
            q[qo + 1] = uniform(a1 + a3);
        };
        // v <- (d s - m) at valid positions, the series' own exact power-of-two scale and mean (the numbers pair_scale forms from q)
        const auto scale_series = [&](const int pad, const double sc, const double mean, const bool dead) __attribute__((always_inline)) {
            int jr = j;
            asm volatile("" : "+v"(jr));
            jr &= S - 1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = 2 * (jr + i * S) - pad;
                const bool v0 = !PADDED || e >= 0, v1 = !PADDED || e + 1 >= 0;
                v[i].x = (v0 && !dead) ? fma(v[i].x, sc, -mean) : 0.0;
                v[i].y = (v1 && !dead) ? fma(v[i].y, sc, -mean) : 0.0;
            }
        };
        // ---- x: statistics, scale, ZX = FFT_M(zx), parked
        load_series(p.xrows + pair * p.xstride, padx, 0);
        {
            double qx[4] = {q[0], q[1], 1.0, 1.0}; // (x on its own: pair_scale's numbers for series A do not depend on series B)
            const two::PairScale px = two::pair_scale(qx, iv, normalize);
            const bool deadx = normalize ? (px.nil || px.nan) : px.nan;
            scale_series(padx, px.sA, px.mA, deadx);
        }
        forward<LM>(v, b, g2l, gs, j);
        {
            int jp = j;
            asm volatile("" : "+v"(jp));
            jp &= S - 1;
#pragma unroll
            for (int r = 0; r < 16; r++)
                *((gd2)scalar_ptr_at(park, r * S) + (unsigned)jp) = d2v{v[BR16(r)].x, v[BR16(r)].y};
        }
        // ---- y: statistics, scale, ZY = FFT_M(zy) in registers
        load_series(p.rows + pair * p.stride, pady, 2);
        const two::PairScale ps = two::pair_scale(q, iv, normalize);
        const bool dead = ps.nil || ps.nan;
        scale_series(pady, ps.sB, ps.mB, dead);
        forward<LM>(v, b, g2l, gs, j);
        // ---- mirror pairs: X from the parked ZX, Y from ZY, P = Y conj X, re-tangled
        __syncthreads(); // (every thread's part of ZX is in memory before anybody reads a mirrored bin: a workgroup barrier with the memory fence)
        {
            int jm = j;
            asm volatile("" : "+v"(jm));
            jm &= S - 1;
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)((65536 / n) * jm));
            // the pair's factors: conj of 2 X at the two bins, rebuilt from the parked ZX[k] (own bin) and ZX[(M - k) mod M]
            const auto conj_x = [](const RawPairXC &z, const double2 W) __attribute__((always_inline)) {
                const TwoBins X = spectrum_pair(z.a, z.b, W);
                return TwoBins{make_double2(X.k.x, -X.k.y), make_double2(X.m.x, -X.m.y)};
            };
            mirror_stage<0, MUSE_REAL_AHEAD, RawPairXC, S>(
                v, b, j, wave, Wj, false,
                [&](const int r) __attribute__((always_inline)) {
                    const d2v zk = *((gd2)scalar_ptr_at(park, r * S) + (unsigned)jm);
                    const d2v zq = *((gd2)scalar_ptr(park) + (unsigned)((M - r * S - jm) & (M - 1))); // (column 0 of r = 0: bin 0 itself)
                    return RawPairXC{make_double2(zk.x, zk.y), make_double2(zq.x, zq.y)};
                },
                conj_x,
                [&](const double2 v8) __attribute__((always_inline)) { // bin M / 2 pairs with itself, W = -i
                    const d2v zh = *((gd2)scalar_ptr_at(park, M / 2));
                    const double2 Wh = make_double2(0.0, -1.0);
                    const TwoBins f = conj_x(RawPairXC{make_double2(zh.x, zh.y), make_double2(zh.x, zh.y)}, Wh);
                    return mirror_pair(v8, v8, Wh, f.k, f.m).k;
                });
        }
        natural_order(v);
        forward<LM>(v, b, g2l, gs, j); // 4 n cc[2m] + 4 n i cc[2m+1] (before the pair's factor), m = j + r S, at v[BR16(r)]
        const double fac = ps.fac * (1.0 / (4.0 * n)); // (2 X, 2 Y, and the 1 / n of the inverse transform: exact)
        if (p.cc_out && !dead) {
            double *const cc = p.cc_out + pair * (long long)n;
            int jc = j;
            asm volatile("" : "+v"(jc));
            jc &= S - 1;
#pragma unroll
            for (int r = 0; r < 16; r++)
                *((d2v __attribute__((address_space(1))) *)scalar_ptr_at(cc, 2 * r * S) + (unsigned)jc) = d2v{v[BR16(r)].x * fac, v[BR16(r)].y * fac};
        }
        double sv = 0.0;
        int code = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const double a0 = v[BR16(r)].x, a1 = v[BR16(r)].y;
            const bool g0 = fabs(a0) > fabs(sv);
            sv = g0 ? a0 : sv;
            code = g0 ? 2 * r : code;
            const bool g1 = fabs(a1) > fabs(sv);
            sv = g1 ? a1 : sv;
            code = g1 ? 2 * r + 1 : code;
        }
        const double ma = fabs(sv);
        const int ia = 2 * (j + (code >> 1) * S) + (code & 1);
        const double cc0 = v[0].x;
        double pa = ma, pb = 0.0;
        pair_max2<S>(pa, pb, red, wave);
        int ca = (ma == pa && pa > 0.0) ? ia : 0x7fffffff, cb = 0x7fffffff;
        pair_min_i2<S>(ca, cb, red, wave);
        const bool own = ca == 0x7fffffff ? j == 0 : (ia == ca && ma == pa);
        if (own) {
            const int idx = ca == 0x7fffffff ? 0 : ca;
            double mv = (ca == 0x7fffffff ? cc0 : sv) * fac;
            int lag = idx > n / 2 ? idx - n : idx;
            if (ps.nil) { mv = 0.0; lag = 0; }               // xcorr.go:110-127
            if (ps.nan) { mv = __builtin_nan(""); lag = 0; } // every cc is NaN: maxAbsIndex keeps index 0
            p.mv[pair] = mv;
            p.lag[pair] = lag;
            if (p.nil_out)
                p.nil_out[pair] = ps.nil ? 1 : 0;
        }
        lds_barrier(); // (red is reused by the next pair's statistics)
    }
}
template <bool PADDED>
__global__ __launch_bounds__(1024, 4) void xcorr_two_sided_real32k(const FusedParams p, const two::PairInv iv)
{
    real_two_sided<14, PADDED>(p, iv);
}
template <bool PADDED>
__global__ __launch_bounds__(512, 4) void xcorr_two_sided_real16k(const FusedParams p, const two::PairInv iv)
{
    real_two_sided<13, PADDED>(p, iv);
}

template <bool PADDED>
__global__ void xcorr_two_sided_real8k(const FusedParams p, const two::PairInv iv); // (below, beside xcorr_fused_real8k)

// two-sided xCorr, n = 8192 (the n = 4096 kernel's tables: p.g2, p.g3a, p.g3b), 16384 (p.gsmall: the 8192-point transform's tables) or 32768
// (the 16384-point one's); launch_two_sided's argument
// checks apply; one M-point slice of p.gscratch per workgroup
hipError_t launch_two_sided_real(const FusedParams &p, int num_cus, hipStream_t stream)
{
    if (!p.xrows || !p.rows || !p.twm || !p.gscratch || !p.mv || !p.lag || (p.n != 8192 && p.n != 16384 && p.n != 32768) || p.npairs < 1 || p.Nx < 1 ||
        p.N < 1 || p.Nx > p.n || p.N > p.n || (p.normalize_y && (p.Nx < 2 || p.N < 2)))
        return hipErrorInvalidValue;
    const long long slices = p.gscratch_slices * 2; // (gscratch_slices counts n-point slices; a workgroup parks M = n / 2 points)
    if (p.n == 8192) { // on the n = 4096 kernel's transforms, four 256-thread workgroups per CU
        if (!p.g2 || !p.g3a || !p.g3b)
            return hipErrorInvalidValue;
        const long long grid = std::min<long long>(std::min<long long>(p.npairs, (long long)num_cus * 8), slices);
        if (grid < 1)
            return hipErrorInvalidValue;
        const two::PairInv iv = two::pair_inv(p.Nx, p.N, p.n);
        if (p.Nx < p.n || p.N < p.n)
            hipLaunchKernelGGL(xcorr_two_sided_real8k<true>, dim3((unsigned)grid), dim3(256), 0, stream, p, iv);
        else
            hipLaunchKernelGGL(xcorr_two_sided_real8k<false>, dim3((unsigned)grid), dim3(256), 0, stream, p, iv);
        return hipGetLastError();
    }
    if (!p.gsmall)
        return hipErrorInvalidValue;
    const long long resident = p.n == 16384 ? 2 : 1;
    const long long grid = std::min<long long>(std::min<long long>(p.npairs, (long long)num_cus * resident * 4), slices);
    if (grid < 1)
        return hipErrorInvalidValue;
    const two::PairInv iv = two::pair_inv(p.Nx, p.N, p.n);
    const bool padded = p.Nx < p.n || p.N < p.n;
    if (p.n == 16384) {
        if (padded)
            hipLaunchKernelGGL(xcorr_two_sided_real16k<true>, dim3((unsigned)grid), dim3(512), 0, stream, p, iv);
        else
            hipLaunchKernelGGL(xcorr_two_sided_real16k<false>, dim3((unsigned)grid), dim3(512), 0, stream, p, iv);
    } else if (padded)
        hipLaunchKernelGGL(xcorr_two_sided_real32k<true>, dim3((unsigned)grid), dim3(1024), 0, stream, p, iv);
    else
        hipLaunchKernelGGL(xcorr_two_sided_real32k<false>, dim3((unsigned)grid), dim3(1024), 0, stream, p, iv);
    return hipGetLastError();
}

// n = 65536: one real series per workgroup iteration as TWO passes of the machinery above (M = 32768 complex points do not fit a
// CU; two transforms of H = 16384 points do, one after the other):
//   z[m] = d[2m] + i d[2m+1];   U[m] = z[m] + z[m + H],   V[m] = (z[m] - z[m + H]) W_M^m          (radix 2, decimation in frequency)
//   even bins  Z[2k]   = FFT_H(U)[k]: mirror pairs k <-> (H - k) mod H   (the n = 32768 kernel's stage: MODE 0)
//   odd bins   Z[2k+1] = FFT_H(V)[k]: mirror pairs k <-> H - 1 - k       (MODE 1: no special column)
//   each pass: Y at its bins, P = Y xc, C at its bins (a mirror pair of bins has one parity), then FFT_H of its C:
//   ce = FFT_H(C[2k]),  co = FFT_H(C[2k+1]);   c[m] = ce[m] + W_M^m co[m],   c[m + H] = ce[m] - W_M^m co[m]   (decimation in time)
//   cc[2m'] = Re c[m'],  cc[2m'+1] = Im c[m'].
// V waits for the second pass and ce for the combine in the workgroup's slice of the context's scratch buffer (2 x 256 KB, every
// thread re-reads what it wrote): per series 512 KB of rows and 1 MB of parked values cross the memory side -- 3 x the algorithmic
// bytes where the four-step kernel (xcorr_fused_long<16>: two series per complex transform of n points, four crossings of a 1 MB
// slice per pair) moves 5 x.  The series is transformed UNSCALED as d = x - x[0] (one series per transform: there is no partner
// whose scale it has to match): N = n -- its mean sits in bin 0 alone, which gets the factor 0; N < n -- every lag is corrected by
// -mean c1[lag] (FusedParams::c1, as in the four-step kernel).
template <bool PADDED>
__global__ __launch_bounds__(1024, 4) void xcorr_fused_real64k(const FusedParams p)
{
    using namespace occ4;
    using namespace fold;
    using namespace small;
    using namespace real;
    constexpr int n = 65536, M = n / 2, H = M / 2, LH = 14, S = H / 16;
    __shared__ double red[112];
    __shared__ double2 g2l[8 * 4];
    __shared__ double2 xbuf[16 * 544];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j = column_of_lane<LH>(t);
    double2 *const b = xbuf;
    const int N = PADDED ? p.N : n, pad = PADDED ? n - N : 0;
    const double invN = PADDED ? p.invN : 1.0 / (double)n, invNm1 = PADDED ? p.invNm1 : 1.0 / (double)(n - 1);
    const double2 *__restrict__ twm = p.twm;
    const double2 *__restrict__ gs = p.gsmall;
    const double2 *__restrict__ xc = p.xc;
    double2 *const parkV = p.gscratch + (size_t)blockIdx.x * (size_t)M; // the workgroup's slice: V [i][t], then ce [r][t]
    double2 *const parkE = parkV + H;
    typedef d2v __attribute__((address_space(1))) *gd2;
    if (t < 8 * 4)
        g2l[t] = tw_factor<4>(twm, t % 4, t / 4);
    __syncthreads();
    const long long total = p.M;
#ifdef MUSE_REAL64_STAMPS
    PhaseClock<true> clk; // diagnostic build (tools/ablate/ab_real64_stamps.sh): shader-clock cycles per phase and wave, summed over the series
#else
    PhaseClock<false> clk;
#endif
    clk.start();
    // The row's 32 requests (points m = j + i S and m + H, i = 0 .. 15) in batches of NB + NB, PRE batches in flight: a batch is
    // requested as the one PRE in front of it has been consumed.  (Requested across the loop's back edge -- behind the combine of the
    // previous row, as the other kernels do -- the batches cost 37 - 120 spilled registers: the 64 they occupy meet the peak of the
    // statistics' and the parked values' temporaries.)
    constexpr int PRE = MUSE_REAL64_PRE;
    typedef d2v __attribute__((aligned(8))) d2u;
    constexpr int NB = MUSE_REAL64_BATCH, NH = 16 / NB; // requests per batch and series half, batches
    d2v s1[NH][NB], s2[NH][NB];
    double K = 0.0;
    const auto request = [&](long long row2, const int h, const double after) __attribute__((always_inline)) {
        if (row2 >= total)
            row2 = total - 1; // (nothing left: an L2-hot dummy)
        const double *const rw = p.rows + row2 * p.stride;
        int jr = j;
        asm volatile("" : "+v"(jr) : "v"(after)); // (the requests' addresses hang on `after`: they cannot be hoisted in front of what produced it)
        jr &= S - 1;
        if (h == 0)
            K = scalar_ptr(rw)[0];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = NB * h + k;
            const bool all_pad = PADDED && 2 * (i + 1) * S <= pad; // (wave-uniform: pointed at the row's own first samples, an L2 hit)
            const long long off = all_pad ? 0ll : 2ll * i * S - pad;
            const d2u a = __builtin_nontemporal_load((gptr<d2u>)scalar_ptr_at(rw, off) + (unsigned)jr);
            const d2u c = __builtin_nontemporal_load((gptr<d2u>)scalar_ptr_at(rw, 2ll * (i * S + H) - pad) + (unsigned)jr);
            s1[h][k] = d2v{a.x, a.y};
            s2[h][k] = d2v{c.x, c.y};
        }
    };
    for (long long row = blockIdx.x; row < total; row += gridDim.x) {
        double2 v[16];
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
        for (int h = 0; h < PRE; h++)
            request(row, h, 0.0);
        // ---- the row, once: points m = j + i S (samples 2m - pad, 2m + 1 - pad; pad positions 0) and m + H (always data: pad < n / 2),
        // d = x - K, the statistics, U kept, V = (z1 - z2) W_M^m parked
        {
            int jr = j;
            asm volatile("" : "+v"(jr));
            jr &= S - 1;
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)(2 * jr)); // W_M^j = W_65536^(2 j)
#pragma unroll
            for (int h = 0; h < NH; h++) {
                if (PRE == 0) {
                    request(row, h, q3);
                    fence();
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int i = NB * h + k;
                    const int e = 2 * (jr + i * S) - pad;
                    const bool v0 = !PADDED || e >= 0, v1 = !PADDED || e + 1 >= 0;
                    const double a0 = v0 ? s1[h][k].x - K : 0.0, a1 = v1 ? s1[h][k].y - K : 0.0;
                    const double c0 = s2[h][k].x - K, c1 = s2[h][k].y - K;
                    q0 += a0 + c0;
                    q1 = fma(a0, a0, fma(c0, c0, q1));
                    q2 += a1 + c1;
                    q3 = fma(a1, a1, fma(c1, c1, q3));
                    v[i] = make_double2(a0 + c0, a1 + c1);
                    const double2 Vt = cmul(make_double2(a0 - c0, a1 - c1), i == 0 ? Wj : cmul(Wj, w32(i))); // W_M^(j + i S) = W_M^j W_32^i
                    *((gd2)scalar_ptr_at(parkV, i * S) + (unsigned)(t & (S - 1))) = d2v{Vt.x, Vt.y};
                }
                fence();
                if (PRE > 0 && h + PRE < NH) // (behind the batch just consumed: its registers are free)
                    request(row, h + PRE, q3);
                fence();
            }
        }
        clk.template stamp<0>(); // rows requested and consumed, V parked (stores issued)
        pair_sum4<S>(q0, q1, q2, q3, red, wave);
        bool zero, nan;
        const double var = uniform(variance(Stat{q0 + q2, q1 + q3}, invN, invNm1, zero, nan));
        const double mean = uniform((q0 + q2) * invN);
        clk.template stamp<1>(); // statistics reduced
        // ---- pass A: even bins
        forward<LH>(v, b, g2l, gs, j);
        clk.template stamp<2>();
        {
            int jm = j;
            asm volatile("" : "+v"(jm));
            jm &= S - 1;
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)(2 * jm)); // W_n^(2 j)
            mirror_stage<0, MUSE_REAL_AHEAD, RawPairXC>(
                v, b, j, wave, Wj, !PADDED,
                [&](const int r) __attribute__((always_inline)) { // xc at bins 2 (j + r S) and M - 2 (j + r S)
                    return RawPairXC{ldg2u(scalar_ptr_at(xc, 2 * r * S), (unsigned)(2 * jm)),
                                     ldg2u(scalar_ptr_at(xc, M - 2 * r * S - 2 * S), (unsigned)(2 * (S - jm)))};
                },
                [&](const RawPairXC &x, const double2) __attribute__((always_inline)) { return TwoBins{x.a, x.b}; },
                [&](const double2 v8) __attribute__((always_inline)) { // bin M / 2 of the series' spectrum: pairs with itself, W = -i
                    const double2 xh = ldg2u(scalar_ptr_at(xc, M / 2), 0u);
                    return mirror_pair(v8, v8, make_double2(0.0, -1.0), xh, xh).k;
                });
        }
        clk.template stamp<3>(); // mirror stage A
        natural_order(v);
        forward<LH>(v, b, g2l, gs, j); // ce[j + r S] at v[BR16(r)]
        clk.template stamp<4>();
#pragma unroll
        for (int r = 0; r < 16; r++)
            *((gd2)scalar_ptr_at(parkE, r * S) + (unsigned)(t & (S - 1))) = d2v{v[BR16(r)].x, v[BR16(r)].y};
        // ---- pass B: odd bins
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const d2v z = *((gd2)scalar_ptr_at(parkV, i * S) + (unsigned)(t & (S - 1)));
            v[i] = make_double2(z.x, z.y);
        }
#ifdef MUSE_REAL64_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (diagnostic: the read-back of V is charged to its own phase)
#endif
        clk.template stamp<5>(); // ce parked (stores issued), V read back
        forward<LH>(v, b, g2l, gs, j);
        clk.template stamp<6>();
        {
            int jm = j;
            asm volatile("" : "+v"(jm));
            jm &= S - 1;
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)(2 * jm + 1)); // W_n^(2 j + 1)
            mirror_stage<1, MUSE_REAL_AHEAD, RawPairXC>(
                v, b, j, wave, Wj, false,
                [&](const int r) __attribute__((always_inline)) { // xc at bins 2 (j + r S) + 1 and M - 2 (j + r S) - 1
                    return RawPairXC{ldg2u(scalar_ptr_at(xc, 2 * r * S + 1), (unsigned)(2 * jm)),
                                     ldg2u(scalar_ptr_at(xc, M - 2 * r * S - 2 * S + 1), (unsigned)(2 * (S - 1 - jm)))};
                },
                [&](const RawPairXC &x, const double2) __attribute__((always_inline)) { return TwoBins{x.a, x.b}; },
                [&](const double2) __attribute__((always_inline)) { return make_double2(0.0, 0.0); });
        }
        clk.template stamp<7>(); // mirror stage B
        natural_order(v);
        forward<LH>(v, b, g2l, gs, j); // co[j + r S] at v[BR16(r)]
        clk.template stamp<8>();
        // ---- c[m] = ce + W_M^m co, c[m + H] = ce - W_M^m co; lags 2m, 2m + 1 (lower half) and 2 (m + H), 2 (m + H) + 1 (upper half);
        // maxAbsIndex (xcorr.go:39-50): per half ascending r = ascending lag, the lower half first
        double sl = 0.0, su = 0.0, cc0 = 0.0;
        int cl = 0, cu = 0;
        {
            int jc = j;
            asm volatile("" : "+v"(jc));
            jc &= S - 1;
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)(2 * jc)); // W_M^j
            const double m2 = 2.0 * mean;                                  // (the re-tangled spectrum carries 2 C)
            constexpr int CB = PADDED ? 2 : 4; // values per batch of requests (N < n: the correction table's entries travel with them)
#pragma unroll
            for (int h = 0; h < 16 / CB; h++) {
                d2v e4[CB], ka[CB], kb[CB];
#pragma unroll
                for (int k = 0; k < CB; k++) {
                    const int r = CB * h + k;
                    e4[k] = *((gd2)scalar_ptr_at(parkE, r * S) + (unsigned)(t & (S - 1)));
                    if (PADDED) { // c1 at lags 2 (j + r S) (+1) and 2 (j + r S + H) (+1)
                        ka[k] = ((gptr<d2v>)scalar_ptr_at(p.c1, 2 * r * S))[(unsigned)jc];
                        kb[k] = ((gptr<d2v>)scalar_ptr_at(p.c1, 2 * (r * S + H)))[(unsigned)jc];
                    }
                }
                fence();
#pragma unroll
                for (int k = 0; k < CB; k++) {
                    const int r = CB * h + k;
                    const double2 o = cmul(v[BR16(r)], r == 0 ? Wj : cmul(Wj, w32(r)));
                    double l0 = e4[k].x + o.x, l1 = e4[k].y + o.y, u0 = e4[k].x - o.x, u1 = e4[k].y - o.y;
                    if (PADDED) { // cc(d - mean 1_valid) = cc(d) - mean c1
                        l0 = fma(-m2, ka[k].x, l0);
                        l1 = fma(-m2, ka[k].y, l1);
                        u0 = fma(-m2, kb[k].x, u0);
                        u1 = fma(-m2, kb[k].y, u1);
                    }
                    if (r == 0)
                        cc0 = l0; // (column 0: cc[0], reported when nothing is above 0)
                    bool g = fabs(l0) > fabs(sl);
                    sl = g ? l0 : sl;
                    cl = g ? 2 * r : cl;
                    g = fabs(l1) > fabs(sl);
                    sl = g ? l1 : sl;
                    cl = g ? 2 * r + 1 : cl;
                    g = fabs(u0) > fabs(su);
                    su = g ? u0 : su;
                    cu = g ? 2 * r : cu;
                    g = fabs(u1) > fabs(su);
                    su = g ? u1 : su;
                    cu = g ? 2 * r + 1 : cu;
                }
            }
        }
        clk.template stamp<9>(); // ce read back, combine, per-thread maximum
        const bool up = fabs(su) > fabs(sl); // (the lower half holds the lower lags: it keeps ties)
        const double sv = up ? su : sl;
        const int code = up ? cu : cl;
        const double ma = fabs(sv);
        const int ia = 2 * (j + (code >> 1) * S + (up ? H : 0)) + (code & 1);
        double pa = ma, pb = 0.0;
        pair_max2<S>(pa, pb, red, wave);
        int ca = (ma == pa && pa > 0.0) ? ia : 0x7fffffff, cb = 0x7fffffff;
        pair_min_i2<S>(ca, cb, red, wave);
        const bool own = ca == 0x7fffffff ? j == 0 : (ia == ca && ma == pa);
        if (own) {
            double y = __builtin_amdgcn_rsq(var);
            y = y * fma(-0.5 * var * y, y, 1.5);
            y = y * fma(-0.5 * var * y, y, 1.5);
            double mv = (ca == 0x7fffffff ? cc0 : sv) * (0.5 * y);
            const int idx = ca == 0x7fffffff ? 0 : ca;
            int lag = idx > n / 2 ? idx - n : idx;
            if (zero) { mv = 0.0; lag = 0; }               // xcorr.go:166-167
            if (nan) { mv = __builtin_nan(""); lag = 0; }
            p.mv[row] = mv;
            p.lag[row] = lag;
        }
        __syncthreads(); // (the slice and red are reused by the next row)
        clk.template stamp<10>(); // workgroup maximum, result, closing barrier
    }
#ifdef MUSE_REAL64_STAMPS
    if (p.dbg && (t & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NPHASE; i++)
            p.dbg[((long long)blockIdx.x * 16 + wave) * NPHASE + i] = clk.acc[i];
    }
#endif
}

// n = 8192: one real series per 256-thread workgroup as a real transform on the n = 4096 kernel's complex machinery
// (foldk_device.h: three radix-16 passes with the twiddles folded into the butterflies, half-round transposes through 34.8 KB, 128
// registers -> four workgroups per CU).  That machinery runs two packed series of 4096 points at 0.45 of the roofline where the
// n = 8192 kernel of xcorr_small.hip (512 threads, four passes, 24 workgroup barriers per pair) runs at 0.28; a real series of 8192
// points is 4096 complex points.  After the first transform thread t = 16 hi + lo holds Z[c + 256 k3] at v[BR16(k3)] with the COLUMN
// c = hi + 16 lo: bin f's partner M - f = (256 - c) + 256 (15 - k3) is column 256 - c, register 15 - k3 -- mirror_stage with S = 256,
// through the same 8 x 272 buffer the transposes use.  The reference's spectrum at the thread's bins comes lane-ordered
// (FusedParams::xcp: [r][t] = xc[c(t) + 256 r], [8 + r][t] = xc[M - c(t) - 256 r]; built per batch by launch_real8k_tables): read in
// bin order the same values are 64 different cache lines per wave instruction.  The second transform takes C where the mirror stage
// leaves it (its plain first stage pairs the registers (r, r + 1)): no renaming.
template <bool PADDED>
__global__ __launch_bounds__(256, 4) void xcorr_fused_real8k(const FusedParams p)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    using namespace real;
    constexpr int n = 8192, M = n / 2, S = 256;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[24];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + XW * wave;
    const int N = PADDED ? p.N : n, pad = PADDED ? n - N : 0;
    const double invN = PADDED ? p.invN : 1.0 / (double)n, invNm1 = PADDED ? p.invNm1 : 1.0 / (double)(n - 1);
    const double2 *__restrict__ twm = p.twm;
    const double2 *__restrict__ xcp = p.xcp;
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();
    const long long total = p.M;
    // point m = t + 256 i of z holds the samples 2m - pad, 2m + 1 - pad (a pad position: 0): one 16-byte request per point
    double x0[16], x1[16], K;
    const auto request = [&](long long row) __attribute__((always_inline)) {
        if (row >= total)
            row = total - 1; // (nothing left: an L2-hot dummy)
        const double *const r = p.rows + row * p.stride;
        int tr = t;
        asm volatile("" : "+v"(tr)); // (offsets derived per request, not hoisted)
        tr &= S - 1;
        K = scalar_ptr(r)[0];
        typedef d2v __attribute__((aligned(8))) d2u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool all_pad = PADDED && i < 8 && 2 * (i + 1) * S <= pad; // (wave-uniform: pointed at the row's own first samples)
            const long long off = all_pad ? 0ll : 2ll * i * S - pad;
            const d2u s = __builtin_nontemporal_load((gptr<d2u>)scalar_ptr_at(r, off) + (unsigned)tr);
            x0[i] = s.x;
            x1[i] = s.y;
        }
    };
    if (blockIdx.x < total)
        request(blockIdx.x);
    for (long long row = blockIdx.x; row < total; row += gridDim.x) {
        double2 v[16];
        int ts = t;
        asm volatile("" : "+v"(ts));
        ts &= S - 1;
        // d = x - K with K the first sample, shifted statistics (xcorr.go:84-95); the series goes into the transform centred and at
        // O(1) (an exact power-of-two scale close to 1 / sigma).  (Transformed unscaled, with the mean left in bin 0 / corrected by
        // the indicator table behind the second transform as the n = 65536 kernel does, the two statistics barriers go -- and nothing
        // is gained: four workgroups per CU hide them, the kernel is bound by its fp64 issue; the padded build loses 3 - 6 % to the
        // table's requests.  profiles/r05_real_transform.txt)
        double q0 = 0.0, q1 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = 2 * (ts + i * S) - pad;
            const bool v0 = !PADDED || i >= 8 || e >= 0, v1 = !PADDED || i >= 8 || e + 1 >= 0; // (pad < n / 2: the upper half is data)
            const double d0 = v0 ? x0[i] - K : 0.0, d1 = v1 ? x1[i] - K : 0.0;
            v[i] = make_double2(d0, d1);
            q0 += d0 + d1;
            q1 = fma(d0, d0, fma(d1, d1, q1));
        }
        q0 = wave_sum_dpp(q0);
        q1 = wave_sum_dpp(q1);
        lds_barrier(); // (red's readers of the previous row are done)
        if (lane == 0) {
            red[2 * wave] = q0;
            red[2 * wave + 1] = q1;
        }
        lds_barrier();
        q0 = uniform((red[0] + red[2]) + (red[4] + red[6]));
        q1 = uniform((red[1] + red[3]) + (red[5] + red[7]));
        bool zero, nan;
        const double var0 = variance(Stat{q0, q1}, invN, invNm1, zero, nan);
        const bool dead = zero || nan;
        const double sc = dead ? 1.0 : pow2_inv_sigma(var0);
        const double var = uniform(var0 * sc * sc);
        const double mean = q0 * invN * sc;
        asm volatile("" : "+v"(ts));
        ts &= S - 1;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = 2 * (ts + i * S) - pad;
            const bool v0 = !PADDED || i >= 8 || e >= 0, v1 = !PADDED || i >= 8 || e + 1 >= 0;
            v[i].x = (v0 && !dead) ? fma(v[i].x, sc, -mean) : 0.0;
            v[i].y = (v1 && !dead) ? fma(v[i].y, sc, -mean) : 0.0;
        }
        // ---- Z = FFT_M(z): Z[hi + 16 lo + 256 k3] at v[BR16(k3)] (as xcorr_fused_n4096_fold)
        dft16_nr(v);
        exchange_cross<0, 1, true>(v, xbuf, wave, t);
        gdft16_nr(v, G2Fetch{g2s, t >> 4});
        exchange_local<1>(v, xw, t);
        gdft16_nr_l2(v, G3Derived(p.g3a, t));
        // ---- mirror pairs on the columns c = hi + 16 lo
        {
            int tm = t;
            asm volatile("" : "+v"(tm));
            tm &= S - 1;
            const int c = (tm >> 4) + 16 * (tm & 15);
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)(8 * c)); // W_8192^c = W_65536^(8 c); bin c + 256 r: times W_32^r
            mirror_stage<0, MUSE_REAL_AHEAD, RawPairXC, S>(
                v, xbuf, c, wave, Wj, false,
                [&](const int r) __attribute__((always_inline)) {
                    return RawPairXC{ldg2u(scalar_ptr_at(xcp, r * S), (unsigned)tm), ldg2u(scalar_ptr_at(xcp, (8 + r) * S), (unsigned)tm)};
                },
                [&](const RawPairXC &x, const double2) __attribute__((always_inline)) { return TwoBins{x.a, x.b}; },
                [&](const double2 v8) __attribute__((always_inline)) { // bin M / 2: pairs with itself, W = -i
                    const double2 xh = ldg2u(scalar_ptr_at(p.xc, M / 2), 0u);
                    return mirror_pair(v8, v8, make_double2(0.0, -1.0), xh, xh).k;
                });
            lds_barrier(); // (the next use of the buffer is a wave-local transpose into a quarter other waves' columns live in)
        }
        // ---- c = FFT_M(C): the plain first stage pairs the registers (r, r + 1)
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            bf_one(v[r], v[r + 1]);
        dft16_rn_s234(v);
        exchange_local<0>(v, xw, t);
        gdft16_nr(v, G2Fetch{g2s, t & 15});
        exchange_cross<1, 1>(v, xbuf, wave, t);
        gdft16_nr_l2(v, G3Derived(p.g3b, t)); // 2 cc[2m] + 2 i cc[2m+1], m = t + 256 m3, at v[BR16(m3)]
        double sv = 0.0;
        int code = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const double a0 = v[BR16(r)].x, a1 = v[BR16(r)].y;
            const bool g0 = fabs(a0) > fabs(sv);
            sv = g0 ? a0 : sv;
            code = g0 ? 2 * r : code;
            const bool g1 = fabs(a1) > fabs(sv);
            sv = g1 ? a1 : sv;
            code = g1 ? 2 * r + 1 : code;
        }
        const double ma = fabs(sv);
        const int ia = 2 * (t + (code >> 1) * S) + (code & 1);
        const double cc0 = v[0].x; // (thread 0: cc[0], reported when nothing is above 0)
        fence();
        request(row + gridDim.x); // the next row: in flight during the reductions and the result write-out
        fence();
        const double wa = wave_max_nonneg(ma);
        if (lane == 0)
            red[8 + wave] = wa;
        lds_barrier();
        const double pa = fmax(fmax(red[8], red[9]), fmax(red[10], red[11]));
        const int wi = wave_min_i_dpp((ma == pa && pa > 0.0) ? ia : 0x7fffffff);
        if (lane == 0)
            ((int *)(red + 12))[wave] = wi;
        lds_barrier();
        const int *ri = (const int *)(red + 12);
        const int ca = min(min(ri[0], ri[1]), min(ri[2], ri[3]));
        const bool own = ca == 0x7fffffff ? t == 0 : (ia == ca && ma == pa);
        if (own) {
            double y = __builtin_amdgcn_rsq(var);
            y = y * fma(-0.5 * var * y, y, 1.5);
            y = y * fma(-0.5 * var * y, y, 1.5);
            double mv = (ca == 0x7fffffff ? cc0 : sv) * (0.5 * y); // (the re-tangled spectrum carries 2 C)
            const int idx = ca == 0x7fffffff ? 0 : ca;
            int lag = idx > n / 2 ? idx - n : idx;
            if (zero) { mv = 0.0; lag = 0; }               // xcorr.go:166-167
            if (nan) { mv = __builtin_nan(""); lag = 0; }
            p.mv[row] = mv;
            p.lag[row] = lag;
        }
    }
}
// The batched two-sided xCorr at n = 8192 in the form of real_two_sided above, on the n = 4096 kernel's transforms: pair i = (x_i, y_i),
// each zero-padded in front on its own, ONE pair per 256-thread workgroup iteration (four workgroups per CU) and three 4096-point
// transforms per pair where the pair-packed kernel (xcorr_two_sided_small<13>: 512 threads, four passes) runs two of 8192 points:
// per lane 2 200 vector instructions and 7 x 32 LDS operations in 4 waves against 1 400 and 6 x 32 in 8.  ZX is parked in the
// workgroup's slice of the scratch buffer in THREAD order -- park[256 k3 + t] = ZX[c(t) + 256 k3], c(t) = (t >> 4) + 16 (t & 15) the
// thread's column: coalesced stores, coalesced loads of a thread's own bins; the mirror bin (M - k) mod M = c' + 256 k3' sits at
// 256 k3' + tcol(c'), tcol(c) = 16 (c & 15) + (c >> 4): sixteen lanes of one hi read sixteen consecutive entries backwards.
template <bool PADDED>
__global__ __launch_bounds__(256, 4) void xcorr_two_sided_real8k(const FusedParams p, const two::PairInv iv)
{
    using namespace occ4;
    using namespace fold;
    using namespace foldk;
    using namespace real;
    constexpr int n = 8192, M = n / 2, S = 256;
    __shared__ double2 xbuf[OCC_XBUF];
    __shared__ double2 g2s[128];
    __shared__ double red[24];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double2 *const xw = xbuf + XW * wave;
    const int padx = PADDED ? n - p.Nx : 0, pady = PADDED ? n - p.N : 0;
    const bool normalize = p.normalize_y != 0;
    const double2 *__restrict__ twm = p.twm;
    double2 *const park = p.gscratch + (size_t)blockIdx.x * (size_t)M;
    typedef d2v __attribute__((address_space(1))) *gd2;
    if (t < 128)
        g2s[t] = p.g2[t];
    __syncthreads();
    const long long total = p.npairs;
    for (long long pair = blockIdx.x; pair < total; pair += gridDim.x) {
        double2 v[16];
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        // one series into v: point m = t + 256 i holds the samples 2m - pad, 2m + 1 - pad (a pad position: 0), d = sample - K; the sums
        // of d and d^2 over the series land in q[qo], q[qo + 1]
        const auto load_series = [&](const double *const r, const int pad, const int qo) __attribute__((always_inline)) {
            const double K = normalize ? scalar_ptr(r)[0] : 0.0;
            int tr = t;
            asm volatile("" : "+v"(tr));
            tr &= S - 1;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int h = 0; h < 2; h++) { // two batches of eight requests
                d2v s8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = 8 * h + k;
                    // (a request inside the pad: the row's own first samples, an L2 hit -- or the 2 S samples in front of a row shorter
                    // than that: guard or earlier rows, never past the group's last row)
                    const bool all_pad = PADDED && 2 * (i + 1) * S <= pad;
                    const long long off = all_pad ? (n - pad >= 2 * S ? 0ll : -2ll * S) : 2ll * i * S - pad;
                    typedef d2v __attribute__((aligned(8))) d2u;
                    const d2u s = __builtin_nontemporal_load((gptr<d2u>)scalar_ptr_at(r, off) + (unsigned)tr);
                    s8[k] = d2v{s.x, s.y};
                }
                fence();
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = 8 * h + k;
                    const int e = 2 * (tr + i * S) - pad;
                    const bool v0 = !PADDED || e >= 0, v1 = !PADDED || e + 1 >= 0;
                    const double d0 = v0 ? s8[k].x - K : 0.0, d1 = v1 ? s8[k].y - K : 0.0;
                    v[i] = make_double2(d0, d1);
                    a0 += d0 + d1;
                    a1 = fma(d0, d0, fma(d1, d1, a1));
                }
            }
            a0 = wave_sum_dpp(a0);
            a1 = wave_sum_dpp(a1);
            lds_barrier(); // (red's previous readers are done)
            if (lane == 0) {
                red[2 * wave] = a0;
                red[2 * wave + 1] = a1;
            }
            lds_barrier();
            q[qo] = uniform((red[0] + red[2]) + (red[4] + red[6]));
            q[qo + 1] = uniform((red[1] + red[3]) + (red[5] + red[7]));
        };
        const auto scale_series = [&](const int pad, const double sc, const double mean, const bool dead) __attribute__((always_inline)) {
            int tr = t;
            asm volatile("" : "+v"(tr));
            tr &= S - 1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = 2 * (tr + i * S) - pad;
                const bool v0 = !PADDED || e >= 0, v1 = !PADDED || e + 1 >= 0;
                v[i].x = (v0 && !dead) ? fma(v[i].x, sc, -mean) : 0.0;
                v[i].y = (v1 && !dead) ? fma(v[i].y, sc, -mean) : 0.0;
            }
        };
        // Z = FFT_M(z): Z[c + 256 k3] at v[BR16(k3)], c = hi + 16 lo (as xcorr_fused_n4096_fold)
        const auto first_transform = [&]() __attribute__((always_inline)) {
            dft16_nr(v);
            exchange_cross<0, 1, true>(v, xbuf, wave, t);
            gdft16_nr(v, G2Fetch{g2s, t >> 4});
            exchange_local<1>(v, xw, t);
            gdft16_nr_l2(v, G3Derived(p.g3a, t));
        };
        // ---- x: statistics, scale, ZX = FFT_M(zx), parked
        load_series(p.xrows + pair * p.xstride, padx, 0);
        {
            double qx[4] = {q[0], q[1], 1.0, 1.0}; // (x on its own: pair_scale's numbers for series A do not depend on series B)
            const two::PairScale px = two::pair_scale(qx, iv, normalize);
            const bool deadx = normalize ? (px.nil || px.nan) : px.nan;
            scale_series(padx, px.sA, px.mA, deadx);
        }
        first_transform();
        {
            int tp = t;
            asm volatile("" : "+v"(tp));
            tp &= S - 1;
#pragma unroll
            for (int r = 0; r < 16; r++)
                *((gd2)scalar_ptr_at(park, r * S) + (unsigned)tp) = d2v{v[BR16(r)].x, v[BR16(r)].y};
        }
        // ---- y: statistics, scale, ZY = FFT_M(zy) in registers
        load_series(p.rows + pair * p.stride, pady, 2);
        const two::PairScale ps = two::pair_scale(q, iv, normalize);
        const bool dead = ps.nil || ps.nan;
        scale_series(pady, ps.sB, ps.mB, dead);
        first_transform();
        // ---- mirror pairs: X from the parked ZX, Y from ZY, P = Y conj X, re-tangled
        __syncthreads(); // (every thread's part of ZX is in memory before anybody reads a mirrored bin)
        {
            int tm = t;
            asm volatile("" : "+v"(tm));
            tm &= S - 1;
            const int c = (tm >> 4) + 16 * (tm & 15);
            const double2 Wj = ldg2u(scalar_ptr(twm), (unsigned)(8 * c)); // W_8192^c = W_65536^(8 c); bin c + 256 r: times W_32^r
            const auto conj_x = [](const RawPairXC &z, const double2 W) __attribute__((always_inline)) {
                const TwoBins X = spectrum_pair(z.a, z.b, W);
                return TwoBins{make_double2(X.k.x, -X.k.y), make_double2(X.m.x, -X.m.y)};
            };
            mirror_stage<0, MUSE_REAL_AHEAD, RawPairXC, S>(
                v, xbuf, c, wave, Wj, false,
                [&](const int r) __attribute__((always_inline)) {
                    const d2v zk = *((gd2)scalar_ptr_at(park, r * S) + (unsigned)tm);
                    const int bm = (M - r * S - c) & (M - 1), cm = bm & (S - 1);   // the mirror bin (column 0 of r = 0: bin 0 itself)
                    const d2v zq = *((gd2)scalar_ptr(park) + (unsigned)((bm & ~(S - 1)) + 16 * (cm & 15) + (cm >> 4)));
                    return RawPairXC{make_double2(zk.x, zk.y), make_double2(zq.x, zq.y)};
                },
                conj_x,
                [&](const double2 v8) __attribute__((always_inline)) { // bin M / 2 (column 0, register 8) pairs with itself, W = -i
                    const d2v zh = *((gd2)scalar_ptr_at(park, 8 * S));
                    const double2 Wh = make_double2(0.0, -1.0);
                    const TwoBins f = conj_x(RawPairXC{make_double2(zh.x, zh.y), make_double2(zh.x, zh.y)}, Wh);
                    return mirror_pair(v8, v8, Wh, f.k, f.m).k;
                });
            lds_barrier(); // (the next use of the buffer is a wave-local transpose into a quarter other waves' columns live in)
        }
        // ---- c = FFT_M(C): 4 n cc[2m] + 4 n i cc[2m+1] (before the pair's factor), m = t + 256 m3, at v[BR16(m3)]
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            bf_one(v[r], v[r + 1]);
        dft16_rn_s234(v);
        exchange_local<0>(v, xw, t);
        gdft16_nr(v, G2Fetch{g2s, t & 15});
        exchange_cross<1, 1>(v, xbuf, wave, t);
        gdft16_nr_l2(v, G3Derived(p.g3b, t));
        const double fac = ps.fac * (1.0 / (4.0 * n)); // (2 X, 2 Y, and the 1 / n of the inverse transform: exact)
        if (p.cc_out && !dead) {
            double *const cc = p.cc_out + pair * (long long)n;
            int tc = t;
            asm volatile("" : "+v"(tc));
            tc &= S - 1;
#pragma unroll
            for (int r = 0; r < 16; r++)
                *((d2v __attribute__((address_space(1))) *)scalar_ptr_at(cc, 2 * r * S) + (unsigned)tc) = d2v{v[BR16(r)].x * fac, v[BR16(r)].y * fac};
        }
        double sv = 0.0;
        int code = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const double a0 = v[BR16(r)].x, a1 = v[BR16(r)].y;
            const bool g0 = fabs(a0) > fabs(sv);
            sv = g0 ? a0 : sv;
            code = g0 ? 2 * r : code;
            const bool g1 = fabs(a1) > fabs(sv);
            sv = g1 ? a1 : sv;
            code = g1 ? 2 * r + 1 : code;
        }
        const double ma = fabs(sv);
        const int ia = 2 * (t + (code >> 1) * S) + (code & 1);
        const double cc0 = v[0].x;
        const double wa = wave_max_nonneg(ma);
        if (lane == 0)
            red[8 + wave] = wa;
        lds_barrier();
        const double pa = fmax(fmax(red[8], red[9]), fmax(red[10], red[11]));
        const int wi = wave_min_i_dpp((ma == pa && pa > 0.0) ? ia : 0x7fffffff);
        if (lane == 0)
            ((int *)(red + 12))[wave] = wi;
        lds_barrier();
        const int *ri = (const int *)(red + 12);
        const int ca = min(min(ri[0], ri[1]), min(ri[2], ri[3]));
        const bool own = ca == 0x7fffffff ? t == 0 : (ia == ca && ma == pa);
        if (own) {
            const int idx = ca == 0x7fffffff ? 0 : ca;
            double mv = (ca == 0x7fffffff ? cc0 : sv) * fac;
            int lag = idx > n / 2 ? idx - n : idx;
            if (ps.nil) { mv = 0.0; lag = 0; }               // xcorr.go:110-127
            if (ps.nan) { mv = __builtin_nan(""); lag = 0; } // every cc is NaN: maxAbsIndex keeps index 0
            p.mv[pair] = mv;
            p.lag[pair] = lag;
            if (p.nil_out)
                p.nil_out[pair] = ps.nil ? 1 : 0;
        }
    }
}

// the reference's spectrum at the bins of xcorr_fused_real8k's threads, lane-ordered: out[r][t] = xc[c(t) + 256 r],
// out[8 + r][t] = xc[4096 - c(t) - 256 r], r = 0 .. 7, c(t) = (t >> 4) + 16 (t & 15)
__global__ void real8k_tables_kernel(const double2 *__restrict__ xc, double2 *__restrict__ out)
{
    const int t = threadIdx.x, r = blockIdx.x;
    const int c = (t >> 4) + 16 * (t & 15);
    out[r * 256 + t] = xc[c + 256 * r];
    out[(8 + r) * 256 + t] = xc[4096 - c - 256 * r];
}
// FusedParams::xcw (n = 32768 on the 16 x 1024 split): out[r 1024 + j] = xc[k], out[8192 + r 1024 + j] = xc[16384 - k],
// k = (j >> 6) + 16 (j & 63) + 1024 r, r < 8 -- the bins of thread j's lower eight registers and their mirror bins, one coalesced row per r
__global__ void real_split_tables_kernel(const double2 *__restrict__ xc, double2 *__restrict__ out)
{
    const int r = blockIdx.x, j = threadIdx.x;
    const int k = (j >> 6) + 16 * (j & 63) + 1024 * r;
    out[r * 1024 + j] = xc[k];
    out[8192 + r * 1024 + j] = xc[16384 - k];
    if (r == 0 && j == 0)
        out[16384] = xc[8192]; // the reference's bin M / 2 (it pairs with itself)
}
hipError_t launch_real_split_tables(const double2 *xc, double2 *out, hipStream_t stream)
{
    hipLaunchKernelGGL(real_split_tables_kernel, dim3(8), dim3(1024), 0, stream, xc, out);
    return hipGetLastError();
}
hipError_t launch_real8k_tables(const double2 *xc, double2 *out, hipStream_t stream)
{
    hipLaunchKernelGGL(real8k_tables_kernel, dim3(8), dim3(256), 0, stream, xc, out);
    return hipGetLastError();
}

// n = 8192 ... 65536, float64 rows, every row (no pair list); N in (n / 2, n]; p.gsmall = the tables of the complex transform the length runs on
// (n = 16384: 8192 points; n = 32768, 65536: 16384 points), p.xc all n bins
hipError_t launch_fused_real(const FusedParams &p_in, int num_cus, hipStream_t stream)
{
    const FusedParams p = with_reciprocals(p_in);
    if (p.n == 8192) { // one real series per 256-thread workgroup on the n = 4096 kernel's transforms; p.xcp: launch_real8k_tables
        if (!p.rows || !p.twm || !p.xc || !p.xcp || !p.g2 || !p.g3a || !p.g3b || !p.mv || !p.lag || p.N > p.n || 2 * p.N <= p.n || p.pair_list || p.R > 1)
            return hipErrorInvalidValue;
        const long long grid = std::min<long long>(p.M, (long long)num_cus * 4 * 4);
        if (p.N < p.n)
            hipLaunchKernelGGL(xcorr_fused_real8k<true>, dim3((unsigned)grid), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL(xcorr_fused_real8k<false>, dim3((unsigned)grid), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if (!p.rows || !p.twm || !p.xc || !p.gsmall || !p.mv || !p.lag || (p.n != 16384 && p.n != 32768 && p.n != 65536) || p.N > p.n || 2 * p.N <= p.n || p.pair_list || p.R > 1)
        return hipErrorInvalidValue;
    if (p.n == 16384) { // p.gsmall: the 8192-point transform's tables; two 512-thread workgroups per CU
        const long long grid = std::min<long long>(p.M, (long long)num_cus * 2 * 8);
        if (p.N < p.n)
            hipLaunchKernelGGL(xcorr_fused_real16k<true>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        else
            hipLaunchKernelGGL(xcorr_fused_real16k<false>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        return hipGetLastError();
    }
    if (p.n == 65536) { // two passes per series: the workgroup's slice of the scratch buffer holds M = n / 2 complex points
        if (!p.gscratch || (p.N < p.n && !p.c1))
            return hipErrorInvalidValue;
        const long long grid = std::min<long long>(std::min<long long>(p.M, (long long)num_cus * 4), p.gscratch_slices * 2);
        if (grid < 1)
            return hipErrorInvalidValue;
        if (p.N < p.n)
            hipLaunchKernelGGL(xcorr_fused_real64k<true>, dim3((unsigned)grid), dim3(1024), 0, stream, p);
        else
            hipLaunchKernelGGL(xcorr_fused_real64k<false>, dim3((unsigned)grid), dim3(1024), 0, stream, p);
        return hipGetLastError();
    }
    const long long grid = std::min<long long>(p.M, (long long)num_cus * 8);
    if (p.N < p.n)
        hipLaunchKernelGGL(xcorr_fused_real32k<true>, dim3((unsigned)grid), dim3(1024), 0, stream, p);
    else
        hipLaunchKernelGGL(xcorr_fused_real32k<false>, dim3((unsigned)grid), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

// n = 32768 with the 16384-point transforms as 16 x 1024 (launch_fused_real's conditions; p.gsmall_b, p.wsplit, p.xcw as FusedParams says)
hipError_t launch_fused_real_split(const FusedParams &p_in, int num_cus, hipStream_t stream)
{
    const FusedParams p = with_reciprocals(p_in);
    if (!p.rows || !p.twm || !p.gsmall || !p.gsmall_b || !p.wsplit || p.n != 32768 || p.N > p.n || 2 * p.N <= p.n || p.pair_list)
        return hipErrorInvalidValue;
    if (p.R > 1) { // many references in one pass: p.xcp_many = the references' xcw tables; one 16384-point slice per workgroup
        if (!p.xcp_many || !p.mv_many || !p.lag_many || !p.gscratch)
            return hipErrorInvalidValue;
        const long long mgrid = std::min<long long>(std::min<long long>(p.M, (long long)num_cus * 4), p.gscratch_slices * 2);
        if (mgrid < 1)
            return hipErrorInvalidValue;
        if (p.N < p.n)
            hipLaunchKernelGGL(xcorr_fused_real32k_multi<true>, dim3((unsigned)mgrid), dim3(1024), 0, stream, p);
        else
            hipLaunchKernelGGL(xcorr_fused_real32k_multi<false>, dim3((unsigned)mgrid), dim3(1024), 0, stream, p);
        return hipGetLastError();
    }
    if (!p.xcw || !p.mv || !p.lag)
        return hipErrorInvalidValue;
    const long long grid = std::min<long long>(p.M, (long long)num_cus * 8);
    if (p.N < p.n)
        hipLaunchKernelGGL(xcorr_fused_real32k_split<true>, dim3((unsigned)grid), dim3(1024), 0, stream, p);
    else
        hipLaunchKernelGGL(xcorr_fused_real32k_split<false>, dim3((unsigned)grid), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

} // namespace muse
