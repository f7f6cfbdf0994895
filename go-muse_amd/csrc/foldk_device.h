// foldk_device.h -- the building blocks of the n = 4096 folded-arithmetic transforms shared by xcorr_r16_fold.hip (one
// pair per workgroup pass) and xcorr_long.hip (the 4096-point rows of the four-step transform): the two LDS
// transposes in half rounds, factor fetchers, the generalised pass with early factors, the spectrum multiply folded
// into the second transform's first stage, and the record / argmax helpers of the n = 4096 kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fold_device.h"
#include "r16_device.h"

namespace muse {

namespace foldk {

using namespace occ4;
using namespace fold;

constexpr int XW = 544; // double2 per wave-private quarter of the 8 x 272 buffer (8 rows x 68)

// output k of the preceding pass sits in register PERM(k): 0 = natural, 1 = bit-reversed (NR passes)
template <int PERM>
__device__ __forceinline__ constexpr int pr(int k)
{
    return PERM ? BR16(k) : k;
}

// Workgroup-wide transpose in two half rounds (layouts and bank analysis: xcorr_r16_fast.hip exchange_cross)
// TAILBAR: one more barrier right behind the late waves' reads, so that the NEXT use of the buffer (a wave-local
// transpose into the wave's private quarter) needs none: the waves are still in step here, whereas a barrier in
// front of that next use also waits out everything the waves drifted apart in between (2.3 k cycles per pair,
// profiles/r02_fold_phase_stamps.txt).
// wcol (MODE 0): the column the writer holds (t unless pass 1 ran on relabelled columns: r16_device.h, wide_column)
template <int MODE, int PERM, bool TAILBAR = false>
__device__ __forceinline__ void exchange_cross(double2 (&v)[16], double2 *xbuf, const int wave, const int t_, const int wcol = -1)
{
    const int t = t_;
    const int hi = t >> 4, lo = t & 15;
    // relabelled writers (wcol >= 0: lanes 0-31 of a wave hold the even columns, lanes 32-63 the odd ones) would put the eight
    // lanes of a ds_write_b128 group on every second 16-byte slot (2-way conflicts: SQ_LDS_BANK_CONFLICT 11 % of the LDS
    // cycles); the image then keeps column c of a row at slot (c & ~15) | rot4(c & 15), rot4 = the low four bits rotated right
    // by one: eight consecutive even (or odd) columns land on eight consecutive slots, and a reader's sixteen lanes (distinct
    // lo) still hit sixteen different slots modulo 16
    const auto rot4 = [](int c) { return (c & ~15) | ((c & 15) >> 1) | ((c & 1) << 3); };
    const int wbase = MODE ? 17 * lo + hi : (wcol >= 0 ? rot4(wcol) : t);
    const int rbase = 272 * (hi & 7) + (MODE ? 17 * lo : (wcol >= 0 ? rot4(lo) : lo));
    const bool early = wave < 2;
    lds_barrier(); // buffer free: every wave is done with its previous (wave-local or shared) use
#pragma unroll
    for (int k = 0; k < 8; k++)
        lds_st2(xbuf + 272 * k + wbase, v[pr<PERM>(k)]);
    lds_barrier();
    if (early) {
        double2 w[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            w[e] = lds_ld2(xbuf + rbase + (MODE ? e : 16 * e));
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++)
            lds_st2(xbuf + 272 * k + wbase, v[pr<PERM>(8 + k)]);
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = w[e];
    } else {
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++)
            lds_st2(xbuf + 272 * k + wbase, v[pr<PERM>(8 + k)]);
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = lds_ld2(xbuf + rbase + (MODE ? e : 16 * e));
    }
    if (TAILBAR)
        lds_barrier();
}

// The same through a FULL 16 x 272 buffer (69,632 B: kernels that run two workgroups per CU): one round, two barriers.
// Row k takes output k, the reader with hi reads row hi; the in-row patterns are the half-round ones (conflict-free).
constexpr int OCC_XBUF_FULL = 16 * 272;
template <int MODE, int PERM>
__device__ __forceinline__ void exchange_cross_full(double2 (&v)[16], double2 *xbuf, const int t, const int wcol = -1)
{
    const int hi = t >> 4, lo = t & 15;
    const auto rot4 = [](int c) { return (c & ~15) | ((c & 15) >> 1) | ((c & 1) << 3); }; // (see exchange_cross)
    const int wbase = MODE ? 17 * lo + hi : (wcol >= 0 ? rot4(wcol) : t);
    const int rbase = 272 * hi + (MODE ? 17 * lo : (wcol >= 0 ? rot4(lo) : lo));
    lds_barrier(); // buffer free: every wave is done with its previous (wave-local or shared) use
#pragma unroll
    for (int k = 0; k < 16; k++)
        lds_st2(xbuf + 272 * k + wbase, v[pr<PERM>(k)]);
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 16; e++)
        v[e] = lds_ld2(xbuf + rbase + (MODE ? e : 16 * e));
}
// Wave-local transpose among the sixteen lanes that share hi, through the wave's quarter of the full buffer
// (xw = xbuf + 1088 wave): lane (hl, lo) writes output k to 272 hl + 17 k + lo and reads input e from 272 hl + 17 lo + e.
// Bank check (MI355X_MICROARCH.md, LDS): a ds_read_b128 group {0-3, 12-15, 20-27} holds (hl, lo) = (0, 0..3), (0, 12..15),
// (1, 4..11): slots 17 lo + 272 hl = lo + 16 (lo + 17 hl), i.e. sixteen different residues mod 16; a ds_write_b128 group is
// eight consecutive lanes on eight consecutive slots.  No barrier: a wave's LDS operations execute in order.
template <int PERM>
__device__ __forceinline__ void exchange_local_full(double2 (&v)[16], double2 *xw, const int t)
{
    const int hl = (t >> 4) & 3, lo = t & 15;
    const int wbase = 272 * hl + lo;
    const int rbase = 272 * hl + 17 * lo;
#pragma unroll
    for (int k = 0; k < 16; k++)
        lds_st2(xw + 17 * k + wbase, v[pr<PERM>(k)]);
    fence();
    asm volatile("" ::: "memory"); // (the compiler must not move a lane's reads above its writes: other lanes' data arrives through them)
    fence();
#pragma unroll
    for (int e = 0; e < 16; e++)
        v[e] = lds_ld2(xw + rbase + e);
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) char *lds_ptr;
#define MUSE_LDS_ADDR(p) ((unsigned)(unsigned long long)(lds_ptr)(p))
#else
#define MUSE_LDS_ADDR(p) 0u
#endif
// Wave-local transpose among the sixteen lanes that share hi (layout: xcorr_r16_fast.hip exchange_local)
template <int PERM>
__device__ __forceinline__ void exchange_local(double2 (&v)[16], double2 *xw, const int t_)
{
    const int t = t_;
    const int hl = (t >> 4) & 3, lo = t & 15;
    const int wbase = 17 * hl + lo;
    const int rbase = 68 * (lo & 7) + 17 * hl;
#pragma unroll
    for (int k = 0; k < 8; k++)
        lds_st2(xw + 68 * k + wbase, v[pr<PERM>(k)]);
    const unsigned waddr = MUSE_LDS_ADDR(xw + wbase), raddr = MUSE_LDS_ADDR(xw + rbase);
    const unsigned long long first = __ballot(lo < 8); // lanes that read in round 0
    d2v w[16], d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        d[k].x = v[pr<PERM>(8 + k)].x;
        d[k].y = v[pr<PERM>(8 + k)].y;
    }
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 exec, %[sv], %[m]\n\t"
                 "ds_read_b128 %[w0], %[ra]\n\t"
                 "ds_read_b128 %[w1], %[ra] offset:16\n\t"
                 "ds_read_b128 %[w2], %[ra] offset:32\n\t"
                 "ds_read_b128 %[w3], %[ra] offset:48\n\t"
                 "ds_read_b128 %[w4], %[ra] offset:64\n\t"
                 "ds_read_b128 %[w5], %[ra] offset:80\n\t"
                 "ds_read_b128 %[w6], %[ra] offset:96\n\t"
                 "ds_read_b128 %[w7], %[ra] offset:112\n\t"
                 "ds_read_b128 %[w8], %[ra] offset:128\n\t"
                 "ds_read_b128 %[w9], %[ra] offset:144\n\t"
                 "ds_read_b128 %[w10], %[ra] offset:160\n\t"
                 "ds_read_b128 %[w11], %[ra] offset:176\n\t"
                 "ds_read_b128 %[w12], %[ra] offset:192\n\t"
                 "ds_read_b128 %[w13], %[ra] offset:208\n\t"
                 "ds_read_b128 %[w14], %[ra] offset:224\n\t"
                 "ds_read_b128 %[w15], %[ra] offset:240\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "ds_write_b128 %[wa], %[d0]\n\t"
                 "ds_write_b128 %[wa], %[d1] offset:1088\n\t"
                 "ds_write_b128 %[wa], %[d2] offset:2176\n\t"
                 "ds_write_b128 %[wa], %[d3] offset:3264\n\t"
                 "ds_write_b128 %[wa], %[d4] offset:4352\n\t"
                 "ds_write_b128 %[wa], %[d5] offset:5440\n\t"
                 "ds_write_b128 %[wa], %[d6] offset:6528\n\t"
                 "ds_write_b128 %[wa], %[d7] offset:7616\n\t"
                 "s_andn2_b64 exec, %[sv], %[m]\n\t"
                 "ds_read_b128 %[w0], %[ra]\n\t"
                 "ds_read_b128 %[w1], %[ra] offset:16\n\t"
                 "ds_read_b128 %[w2], %[ra] offset:32\n\t"
                 "ds_read_b128 %[w3], %[ra] offset:48\n\t"
                 "ds_read_b128 %[w4], %[ra] offset:64\n\t"
                 "ds_read_b128 %[w5], %[ra] offset:80\n\t"
                 "ds_read_b128 %[w6], %[ra] offset:96\n\t"
                 "ds_read_b128 %[w7], %[ra] offset:112\n\t"
                 "ds_read_b128 %[w8], %[ra] offset:128\n\t"
                 "ds_read_b128 %[w9], %[ra] offset:144\n\t"
                 "ds_read_b128 %[w10], %[ra] offset:160\n\t"
                 "ds_read_b128 %[w11], %[ra] offset:176\n\t"
                 "ds_read_b128 %[w12], %[ra] offset:192\n\t"
                 "ds_read_b128 %[w13], %[ra] offset:208\n\t"
                 "ds_read_b128 %[w14], %[ra] offset:224\n\t"
                 "ds_read_b128 %[w15], %[ra] offset:240\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [w0] "=&v"(w[0]), [w1] "=&v"(w[1]), [w2] "=&v"(w[2]), [w3] "=&v"(w[3]), [w4] "=&v"(w[4]),
                   [w5] "=&v"(w[5]), [w6] "=&v"(w[6]), [w7] "=&v"(w[7]), [w8] "=&v"(w[8]), [w9] "=&v"(w[9]),
                   [w10] "=&v"(w[10]), [w11] "=&v"(w[11]), [w12] "=&v"(w[12]), [w13] "=&v"(w[13]),
                   [w14] "=&v"(w[14]), [w15] "=&v"(w[15]), [sv] "=&s"(sv)
                 : [ra] "v"(raddr), [wa] "v"(waddr), [m] "s"(first), [d0] "v"(d[0]), [d1] "v"(d[1]), [d2] "v"(d[2]),
                   [d3] "v"(d[3]), [d4] "v"(d[4]), [d5] "v"(d[5]), [d6] "v"(d[6]), [d7] "v"(d[7])
                 : "memory", "scc");
#pragma unroll
    for (int e = 0; e < 16; e++)
        v[e] = make_double2(w[e].x, w[e].y);
}

// the eight per-thread factors of a generalised pass (fold_device.h):
//   pass 2 (delta = j / 16): LDS table g2s[16 s + j];  pass 3 (delta = u / 256): lane-ordered global table [8][256]
struct G2Fetch {
    const double2 *p;
    int j;
    __device__ __forceinline__ double2 operator()(int s) const { return p[16 * s + j]; }
};
struct G3Fetch {
    const double2 *p;
    int t;
    __device__ __forceinline__ double2 operator()(int s) const
    {   // one scalar base per two planes: the odd plane sits at immediate offset -4096 B
        return ldg2(scalar_ptr_at(p, ((s + 1) & ~1) * 256), t - 256 * (s & 1));
    }
};

// the eight factors of a pass from TWO table entries: w8 (plane 2) and w16 (plane 4); w4 = w8^2, w2 = w4^2, w8 W_8 and
// w16 W_16^q by constant multiplications (24 fp64 instructions instead of six 16-byte L2 loads per thread)
struct G3Derived {
    double2 g[8];
    __device__ __forceinline__ static double2 sq(const double2 w) { return make_double2(fma_(w.x, w.x, -(w.y * w.y)), (w.x + w.x) * w.y); }
    __device__ __forceinline__ static double2 mulc(const double2 w, const double c, const double s) // w (c - i s)
    {
        return make_double2(fma_(w.y, s, w.x * c), fma_(-w.x, s, w.y * c));
    }
    __device__ __forceinline__ G3Derived(const double2 *tab, int t)
    {
        constexpr double H = 0.70710678118654752440;
        const double2 w8 = G3Fetch{tab, t}(2), w16 = G3Fetch{tab, t}(4);
        g[2] = w8;
        g[4] = w16;
        g[1] = sq(w8);
        g[0] = sq(g[1]);
        g[3] = make_double2(H * (w8.x + w8.y), H * (w8.y - w8.x)); // w8 W_8
        g[5] = mulc(w16, C16_1, S16_1);
        g[6] = make_double2(H * (w16.x + w16.y), H * (w16.y - w16.x));
        g[7] = mulc(w16, S16_1, C16_1);
    }
    __device__ __forceinline__ double2 operator()(int s) const { return g[s]; }
};

// generalised pass with its factors in L2 (pass 3): the second batch of four is requested behind the second stage
// (two batches in flight would not fit 128 registers), stages separated by scheduling fences
template <typename F>
__device__ __forceinline__ void gdft16_nr_l2(double2 (&v)[16], F fetch)
{
    double2 ga[4], gb[4];
#pragma unroll
    for (int s = 0; s < 4; s++)
        ga[s] = fetch(s);
    gdft16_nr_s12(v, ga[0], ga[1]);
    fence();
#pragma unroll
    for (int s = 0; s < 4; s++)
        gb[s] = fetch(4 + s);
    fence();
    gdft16_nr_s3(v, ga[2], ga[3]);
    fence();
    fence();
    gdft16_nr_s4(v, gb[0], gb[1], gb[2], gb[3]);
}

// LDS record of one pair (one per parity): as in xcorr_r16_fast.hip
constexpr int REC = 36;

// cross-wave argmax combine + variance + store (lanes 0 / 1, one series each);
// returns true when the series' statistics are NaN/Inf or the pair's sigmas are too far apart (the pair is redone)
__device__ __forceinline__ bool finalize(const double *r, const int series, const double invN, const double invNm1,
                                         double *mv_out, int *lag_out)
{
    double m[4], s[4], ix[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        m[w] = r[6 * w + 3 * series];
        s[w] = r[6 * w + 3 * series + 1];
        ix[w] = r[6 * w + 3 * series + 2];
    }
    const double s2 = (r[24 + series] + r[26 + series]) + (r[28 + series] + r[30 + series]);
    const Stat st{r[32 + series], s2};
    double best = m[0], bsv = s[0], bidx = ix[0];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        if (m[w] > best || (m[w] == best && ix[w] < bidx)) {
            best = m[w];
            bsv = s[w];
            bidx = ix[w];
        }
    }
    bool zero, nan;
    const double var = variance(st, invN, invNm1, zero, nan);
    const int idx = (best > 0.0) ? (int)bidx : 0; // nothing above 0: index 0, mv = cc[0]
    double y = __builtin_amdgcn_rsq(var);
    y = y * fma(-0.5 * var * y, y, 1.5);
    y = y * fma(-0.5 * var * y, y, 1.5);
    double mv = ((best > 0.0) ? bsv : s[0]) * y;
    int lag = idx > 2048 ? idx - 4096 : idx;
    if (zero) { mv = 0.0; lag = 0; }              // xcorr.go:166-167
    if (nan) { mv = __builtin_nan(""); lag = 0; } // placeholder: the pair is redone
    *mv_out = mv;
    *lag_out = lag;
    bool redo = nan;
    if (series == 0 && r[35] != 0.0) { // the pair's other series: sigmas too far apart for one shared transform?
        const double s2b = (r[25] + r[27]) + (r[29] + r[31]);
        const Stat sb{r[33], s2b};
        bool zb, nb;
        const double varb = variance(sb, invN, invNm1, zb, nb);
        redo = redo || (!nb && sigma_spread_too_wide(var, varb));
    }
    return redo;
}

// maxAbsIndex (xcorr.go:39-50) over the wave's 16 x 64 values of both series; value of lag index t + 256 m sits in
// register BR16(m).  Writes the wave's {max |cc|, signed value (cc[0] when nothing is above 0), index} per series.
// Few instructions of ANY kind: a wave issues at most one instruction every four cycles, scalar ones included, so the
// locate step keeps its bookkeeping in the vector carry (v_cmp_eq + v_addc shift "|v| == max" into a per-lane 16-bit mask)
// instead of a scalar select chain per register.
#define MUSE_HIWORD_CASE(m) case m: h = __builtin_amdgcn_readlane(__double2hiint(S == 0 ? v[BR16(m)].x : v[BR16(m)].y), l); break;
template <int S>
__device__ __forceinline__ int hiword_at(const double2 (&v)[16], const int m, const int l) // m, l wave-uniform
{
    int h = 0;
    switch (m) {
        MUSE_HIWORD_CASE(0) MUSE_HIWORD_CASE(1) MUSE_HIWORD_CASE(2) MUSE_HIWORD_CASE(3)
        MUSE_HIWORD_CASE(4) MUSE_HIWORD_CASE(5) MUSE_HIWORD_CASE(6) MUSE_HIWORD_CASE(7)
        MUSE_HIWORD_CASE(8) MUSE_HIWORD_CASE(9) MUSE_HIWORD_CASE(10) MUSE_HIWORD_CASE(11)
        MUSE_HIWORD_CASE(12) MUSE_HIWORD_CASE(13) MUSE_HIWORD_CASE(14) MUSE_HIWORD_CASE(15)
    }
    return h;
}
#undef MUSE_HIWORD_CASE
// lowest index among the lanes whose mask is non-zero (bit m of a lane's mask: its value of register BR16(m) is the
// maximum): lowest m first, then the lowest lane.  One trip unless several lanes hold exactly the maximum.
__device__ __forceinline__ void lowest_hit(const unsigned acc, int &m_out, int &l_out)
{
    unsigned long long c = __ballot(acc != 0u);
    int bm = 16, bl = 0;
    while (c != 0ull) {
        const int l = __ffsll((long long)c) - 1;
        const int m = __ffs((int)__builtin_amdgcn_readlane((int)acc, l)) - 1;
        if (m < bm) {
            bm = m;
            bl = l;
        }
        c &= c - 1ull;
    }
    m_out = bm;
    l_out = bl;
}
__device__ __forceinline__ void wave_argmax_store(const double2 (&v)[16], const int wave, const int lane, double *ra_)
{
    double ma = 0.0, mb = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        ma = fmax(ma, fabs(v[k].x));
        mb = fmax(mb, fabs(v[k].y));
    }
    const double wa = wave_max_nonneg(ma), wb = wave_max_nonneg(mb);
    unsigned accA = 0u, accB = 0u;
#pragma unroll
    for (int m = 15; m >= 0; m--) { // acc = 2 acc + (|v| == w): register BR16(m) ends up at bit m
        const int k = BR16(m);
        asm("v_cmp_eq_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(accA) : "v"(v[k].x), "s"(wa) : "vcc");
        asm("v_cmp_eq_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(accB) : "v"(v[k].y), "s"(wb) : "vcc");
    }
    int widxA = 0x7fffffff, widxB = 0x7fffffff;
    double svA = 0.0, svB = 0.0;
    if (__double_as_longlong(wa) != 0ll) { // (wa >= 0 or NaN: nothing equals a NaN, the masks are empty then)
        int m, l;
        lowest_hit(accA, m, l);
        if (m < 16) {
            widxA = wave * 64 + l + 256 * m;
            svA = (hiword_at<0>(v, m, l) < 0) ? -wa : wa;
        }
    }
    if (__double_as_longlong(wb) != 0ll) {
        int m, l;
        lowest_hit(accB, m, l);
        if (m < 16) {
            widxB = wave * 64 + l + 256 * m;
            svB = (hiword_at<1>(v, m, l) < 0) ? -wb : wb;
        }
    }
    const double cc0a = v[0].x, cc0b = v[0].y; // index t + 256 * 0 (BR16(0) = 0): cc[0] in wave 0 lane 0
    if (lane == 0) {
        ra_[0] = widxA == 0x7fffffff ? 0.0 : wa;
        ra_[1] = widxA == 0x7fffffff ? cc0a : svA;
        ra_[2] = (double)widxA;
        ra_[3] = widxB == 0x7fffffff ? 0.0 : wb;
        ra_[4] = widxB == 0x7fffffff ? cc0b : svB;
        ra_[5] = (double)widxB;
    }
}

// One series only (S = 0: the .x components, S = 1: the .y components); out3 = {max |cc|, signed value, index}.
template <int S>
__device__ __forceinline__ void wave_argmax_store_one(const double2 (&v)[16], const int wave, const int lane, double *out3)
{
    double ma = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        ma = fmax(ma, fabs(S == 0 ? v[k].x : v[k].y));
    const double wa = wave_max_nonneg(ma);
    unsigned acc = 0u;
#pragma unroll
    for (int m = 15; m >= 0; m--) {
        const int k = BR16(m);
        asm("v_cmp_eq_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(S == 0 ? v[k].x : v[k].y), "s"(wa) : "vcc");
    }
    int widx = 0x7fffffff;
    double sv = 0.0;
    if (__double_as_longlong(wa) != 0ll) {
        int m, l;
        lowest_hit(acc, m, l);
        if (m < 16) {
            widx = wave * 64 + l + 256 * m;
            sv = (hiword_at<S>(v, m, l) < 0) ? -wa : wa;
        }
    }
    const double cc0 = S == 0 ? v[0].x : v[0].y;
    if (lane == 0) {
        out3[0] = widx == 0x7fffffff ? 0.0 : wa;
        out3[1] = widx == 0x7fffffff ? cc0 : sv;
        out3[2] = (double)widx;
    }
}

// the same with all sixteen factors already in registers (xf[j] = xc for k3 = j)
__device__ __forceinline__ void xc_stage1_pre(double2 (&v)[16], const double2 (&xf)[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bf_xc(v[BR16(2 * i)], v[BR16(2 * i) + 1], xf[2 * i], xf[2 * i + 8]);
        bf_xc(v[BR16(2 * i + 1)], v[BR16(2 * i + 1) + 1], xf[2 * i + 1], xf[2 * i + 9]);
    }
}

// spectrum multiply folded into the first stage of the second transform's plain pass:
// z[b] at v[BR16(b)] (b = k3), xc for k3 = b from the lane-ordered table; four batches of four factors, two in flight
template <typename F>
__device__ __forceinline__ void xc_stage1(double2 (&v)[16], F xcl)
{
    double2 xa[4], xb[4];
    // batch i covers butterflies b = 2 i, 2 i + 1: factors xc[2i], xc[2i + 8], xc[2i + 1], xc[2i + 9]
#define MUSE_XC_LOAD(dst, i)          \
    dst[0] = xcl(2 * (i));            \
    dst[1] = xcl(2 * (i) + 8);        \
    dst[2] = xcl(2 * (i) + 1);        \
    dst[3] = xcl(2 * (i) + 9);
#define MUSE_XC_USE(src, i)                                                      \
    bf_xc(v[BR16(2 * (i))], v[BR16(2 * (i)) + 1], src[0], src[1]);               \
    bf_xc(v[BR16(2 * (i) + 1)], v[BR16(2 * (i) + 1) + 1], src[2], src[3]);
    MUSE_XC_LOAD(xa, 0)
    MUSE_XC_LOAD(xb, 1)
    fence();
    MUSE_XC_USE(xa, 0)
    fence();
    MUSE_XC_LOAD(xa, 2)
    MUSE_XC_USE(xb, 1)
    fence();
    MUSE_XC_LOAD(xb, 3)
    MUSE_XC_USE(xa, 2)
    fence();
    MUSE_XC_USE(xb, 3)
#undef MUSE_XC_LOAD
#undef MUSE_XC_USE
}

} // namespace foldk

} // namespace muse
