// xcorr_kernels.hip -- gfx950 (MI355X / CDNA4) device code for go-muse's
// z-normalized cross-correlation hot path.
//
// What one launch of the fused kernel computes, per comparison series y
// (reference: /root/reference/xcorr.go:160-197 xCorrWithX, called from
// muse_batch.go:73 and muse.go:71):
//     y  <- zNormalize(y)                      xcorr.go:164  (84-95)
//     s  <- [0 ... 0 | y]  (leading zeros)     xcorr.go:176-181
//     C  <- conj(FFT(s)) * X                   xcorr.go:183-185
//     cc <- IFFT(C) / n                        xcorr.go:186-187
//     mi <- first index of max |cc|; mv=cc[mi] xcorr.go:189-190 (39-50)
//     lag <- mi > n/2 ? mi - n : mi            xcorr.go:192-194
// and emits (lag, mv); sigma == 0 gives (0, 0.0) (xcorr.go:166-167).
//
// MI355X-first formulation (no rocFFT/hipFFT, no MFMA: add-heavy fp64
// butterflies on the VALU, data staged through LDS, one coalesced pass over
// the HBM-resident matrix):
//   * two real series are packed as one complex signal z = yA + i*yB.  With
//     Z = FFT(z):  FFT(Z[f] * conj(X[f]) / n)[k] = ccA[k] + i*ccB[k], so a
//     pair of series costs two FORWARD complex FFTs of length n and no
//     real-FFT untangling pass; the table xc[f] = conj(X_full[f])/n is
//     precomputed per batch (1/n is a power of two: exact).
//   * tuned n = 4096 kernel: 256 threads hold 16 complex points each; the FFT
//     is three radix-16 passes in registers with two LDS transposes (b128,
//     conflict-free by construction, see exchange comments); thread t starts
//     and ends with elements t + 256*a, so global loads are coalesced and the
//     second FFT consumes the first one's output layout directly.
//   * generic kernel: any power-of-two n <= 65536, in-place radix-2 DIF then
//     DIT in LDS (bit-reversed middle, no reorder pass).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_device.h"
#include "xcorr_kernels.h"

namespace muse {

__device__ __forceinline__ void lds_dif(double2 *z, int n, int logn, const double2 *__restrict__ twm)
{
    const int T = blockDim.x;
    for (int lh = logn - 1; lh >= 0; lh--) {
        const int half = 1 << lh;
        for (int j = threadIdx.x; j < (n >> 1); j += T) {
            const int pos = j & (half - 1);
            const int i0 = ((j >> lh) << (lh + 1)) + pos, i1 = i0 + half;
            const double2 a = z[i0], b = z[i1];
            z[i0] = cadd(a, b);
            // W_{2*half}^pos = W_65536^(pos * 65536/(2*half))
            z[i1] = cmul(csub(a, b), twm[pos << (15 - lh)]);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void lds_dit(double2 *z, int n, int logn, const double2 *__restrict__ twm)
{
    const int T = blockDim.x;
    for (int lh = 0; lh < logn; lh++) {
        const int half = 1 << lh;
        for (int j = threadIdx.x; j < (n >> 1); j += T) {
            const int pos = j & (half - 1);
            const int i0 = ((j >> lh) << (lh + 1)) + pos, i1 = i0 + half;
            const double2 a = z[i0], b = cmul(z[i1], twm[pos << (15 - lh)]);
            z[i0] = cadd(a, b);
            z[i1] = csub(a, b);
        }
        __syncthreads();
    }
}

// loads (and optionally z-normalises) up to two rows into z[] as re/im,
// right-aligned in n (leading zeros).  Returns flags through fa/fb.
__device__ __forceinline__ void lds_load_znorm(double2 *z, double *red, const double *__restrict__ ra,
                                               const double *__restrict__ rb, bool hasB, int N, int n,
                                               bool normalize, double post_scale_a, ZnFlags &fa, ZnFlags &fb)
{
    const int T = blockDim.x, t = threadIdx.x;
    const int pad = n - N;
    double s[2] = {0.0, 0.0};
    for (int i = t; i < n; i += T) {
        const int j = i - pad;
        double xa = 0.0, xb = 0.0;
        if (j >= 0) {
            xa = ra[j];
            if (hasB)
                xb = rb[j];
        }
        z[i] = make_double2(xa, xb);
        s[0] += xa;
        s[1] += xb;
    }
    fa.zero = fa.nan = fb.zero = fb.nan = false;
    if (normalize) {
        block_sum<2>(s, red);
        const double ca = -s[0] / (double)N, cb = -s[1] / (double)N;
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = t; i < n; i += T) {
            if (i - pad >= 0) {
                double2 e = z[i];
                e.x += ca;
                e.y += cb;
                z[i] = e;
                q[0] += e.x;
                q[1] = fma(e.x, e.x, q[1]);
                q[2] += e.y;
                q[3] = fma(e.y, e.y, q[3]);
            }
        }
        block_sum<4>(q, red + 8);
        const double ia = zn_scale(q[0], q[1], N, fa) * post_scale_a;
        const double ib = zn_scale(q[2], q[3], N, fb);
        const bool deadA = fa.zero || fa.nan, deadB = fb.zero || fb.nan;
        for (int i = t; i < n; i += T) {
            double2 e = z[i];
            e.x = deadA ? 0.0 : e.x * ia;
            e.y = deadB ? 0.0 : e.y * ib;
            z[i] = e;
        }
    }
    __syncthreads();
}

// block argmax over z[i].x (sel=0) or z[i].y (sel=1): first index of max |.|
__device__ __forceinline__ void lds_argmax(const double2 *z, double *red, int *redi, int n, int sel, int &idx,
                                           double &mv)
{
    const int T = blockDim.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double m = 0.0;
    int mi = 0x7fffffff;
    for (int i = t; i < n; i += T) {
        const double a = fabs(sel ? z[i].y : z[i].x);
        if (a > m) {
            m = a;
            mi = i;
        }
    }
    double wm = wave_max(m);
    if (lane == 0)
        red[32 + wave] = wm;
    __syncthreads();
    const double M = fmax(fmax(red[32], red[33]), fmax(red[34], red[35]));
    int c = (m == M && M > 0.0) ? mi : 0x7fffffff;
    c = wave_min_i(c);
    if (lane == 0)
        redi[wave] = c;
    __syncthreads();
    int I = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
    if (I == 0x7fffffff)
        I = 0;
    idx = I;
    mv = sel ? z[I].y : z[I].x;
    __syncthreads(); // red/redi free for the next call
}

__global__ __launch_bounds__(256) void xcorr_fused_generic(const FusedParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n = p.n, logn = p.logn, N = p.N, t = threadIdx.x;
    double2 *z = reinterpret_cast<double2 *>(smem_raw);
    double *red = reinterpret_cast<double *>(z + (p.gscratch ? 0 : n));
    if (p.gscratch) // n > 8192: work buffer in global memory, one slice per workgroup
        z = p.gscratch + (size_t)blockIdx.x * (size_t)n;
    int *redi = reinterpret_cast<int *>(red + 48);

    for (long long pair = blockIdx.x; pair < p.npairs; pair += gridDim.x) {
        const long long rA = 2 * pair, rB = rA + 1;
        const bool hasB = rB < p.M;
        ZnFlags fa, fb;
        lds_load_znorm(z, red, p.rows + rA * p.stride, p.rows + (hasB ? rB : rA) * p.stride, hasB, N, n,
                       p.normalize_y != 0, 1.0, fa, fb);
        lds_dif(z, n, logn, p.twm);
        for (int q = t; q < n; q += blockDim.x) {
            const int f = (int)(__brev((unsigned)q) >> (32 - logn));
            z[q] = cmul(z[q], p.xc[f]);
        }
        __syncthreads();
        lds_dit(z, n, logn, p.twm);
        if (p.cc_out) { // debug: full correlation of this pair
            for (int i = t; i < n; i += blockDim.x) {
                p.cc_out[rA * (long long)n + i] = z[i].x;
                if (hasB)
                    p.cc_out[rB * (long long)n + i] = z[i].y;
            }
        }
        int idx;
        double mv;
        lds_argmax(z, red, redi, n, 0, idx, mv);
        if (t == 0) {
            int lag = idx > n / 2 ? idx - n : idx;
            if (fa.zero) { mv = 0.0; lag = 0; }
            if (fa.nan) { mv = __builtin_nan(""); lag = 0; }
            p.mv[rA] = mv;
            p.lag[rA] = lag;
            if (p.nil_out)
                p.nil_out[rA] = fa.zero ? 1 : 0;
        }
        if (hasB) {
            lds_argmax(z, red, redi, n, 1, idx, mv);
            if (t == 0) {
                int lag = idx > n / 2 ? idx - n : idx;
                if (fb.zero) { mv = 0.0; lag = 0; }
                if (fb.nan) { mv = __builtin_nan(""); lag = 0; }
                p.mv[rB] = mv;
                p.lag[rB] = lag;
                if (p.nil_out)
                    p.nil_out[rB] = fb.zero ? 1 : 0;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------ reference spectrum kernel
// NewBatch precompute (muse_batch.go:35-47): x = zNormalize(ref)/(N-1),
// leading-zero pad to n, FFT.  One workgroup.  Writes
//   X[f], f <= n/2           (the batch's x, for muse_batch_spectrum)
//   xc[f] = conj(X_full[f]) * xc_scale, f < n   (table the fused kernels use)
// status[0] = 1 when sigma == 0 (or NaN) with normalize set.
__global__ __launch_bounds__(256) void ref_spectrum_kernel(const double *__restrict__ ref, int N, int n, int logn,
                                                           int normalize, double x_scale, double xc_scale,
                                                           const double2 *__restrict__ twm, double2 *X,
                                                           double2 *xc, float2 *xcf, double *xs, double2 *gscratch,
                                                           int *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2 *z = reinterpret_cast<double2 *>(smem_raw);
    double *red = reinterpret_cast<double *>(z + (gscratch ? 0 : n));
    if (gscratch)
        z = gscratch;
    ZnFlags fa, fb;
    // post scale 1/(N-1) on the real part (muse_batch.go:42)
    lds_load_znorm(z, red, ref, ref, false, N, n, normalize != 0, x_scale, fa, fb);
    if (!normalize && x_scale != 1.0) {
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            z[i].x *= x_scale;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        status[0] = (fa.zero || fa.nan) ? 1 : 0;
    if (xs) { // time-domain x (normalised, scaled, leading-zero padded): exact re-evaluation table
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            xs[i] = z[i].x;
    }
    lds_dif(z, n, logn, twm);
    for (int q = threadIdx.x; q < n; q += blockDim.x) {
        const int f = (int)(__brev((unsigned)q) >> (32 - logn));
        const double2 v = z[q];
        if (f <= n / 2)
            X[f] = v;
        xc[f] = make_double2(v.x * xc_scale, -v.y * xc_scale);
        if (xcf)
            xcf[f] = make_float2((float)(v.x * xc_scale), (float)(-v.y * xc_scale));
    }
}

// --------------------------------------------------- direct O(n^2) kernel
// Single pair, any n (used by the debug entry points when n is not a power of
// two, e.g. the n = 5 known-answer tables of xcorr_test.go).  One workgroup.
//   cc[k] = scale * sum_j y_pad[j] * x_pad[(j + k) mod n]
__global__ __launch_bounds__(256) void xcorr_direct_kernel(const double *__restrict__ x, int lenx,
                                                           const double *__restrict__ y, int leny, int n,
                                                           int normalize_x, int normalize_y, double x_scale,
                                                           double cc_scale, double *cc, int *lag_out, double *mv_out,
                                                           int *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *xs = reinterpret_cast<double *>(smem_raw);
    double *ys = xs + n;
    double *red = ys + n;
    const int T = blockDim.x, t = threadIdx.x;
    int nil = 0;
    for (int which = 0; which < 2; which++) {
        const double *src = which ? y : x;
        double *dst = which ? ys : xs;
        const int len = which ? leny : lenx;
        const int pad = n - len;
        const bool norm = which ? normalize_y : normalize_x;
        double s[2] = {0.0, 0.0};
        for (int i = t; i < n; i += T) {
            const int j = i - pad;
            const double e = j >= 0 ? src[j] : 0.0;
            dst[i] = e;
            s[0] += e;
        }
        double scale = which ? 1.0 : x_scale;
        if (norm) {
            block_sum<2>(s, red);
            const double c = -s[0] / (double)len;
            double q[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = t; i < n; i += T) {
                if (i - pad >= 0) {
                    const double e = dst[i] + c;
                    dst[i] = e;
                    q[0] += e;
                    q[1] = fma(e, e, q[1]);
                }
            }
            block_sum<4>(q, red + 8);
            ZnFlags f;
            scale *= zn_scale(q[0], q[1], len, f);
            if (f.zero || f.nan)
                nil = 1;
            __syncthreads();
        }
        for (int i = t; i < n; i += T)
            dst[i] *= scale;
        __syncthreads();
    }
    for (int k = t; k < n; k += T) {
        double acc = 0.0;
        for (int j = 0; j < n; j++) {
            int q = j + k;
            if (q >= n)
                q -= n;
            acc = fma(ys[j], xs[q], acc);
        }
        cc[k] = acc * cc_scale;
    }
    __syncthreads();
    if (t == 0) {
        int mi = 0;
        double mval = 0.0;
        for (int i = 0; i < n; i++) { // xcorr.go:39-50
            const double v = cc[i];
            if (fabs(v) > fabs(mval)) {
                mval = v;
                mi = i;
            }
        }
        double mv = cc[mi];
        if (mi > n / 2)
            mi -= n;
        if (nil) {
            mi = 0;
            mv = 0.0;
        }
        *lag_out = mi;
        *mv_out = mv;
        *status = nil;
    }
}

// -------------------------------------------------- synthetic rect + noise
// SURVEY 8d workload, after example_test.go:15-20: y[t] = A*1[|t-c| <= w/2]
// + 0.1*g(row,t).  Counter-based: every value is a pure function of
// (seed, global row, t).  Row kinds: 1/1024 constant rows (sigma == 0 path),
// 1/1024 exact copies of the reference (score 1, lag 0).
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ double gauss(uint64_t seed, uint64_t row, uint64_t t)
{
    const uint64_t h = mix64(mix64(seed ^ (row * 0xD1342543DE82EF95ull)) ^ t);
    const uint64_t h2 = mix64(h);
    const double u1 = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740993.0); // (0,1)
    const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);        // [0,1)
    return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}
__device__ __forceinline__ double synth_value(uint64_t seed, long long grow, int t, int N, unsigned flags = 0u)
{
    const uint64_t REF_ROW = 0xFFFFFFFFFFFFFFFFull;
    uint64_t row = (uint64_t)grow;
    if (grow >= 0) {
        const uint64_t kind = mix64(seed ^ (row * 0xA24BAED4963EE407ull)) & 1023ull;
        if (kind == 0 && !(flags & MUSE_SYNTH_NO_CONSTANTS)) // constant row, dyadic value: sum and mean are exact
            return 0.5 * (double)(1 + (row % 7));
        if (kind == 1 && !(flags & MUSE_SYNTH_NO_COPIES)) // exact copy of the reference
            row = REF_ROW;
    } else {
        row = REF_ROW;
    }
    double A, c, w;
    if (row == REF_ROW) {
        A = 1.5;
        c = (double)(N / 2);
        w = 10.0;
    } else {
        const uint64_t h = mix64(seed ^ (row * 0x9FB21C651E98DF25ull));
        const double ua = (double)(h & 0xFFFF) / 65536.0;
        const double uc = (double)((h >> 16) & 0xFFFF) / 65536.0;
        const int iw = 4 + (int)((h >> 32) % 61);
        A = 0.5 + ua * 42.5;
        if ((h >> 48) & 1)
            A = -A;
        c = (double)(N / 4) + floor(uc * (double)(N / 2));
        w = (double)iw;
    }
    const double rect = (fabs((double)t - c) <= 0.5 * w) ? A : 0.0;
    return rect + 0.1 * gauss(seed, row, (uint64_t)t);
}

template <typename T>
__global__ void synth_fill_kernel(T *rows, long long stride, long long first, long long count,
                                  long long global_first, int N, unsigned long long seed, unsigned flags)
{
    const long long total = count * (long long)N;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / N;
        const int t = (int)(e - r * N);
        rows[(first + r) * stride + t] = (T)synth_value(seed, global_first + r, t, N, flags);
    }
}
__global__ void synth_ref_kernel(double *ref, int N, unsigned long long seed)
{
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < N; t += gridDim.x * blockDim.x)
        ref[t] = synth_value(seed, -1, t, N);
}

// ----------------------------------------------------------- host launchers
__global__ __launch_bounds__(256) void lane_order_kernel(const double2 *__restrict__ in, double2 *__restrict__ out)
{
    const int t = threadIdx.x, k = blockIdx.x;
    out[256 * k + t] = in[256 * k + (t >> 4) + 16 * (t & 15)];
}

// one workgroup per 256 lags; a running window sum would do, but n^2 = 1.7e7 additions once per batch are free
__global__ __launch_bounds__(256) void indicator_corr_kernel(const double *__restrict__ xs, int n, int pad, double *__restrict__ c1)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n)
        return;
    double s = 0.0, c = 0.0; // compensated: the table corrects values of the same magnitude it is built from
    for (int j = pad; j < n; j++) {
        const double y = xs[(j + k) & (n - 1)] - c;
        const double tt = s + y;
        c = (tt - s) - y;
        s = tt;
    }
    c1[k] = s;
}

hipError_t launch_indicator_corr(const double *xs, int n, int pad, double *c1, hipStream_t stream)
{
    hipLaunchKernelGGL(indicator_corr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, xs, n, pad, c1);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void lane_order_rows_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, int R1)
{
    const int t = threadIdx.x, k = blockIdx.x, k1 = blockIdx.y;
    out[4096 * k1 + 256 * k + t] = in[k1 + R1 * (256 * k + (t >> 4) + 16 * (t & 15))];
}

hipError_t launch_lane_order_rows(const double2 *in, double2 *out, int R1, hipStream_t stream)
{
    hipLaunchKernelGGL(lane_order_rows_kernel, dim3(16, (unsigned)R1), dim3(256), 0, stream, in, out, R1);
    return hipGetLastError();
}

hipError_t launch_lane_order(const double2 *in, double2 *out, hipStream_t stream)
{
    hipLaunchKernelGGL(lane_order_kernel, dim3(16), dim3(256), 0, stream, in, out);
    return hipGetLastError();
}

hipError_t launch_fused(const FusedParams &p, int variant, int num_cus, hipStream_t stream)
{
    if (p.npairs <= 0)
        return hipSuccess;
    if (variant == KERNEL_R16_FOLD)
        return launch_fused_fold(p, num_cus, stream);
    if (variant == KERNEL_SMALL)
        return launch_fused_small(p, num_cus, stream);
    if (variant == KERNEL_LONG)
        return launch_fused_long(p, num_cus, stream);
    if (variant == KERNEL_REAL)
        return launch_fused_real(p, num_cus, stream);
    if (variant == KERNEL_REAL_SPLIT)
        return launch_fused_real_split(p, num_cus, stream);
    if (variant == KERNEL_STOCKHAM)
        return launch_fused_stockham(p, num_cus, stream);
    if (variant == KERNEL_R16_OCC3)
        return launch_fused_occ4(p, num_cus, stream);
    const bool global_mode = p.n > GENERIC_LDS_MAX_N;
    if (global_mode && !p.gscratch)
        return hipErrorInvalidValue;
    const size_t lds = (global_mode ? 0 : (size_t)p.n * sizeof(double2)) + 64 * sizeof(double);
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(xcorr_fused_generic),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        configured = lds;
    }
    long long grid = p.npairs;
    const long long cap = (long long)num_cus * (global_mode ? GENERIC_GLOBAL_WGS_PER_CU : 16); // global mode: one scratch slice each
    if (grid > cap)
        grid = cap;
    if (global_mode && grid > p.gscratch_slices)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(xcorr_fused_generic, dim3((unsigned)grid), dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_ref_spectrum(const double *ref_dev, int N, int n, int logn, int normalize, double x_scale,
                               double xc_scale, const double2 *twm, double2 *X, double2 *xc, float2 *xcf, double *xs,
                               double2 *gscratch, int *status, hipStream_t stream)
{
    if (n > GENERIC_LDS_MAX_N && !gscratch)
        return hipErrorInvalidValue;
    const size_t lds = (n > GENERIC_LDS_MAX_N ? 0 : (size_t)n * sizeof(double2)) + 64 * sizeof(double);
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ref_spectrum_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        configured = lds;
    }
    hipLaunchKernelGGL(ref_spectrum_kernel, dim3(1), dim3(256), lds, stream, ref_dev, N, n, logn, normalize, x_scale,
                       xc_scale, twm, X, xc, xcf, xs, n > GENERIC_LDS_MAX_N ? gscratch : nullptr, status);
    return hipGetLastError();
}

hipError_t launch_direct(const double *x, int lenx, const double *y, int leny, int n, int normalize_x,
                         int normalize_y, double x_scale, double cc_scale, double *cc, int *lag, double *mv,
                         int *status, hipStream_t stream)
{
    const size_t lds = (size_t)n * 2 * sizeof(double) + 64 * sizeof(double);
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(xcorr_direct_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        configured = lds;
    }
    hipLaunchKernelGGL(xcorr_direct_kernel, dim3(1), dim3(256), lds, stream, x, lenx, y, leny, n, normalize_x,
                       normalize_y, x_scale, cc_scale, cc, lag, mv, status);
    return hipGetLastError();
}

// ---------------------------------------------------------------- two-sided xCorr: pairs whose statistics left the float64 range
// The batched xCorr kernels take sum d^2 for their exact power-of-two scales; for FINITE samples with |x| >~ 1e154 that sum is
// Inf and the pair comes back like a NaN pair, where the reference (xcorr.go:108-143) still returns numbers: raw, cc is
// finite whenever the products X conj(Y) are; normalized, gonum's StdDev overflows to +Inf and the series becomes all zeros
// (x / Inf), or -- when even (sum d)^2 overflows -- NaN.  This kernel looks at such a pair again, one workgroup per listed
// pair, and decides (code) what the reference's arithmetic gives:
//   0  NaN stands: a NaN / Inf sample, a NaN variance, or products that overflow float64 in the reference too;
//   1  every cc is zero (lag 0, value 0): a series scaled by 1 / Inf, or a raw pair with an all-zero series;
//   2  recompute on copies scaled by exact powers of two, (gx, gy): normalized -- each series to magnitude 1 (z-normalisation
//      cancels any scale); raw -- x 2^-k and y 2^+k, which leaves x[i] y[j] and with it every cc unchanged (bilinear) while
//      both series meet at the geometric mean of their magnitudes.
// The variance is the corrected two-pass of gonum stat.StdDev (xcorr.go:88): (sum d^2 - (sum d)^2 / N) / (N - 1) around the mean.
struct SeriesLook {
    bool finite;   // every sample is a number
    double maxabs; // max |x|
    double var;    // normalized: the reference's variance (may be +Inf or NaN)
};
__device__ SeriesLook look_at_series(const double *__restrict__ x, const int N, const bool normalize, double *red)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const auto block_sum = [&](double v) {
        v = wave_sum_dpp(v);
        __syncthreads();
        if (lane == 0)
            red[wave] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const auto block_max = [&](double v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0)
            red[wave] = v;
        __syncthreads();
        return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    };
    double bad = 0.0, mx = 0.0, sum = 0.0;
    for (int i = t; i < N; i += 256) {
        const double v = x[i];
        bad = __builtin_isfinite(v) ? bad : 1.0;
        mx = fmax(mx, fabs(v)); // (fmax drops a NaN operand: `bad` carries it)
        sum += v;
    }
    SeriesLook s;
    s.finite = block_max(bad) == 0.0;
    s.maxabs = block_max(mx);
    s.var = 0.0;
    if (normalize) {
        const double mean = block_sum(sum) / (double)N;
        double ss = 0.0, comp = 0.0;
        for (int i = t; i < N; i += 256) {
            const double d = x[i] - mean;
            ss = fma(d, d, ss);
            comp += d;
        }
        ss = block_sum(ss);
        comp = block_sum(comp);
        s.var = (ss - comp * comp / (double)N) / (double)(N - 1);
    }
    return s;
}
__device__ __forceinline__ int exp_of(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7ff) - 1023; }
__device__ __forceinline__ double pow2(int e) { return __longlong_as_double((long long)(e + 1023) << 52); } // -1022 <= e <= 1023

__global__ __launch_bounds__(256) void two_sided_rescue_kernel(const double *__restrict__ xrows, long long xstride, int Nx,
                                                               const double *__restrict__ yrows, long long ystride, int Ny,
                                                               int n, int normalize, const long long *__restrict__ list,
                                                               int *__restrict__ code, double2 *__restrict__ scale)
{
    __shared__ double red[4];
    const long long pair = list[blockIdx.x];
    const SeriesLook a = look_at_series(xrows + pair * xstride, Nx, normalize != 0, red);
    const SeriesLook b = look_at_series(yrows + pair * ystride, Ny, normalize != 0, red);
    if (threadIdx.x != 0)
        return;
    int c = 0;
    double gx = 1.0, gy = 1.0;
    if (a.finite && b.finite) {
        if (normalize) {
            const bool nanv = a.var != a.var || b.var != b.var;
            if (!nanv && (__builtin_isinf(a.var) || __builtin_isinf(b.var)))
                c = 1; // x / sigma with sigma = +Inf: the series is all zeros, and so is every cc
            else if (!nanv && a.maxabs > 0.0 && b.maxabs > 0.0) {
                c = 2;
                gx = pow2(-max(-1022, min(1022, exp_of(a.maxabs))));
                gy = pow2(-max(-1022, min(1022, exp_of(b.maxabs))));
            }
        } else if (a.maxabs == 0.0 || b.maxabs == 0.0) {
            c = 1;
        } else {
            const int ex = exp_of(a.maxabs), ey = exp_of(b.maxabs);
            // both series are brought to magnitude ~1 (exact powers of two), the pair is recomputed there, and the HOST scales lag
            // value and cc back by 2^(ex + ey) (capi_xcorr.hip): the result overflows exactly where the reference's own products
            // X conj(Y) (up to n max|x| max|y|) leave the float64 range -- NaN stands there, as in the reference -- and not 24 ... 50
            // binades earlier, as the round-5 bound on the recomputation's intermediates had it.  Below 2^-1000 every product
            // underflows in the reference as well: every cc is zero.
            if (ex + ey < -1000) {
                c = 1;
            } else {
                c = 2;
                gx = pow2(-max(-1022, min(1022, ex)));
                gy = pow2(-max(-1022, min(1022, ey)));
            }
        }
    }
    code[blockIdx.x] = c;
    scale[blockIdx.x] = make_double2(gx, gy);
}
// dst row k = src row list[k] * g[k].x (which = 0) or * g[k].y (which = 1): exact (powers of two)
__global__ __launch_bounds__(256) void scale_listed_rows_kernel(const double *__restrict__ src, long long stride, int N,
                                                                const long long *__restrict__ list, const double2 *__restrict__ g,
                                                                int which, double *__restrict__ dst)
{
    const long long pair = list[blockIdx.x];
    const double f = which ? g[blockIdx.x].y : g[blockIdx.x].x;
    const double *s = src + pair * stride;
    double *d = dst + (long long)blockIdx.x * N;
    for (int i = threadIdx.x; i < N; i += 256)
        d[i] = s[i] * f;
}
hipError_t launch_two_sided_rescue(const double *xrows, long long xstride, int Nx, const double *yrows, long long ystride, int Ny,
                                   int n, int normalize, const long long *list, int count, int *code, double2 *scale,
                                   hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(two_sided_rescue_kernel, dim3((unsigned)count), dim3(256), 0, stream, xrows, xstride, Nx, yrows, ystride, Ny,
                       n, normalize, list, code, scale);
    return hipGetLastError();
}
hipError_t launch_scale_listed_rows(const double *src, long long stride, int N, const long long *list, const double2 *g, int which,
                                    int count, double *dst, hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(scale_listed_rows_kernel, dim3((unsigned)count), dim3(256), 0, stream, src, stride, N, list, g, which, dst);
    return hipGetLastError();
}

hipError_t launch_synth(double *rows, long long stride, long long first, long long count, long long global_first,
                        int N, unsigned long long seed, unsigned flags, hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(synth_fill_kernel<double>, dim3(4096), dim3(256), 0, stream, rows, stride, first, count, global_first,
                       N, seed, flags);
    return hipGetLastError();
}
// the same workload rounded to float32 (float32-storage groups)
hipError_t launch_synth_f32(float *rows, long long stride, long long first, long long count, long long global_first,
                            int N, unsigned long long seed, unsigned flags, hipStream_t stream)
{
    if (count <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(synth_fill_kernel<float>, dim3(4096), dim3(256), 0, stream, rows, stride, first, count, global_first,
                       N, seed, flags);
    return hipGetLastError();
}
hipError_t launch_synth_ref(double *ref, int N, unsigned long long seed, hipStream_t stream)
{
    hipLaunchKernelGGL(synth_ref_kernel, dim3(16), dim3(256), 0, stream, ref, N, seed);
    return hipGetLastError();
}

} // namespace muse
