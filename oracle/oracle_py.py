"""ctypes view of oracle/libmuse_oracle.so (the CPU restatement of go-muse's
XCorr / Batch.Run path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmuse_oracle.so")

_i64 = ctypes.c_int64
_dp = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


_FAST_SO = os.path.join(_HERE, "libmuse_cpu_fast.so")


def build(force=False):
    for so, src in ((_SO, "muse_oracle.c"), (_FAST_SO, "muse_cpu_fast.c")):
        srcp = os.path.join(_HERE, src)
        if not force and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(srcp):
            continue
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_next_pow2.restype = _i64
        L.oracle_next_pow2.argtypes = [ctypes.c_double]
        L.oracle_znormalize.argtypes = [_dp, _i64]
        L.oracle_zero_pad.restype = _i64
        L.oracle_zero_pad.argtypes = [_dp, _i64, _i64, _dp]
        L.oracle_max_abs_index.restype = _i64
        L.oracle_max_abs_index.argtypes = [_dp, _i64]
        L.oracle_rfft.argtypes = [_dp, _i64, _dp]
        L.oracle_irfft.argtypes = [_dp, _i64, _dp]
        L.oracle_xcorr.argtypes = [_dp, _i64, _dp, _i64, _i64, ctypes.c_int,
                                   _dp, _i64p, _i64p, _dp]
        L.oracle_ref_spectrum.argtypes = [_dp, _i64, _i64, _dp]
        L.oracle_xcorr_with_x.argtypes = [_dp, _dp, _i64, _i64, _dp, _i64p,
                                          _dp, _dp]
        L.oracle_xcorr_direct_ld.argtypes = [_dp, _dp, _i64, _i64, _dp]
        L.oracle_batch_scores.argtypes = [_dp, _dp, _i64, _i64, _i64,
                                          ctypes.c_int, _i32p, _dp, _dp]
        L.oracle_results.restype = _i64
        L.oracle_results.argtypes = [_i32p, _dp, _i64, _i32p, _i64,
                                     ctypes.c_int, _i64, _i64, ctypes.c_double,
                                     ctypes.c_int, _i64p, _i32p, _dp, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def next_pow2(v):
    return int(lib().oracle_next_pow2(float(v)))


def znormalize(x):
    """Returns (z, std_zero_flag); x is not modified."""
    z = _f64(x).copy()
    rc = lib().oracle_znormalize(_d(z), len(z))
    return z, bool(rc)


def zero_pad(x, n):
    x = _f64(x)
    out = np.zeros(max(n, len(x)))
    m = lib().oracle_zero_pad(_d(x), len(x), n, _d(out))
    return out[:m]


def rfft(x):
    x = _f64(x)
    n = len(x)
    c = np.zeros(2 * (n // 2 + 1))
    lib().oracle_rfft(_d(x), n, _d(c))
    return c[0::2] + 1j * c[1::2]


def xcorr(x, y, n, normalize):
    """xCorr (xcorr.go:102): returns (cc or None, lag, mv)."""
    x, y = _f64(x), _f64(y)
    nn = max(n, len(x), len(y))
    cc = np.zeros(nn)
    n_out, lag, mv = _i64(0), _i64(0), ctypes.c_double(0)
    rc = lib().oracle_xcorr(_d(x), len(x), _d(y), len(y), n, int(bool(normalize)),
                            _d(cc), ctypes.byref(n_out), ctypes.byref(lag),
                            ctypes.byref(mv))
    if rc:
        return None, 0, 0.0
    return cc, int(lag.value), float(mv.value)


def ref_spectrum(ref, n=None):
    """NewBatch precompute (muse_batch.go:35-47). Returns (X interleaved, n) or raises."""
    ref = _f64(ref)
    N = len(ref)
    if n is None:
        n = next_pow2(N)
    X = np.zeros(2 * (n // 2 + 1))
    if lib().oracle_ref_spectrum(_d(ref), N, n, _d(X)):
        raise ValueError("Invalid input query, Standard deviation of zero")
    return X, n


def xcorr_with_x(X, y, n, want_cc=True):
    """xCorrWithX (xcorr.go:160): returns (cc or None, lag, mv, gap)."""
    y = _f64(y)
    cc = np.zeros(n)
    lag, mv, gap = _i64(0), ctypes.c_double(0), ctypes.c_double(0)
    rc = lib().oracle_xcorr_with_x(_d(X), _d(y), len(y), n, _d(cc),
                                   ctypes.byref(lag), ctypes.byref(mv),
                                   ctypes.byref(gap))
    if rc:
        return None, 0, 0.0, 0.0
    return (cc if want_cc else None), int(lag.value), float(mv.value), float(gap.value)


def xcorr_direct_ld(ref, y, n):
    ref, y = _f64(ref), _f64(y)
    cc = np.zeros(n)
    rc = lib().oracle_xcorr_direct_ld(_d(ref), _d(y), len(y), n, _d(cc))
    return None if rc else cc


def batch_scores(ref, rows, nthreads=1, want_gap=True):
    """Per-series (lag int32[M], mv f64[M], gap f64[M]) for a row-major M x N matrix."""
    ref = _f64(ref)
    rows = np.asarray(rows, dtype=np.float64)
    assert rows.ndim == 2 and rows.strides[1] == 8
    M, N = rows.shape
    stride = rows.strides[0] // 8 if M > 1 else N
    lag = np.zeros(M, dtype=np.int32)
    mv = np.zeros(M)
    gap = np.zeros(M) if want_gap else None
    rc = lib().oracle_batch_scores(
        _d(ref), rows.ctypes.data_as(_dp), M, N, stride, int(nthreads),
        lag.ctypes.data_as(_i32p), _d(mv), _d(gap) if want_gap else None)
    if rc == 1:
        raise ValueError("Invalid input query, Standard deviation of zero")
    if rc:
        raise ValueError("bad arguments")
    return lag, mv, gap


_fast = None


def build_fast(force=False):
    """(Re)build libmuse_cpu_fast.so on THIS host: it is compiled -march=native, so a copy built elsewhere (the build
    container) must not be trusted on another CPU -- bench.py forces a rebuild before timing."""
    global _fast
    srcp = os.path.join(_HERE, "muse_cpu_fast.c")
    if force or not os.path.exists(_FAST_SO) or os.path.getmtime(_FAST_SO) < os.path.getmtime(srcp):
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_FAST_SO)], stdout=subprocess.DEVNULL)
        _fast = None
    return _FAST_SO


def fast_batch_scores(ref, rows, nthreads=1):
    """The TIMED CPU baseline (muse_cpu_fast.c: same algorithm, radix-4 Stockham FFT, -O3 -march=native): per-series
    (lag int32[M], mv f64[M]).  Built on the box it runs on (-march=native).  Not the checker."""
    global _fast
    if _fast is None:
        build_fast()
        _fast = ctypes.CDLL(_FAST_SO)
        _fast.fast_batch_scores.argtypes = [_dp, _dp, _i64, _i64, _i64, ctypes.c_int, _i32p, _dp]
    ref = _f64(ref)
    rows = np.asarray(rows, dtype=np.float64)
    assert rows.ndim == 2 and rows.strides[1] == 8
    M, N = rows.shape
    stride = rows.strides[0] // 8 if M > 1 else N
    lag = np.zeros(M, dtype=np.int32)
    mv = np.zeros(M)
    rc = _fast.fast_batch_scores(_d(ref), rows.ctypes.data_as(_dp), M, N, stride, int(nthreads),
                                 lag.ctypes.data_as(_i32p), _d(mv))
    if rc == 1:
        raise ValueError("Invalid input query, Standard deviation of zero")
    if rc:
        raise ValueError("bad arguments")
    return lag, mv


def results(lag, mv, group_id=None, G=0, abs_scores=True, max_lag=10, top_n=20,
            threshold=0.0, sign_filter=0):
    """Batch.Run post-processing + Results.Update/Fetch.
    Returns (series idx[], lag[], score[], mean_abs) in Fetch order."""
    lag = np.ascontiguousarray(lag, dtype=np.int32)
    mv = _f64(mv)
    M = len(mv)
    gid = None
    if group_id is not None:
        gid = np.ascontiguousarray(group_id, dtype=np.int32)
    cap = max(int(top_n), 1)
    o_s = np.zeros(cap, dtype=np.int64)
    o_l = np.zeros(cap, dtype=np.int32)
    o_v = np.zeros(cap)
    mean = ctypes.c_double(0)
    cnt = lib().oracle_results(
        lag.ctypes.data_as(_i32p), _d(mv), M,
        gid.ctypes.data_as(_i32p) if gid is not None else None, int(G),
        int(bool(abs_scores)), int(max_lag), int(top_n), float(threshold),
        int(sign_filter), o_s.ctypes.data_as(_i64p), o_l.ctypes.data_as(_i32p),
        _d(o_v), ctypes.byref(mean))
    return o_s[:cnt].copy(), o_l[:cnt].copy(), o_v[:cnt].copy(), float(mean.value)
