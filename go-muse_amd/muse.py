"""Host-side mirror of go-muse's exported API for the Batch.Run / Muse.Run path,
driving libmuse_hip.so through its C ABI.

Names, argument meaning and error behaviour follow the reference
(/root/reference): NewLabels/Labels (labels.go), NewSeries/Series (series.go),
NewGroup/Group (group.go), NewResults/Results/Score (results.go, scores.go),
NewBatch/Batch.Run (muse_batch.go:23,99), New/Muse.Run (muse.go:23,46).  Go's
`(value, error)` returns become Python exceptions: MuseError for the errors the
reference returns, ValueError for Group.Add's.  All arithmetic happens on the
GPU; this layer does label bookkeeping only (the reference's L2).

One documented difference: the reference's zNormalize overwrites the caller's
slices in place (xcorr.go:86,93); this engine never mutates user
data, so every Run behaves like the reference's FIRST Run on fresh inputs.
"""
import ctypes
import math
import sys
import threading
import uuid

import numpy as np

from . import binding as B
from .binding import MuseError

DefaultLabel = "uid"  # labels.go:7

SignFilter_POS = 1   # results.go:22-26
SignFilter_NEG = -1
SignFilter_ANY = 0


# ------------------------------------------------------------------ engine
class Engine:
    """One muse_ctx = one GPU."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        B.check(B.load().muse_ctx_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            B.load().muse_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            if not sys.is_finalizing():   # at exit the HIP runtime tears itself down
                self.close()
        except Exception:
            pass

    def synchronize(self):
        B.check(B.load().muse_ctx_synchronize(self._h))

    def device_info(self):
        name = ctypes.create_string_buffer(128)
        cus, hbm = ctypes.c_int32(0), ctypes.c_int64(0)
        B.check(B.load().muse_ctx_device_info(self._h, name, 128, ctypes.byref(cus), ctypes.byref(hbm)))
        return name.value.decode(), int(cus.value), int(hbm.value)

    def set_kernel(self, variant):
        B.check(B.load().muse_ctx_set_kernel(self._h, int(variant)))

    def set_screening(self, enable, min_rows=None):
        """OPT-IN filter-and-refine Run (include/muse_hip.h: muse_ctx_set_screening); min_rows: smallest group it is used for"""
        B.check(B.load().muse_ctx_set_screening(self._h, (int(min_rows) if min_rows and min_rows > 1 else 1) if enable else 0))

    def set_screen_bound_scale(self, scale):
        """test hook (include/muse_hip_test.h): scales the error bound the filter-and-refine Run assumes"""
        B.check(B.load().muse_test_set_screen_bound_scale(self._h, float(scale)))

    def trim(self):
        """muse_ctx_trim: hands the context's cached device / pinned blocks back to the system"""
        B.check(B.load().muse_ctx_trim(self._h))

    def pool_stats(self):
        """test hook: (device idle bytes, device idle blocks, pinned idle bytes, pinned idle blocks) of the allocation cache"""
        v = [ctypes.c_int64(0) for _ in range(4)]
        B.check(B.load().muse_test_pool_stats(self._h, *[ctypes.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def xcorr_repeat(self, repeat):
        """measurement hook: muse_xcorr_groups launches its kernel `repeat` times back to back"""
        B.check(B.load().muse_test_xcorr_repeat(self._h, int(repeat)))

    def huge_batch_mb(self, megabytes):
        """measurement hook: work buffer of one batch of the long-series pass (0 = built-in 128 MB)"""
        B.check(B.load().muse_test_huge_batch_mb(self._h, int(megabytes)))

    def rows_always_copy(self, on):
        """test hook: muse_batch_run_rows copies even the smallest groups to HBM instead of letting the kernel read the pinned buffer"""
        B.check(B.load().muse_test_rows_always_copy(self._h, 1 if on else 0))

    def wave_argmax(self, cc_a, cc_b):
        """test hook: the n = 4096 kernels' per-wave argmax step on 2 x 4096 given values; (4, 2, 3) array of
        {max |cc|, signed value, index} per wave and series"""
        a = np.ascontiguousarray(cc_a, dtype=np.float64)
        b = np.ascontiguousarray(cc_b, dtype=np.float64)
        if a.shape != (4096,) or b.shape != (4096,):
            raise ValueError("wave_argmax: two vectors of 4096 values")
        out = np.zeros(24)
        B.check(B.load().muse_test_wave_argmax(self._h, B.dptr(a), B.dptr(b), B.dptr(out)))
        return out.reshape(4, 2, 3)

    def clock_probe_start(self, window_ms=1.0, total_ms=1000.0):
        """measurement hook: a one-wave kernel samples the shader clock for total_ms while the caller's kernels run"""
        B.check(B.load().muse_test_clock_probe_start(self._h, float(window_ms), float(total_ms)))

    def clock_probe_stop(self):
        B.check(B.load().muse_test_clock_probe_stop(self._h))

    def clock_probe_read(self):
        """waits for the probe; the clock of every window in MHz"""
        mhz = np.zeros(4096)
        n = ctypes.c_int32(0)
        B.check(B.load().muse_test_clock_probe_read(self._h, B.dptr(mhz), len(mhz), ctypes.byref(n)))
        return mhz[:int(n.value)].copy()

    def kernel_name(self, dbatch):
        """name of the kernel automatic selection takes for this batch's all-scores pass"""
        name = ctypes.create_string_buffer(128)
        B.check(B.load().muse_batch_kernel_name(dbatch._h, name, 128))
        return name.value.decode()

    def kernel_timing(self, enable):
        B.check(B.load().muse_ctx_kernel_timing(self._h, 1 if enable else 0))

    def kernel_time(self):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        B.check(B.load().muse_ctx_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return float(ms.value), int(cnt.value)

    def redo_time(self):
        """(ms, brackets) of the launches that redo listed pairs behind a fused launch: what kernel_time() leaves out"""
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        B.check(B.load().muse_ctx_redo_time(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return float(ms.value), int(cnt.value)

    def pci_bus_id(self):
        buf = ctypes.create_string_buffer(32)
        B.check(B.load().muse_ctx_device_pci_bus_id(self._h, buf, 32))
        return buf.value.decode()

    # single-pair entry points (xcorr_test.go-style known-answer access)
    def xcorr_with_x(self, ref, y, n=None):
        ref, y = B.as_f64(ref), B.as_f64(y)
        N = len(y)
        if n is None:
            n = next_pow2(N)
        cc = np.zeros(max(n, 1))
        lag, nil, mv = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(0)
        B.check(B.load().muse_xcorr_with_x(self._h, B.dptr(ref), B.dptr(y), N, int(n), B.dptr(cc),
                                           ctypes.byref(lag), ctypes.byref(mv), ctypes.byref(nil)))
        if nil.value:
            return None, 0, 0.0
        return cc, int(lag.value), float(mv.value)

    def xcorr(self, x, y, n, normalize):
        x, y = B.as_f64(x), B.as_f64(y)
        nn = max(int(n), len(x), len(y))
        cc = np.zeros(nn)
        lag, nil, mv = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(0)
        B.check(B.load().muse_xcorr(self._h, B.dptr(x), len(x), B.dptr(y), len(y), int(n),
                                    1 if normalize else 0, B.dptr(cc), ctypes.byref(lag),
                                    ctypes.byref(mv), ctypes.byref(nil)))
        if nil.value:
            return None, 0, 0.0
        return cc, int(lag.value), float(mv.value)


    def xcorr_batch(self, x_rows, y_rows, n, normalize, want_cc=False):
        """xCorr (xcorr.go:102-153) for M independent pairs in one launch: x_rows is M x lenx, y_rows M x leny.
        Returns (lag[M], mv[M], is_nil[M]) or, with want_cc, (cc[M, n], lag, mv, is_nil)."""
        x, y = np.ascontiguousarray(x_rows, dtype=np.float64), np.ascontiguousarray(y_rows, dtype=np.float64)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("x_rows and y_rows must be 2-D with one row per pair")
        M = x.shape[0]
        nn = max(int(n), x.shape[1], y.shape[1])
        lag, mv, nil = np.zeros(M, dtype=np.int32), np.zeros(M), np.zeros(M, dtype=np.int32)
        cc = np.zeros((M, nn)) if want_cc else None
        B.check(B.load().muse_xcorr_batch(self._h, B.dptr(x), B.dptr(y), M, x.shape[1], y.shape[1], int(n),
                                          1 if normalize else 0, B.i32ptr(lag), B.dptr(mv), B.i32ptr(nil),
                                          B.dptr(cc) if want_cc else None))
        return (cc, lag, mv, nil) if want_cc else (lag, mv, nil)


def xcorr_groups(gx, gy, n, normalize, want_cc=False):
    """The same over two device-resident groups (row i of gx with row i of gy): nothing is uploaded."""
    M = gx.M
    nn = max(int(n), gx.N, gy.N)
    lag, mv, nil = np.zeros(M, dtype=np.int32), np.zeros(M), np.zeros(M, dtype=np.int32)
    cc = np.zeros((M, nn)) if want_cc else None
    B.check(B.load().muse_xcorr_groups(gx._h, gy._h, int(n), 1 if normalize else 0, B.i32ptr(lag), B.dptr(mv),
                                       B.i32ptr(nil), B.dptr(cc) if want_cc else None))
    return (cc, lag, mv, nil) if want_cc else (lag, mv, nil)


_engines = {}


def get_engine(device=0):
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]


def next_pow2(val):
    """nextPowOf2, xcorr.go:19-24 (host-side scalar, same floating formula)."""
    return int(B.load().muse_next_pow2(float(val)))


class DeviceGroup:
    """Device-resident M x N matrix (muse_group): float64 storage by default; f32=True keeps the rows as float32 in HBM
    (muse_group_create_f32, opt-in: half the bytes per Run, samples rounded on the way in, arithmetic still float64)."""

    def __init__(self, engine, N, capacity=0, f32=False):
        self.engine = engine
        self._h = ctypes.c_void_p()
        create = B.load().muse_group_create_f32 if f32 else B.load().muse_group_create
        B.check(create(engine._h, int(capacity), int(N), ctypes.byref(self._h)))
        self.N = int(N)
        self.f32 = bool(f32)

    @classmethod
    def from_rows(cls, engine, rows, f32=False):
        rows = np.asarray(rows, dtype=np.float64)
        if rows.ndim != 2:
            raise ValueError("rows must be 2-D")
        g = cls(engine, rows.shape[1], rows.shape[0], f32=f32)
        g.append(rows)
        return g

    @classmethod
    def synthetic(cls, engine, M, N, seed=0x6D757365, global_first=0, copies=True, constants=True, f32=False):
        """rect+noise workload generated on the device; returns (group, ref).  copies / constants: plant the
        1-in-1024 exact copies of the reference / constant rows (muse_hip.h, MUSE_SYNTH_NO_*)."""
        g = cls(engine, N, M, f32=f32)
        ref = np.zeros(N)
        flags = (0 if copies else 1) | (0 if constants else 2)
        B.check(B.load().muse_group_fill_synthetic(g._h, 0, int(M), int(global_first),
                                                   ctypes.c_uint64(seed), ctypes.c_uint32(flags), B.dptr(ref)))
        return g, ref

    def append(self, rows):
        rows = np.asarray(rows, dtype=np.float64)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.shape[0] == 0:
            return
        if rows.strides[1] != 8:
            rows = np.ascontiguousarray(rows)
        if rows.shape[1] != self.N:
            raise MuseError(B.MUSE_ERR_LENGTH, "Timeseries has length %d, but current group has length %d"
                            % (rows.shape[1], self.N))
        stride = rows.strides[0] // 8 if rows.shape[0] > 1 else self.N
        B.check(B.load().muse_group_append(self._h, rows.ctypes.data_as(B._dp), rows.shape[0], stride))

    def stage(self, count):
        """muse_group_stage: a window of pinned host memory for up to `count` more rows -> a (granted, N) float64 array VIEW of
        it; fill it, then commit(first, k) every piece (any order, each row once).  The rows join the group with the last commit."""
        win, granted = B._dp(), ctypes.c_int64(0)
        B.check(B.load().muse_group_stage(self._h, int(count), ctypes.byref(win), ctypes.byref(granted)))
        k = int(granted.value)
        if k == 0:
            return np.zeros((0, self.N))
        return np.ctypeslib.as_array(win, shape=(k, self.N))

    def commit(self, first, count):
        B.check(B.load().muse_group_commit(self._h, int(first), int(count)))

    def append_staged(self, series, piece_rows=None):
        """Group.Add for a list of separate per-series arrays: each is copied ONCE, straight into the pinned window, and the
        window goes out piece by piece beside the copying (the host mirrors' upload path)"""
        i, total = 0, len(series)
        while i < total:
            win = self.stage(total - i)
            k = win.shape[0]
            step = piece_rows or max(1, (512 << 10) // (8 * self.N))
            for lo in range(0, k, step):
                hi = min(k, lo + step)
                for r in range(lo, hi):
                    win[r, :] = series[i + r]
                self.commit(lo, hi - lo)
            i += k

    @property
    def M(self):
        m, n = ctypes.c_int64(0), ctypes.c_int32(0)
        B.check(B.load().muse_group_shape(self._h, ctypes.byref(m), ctypes.byref(n)))
        return int(m.value)

    def read(self, first, count):
        out = np.zeros((int(count), self.N))
        B.check(B.load().muse_group_read(self._h, int(first), int(count), B.dptr(out)))
        return out

    def close(self):
        if self._h:
            B.load().muse_group_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            if not sys.is_finalizing():   # at exit the HIP runtime tears itself down
                self.close()
        except Exception:
            pass


class DeviceBatch:
    """muse_batch: resident reference spectrum + per-series result buffers."""

    def __init__(self, engine, dgroup, ref, like=None):
        self.engine, self.dgroup = engine, dgroup
        self._h = ctypes.c_void_p()
        if like is not None:    # same reference, another group: shares the spectrum tables (muse_batch_create_like)
            B.check(B.load().muse_batch_create_like(like._h, dgroup._h, ctypes.byref(self._h)))
        else:
            ref = B.as_f64(ref)
            B.check(B.load().muse_batch_create(engine._h, dgroup._h, B.dptr(ref), len(ref), ctypes.byref(self._h)))
        n = ctypes.c_int32(0)
        B.check(B.load().muse_batch_fft_len(self._h, ctypes.byref(n)))
        self.n = int(n.value)

    @classmethod
    def like(cls, template, dgroup):
        """a batch for `dgroup` against the template's reference (no transform, two small allocations)"""
        return cls(template.engine, dgroup, None, like=template)

    def spectrum(self):
        out = np.zeros(2 * (self.n // 2 + 1))
        B.check(B.load().muse_batch_spectrum(self._h, B.dptr(out)))
        return out[0::2] + 1j * out[1::2]

    def score(self):
        """enqueue the fused kernel (asynchronous)"""
        B.check(B.load().muse_batch_score(self._h))

    def scores(self):
        M = self.dgroup.M
        lag = np.zeros(M, dtype=np.int32)
        mv = np.zeros(M)
        B.check(B.load().muse_batch_scores(self._h, B.i32ptr(lag), B.dptr(mv)))
        return lag, mv

    def read_scores(self):
        """(lag, mv) of the last scoring pass without re-scoring (muse_batch_read_scores)"""
        M = self.dgroup.M
        lag = np.zeros(M, dtype=np.int32)
        mv = np.zeros(M)
        B.check(B.load().muse_batch_read_scores(self._h, B.i32ptr(lag), B.dptr(mv)))
        return lag, mv

    def last_run_info(self):
        """(screened, refined_pairs) of the last run on this batch (muse_batch_last_run_info)"""
        scr, ref = ctypes.c_int32(0), ctypes.c_int64(0)
        B.check(B.load().muse_batch_last_run_info(self._h, ctypes.byref(scr), ctypes.byref(ref)))
        return bool(scr.value), int(ref.value)

    RUN_PATHS = {0: "fp64", 1: "screened", 2: "fp64 (these filters were costly to screen)", 3: "fp64 (guard tripped)"}

    def last_run_path(self):
        """MUSE_RUN_PATH_* of the last run on this batch (muse_batch_last_run_path)"""
        path = ctypes.c_int32(0)
        B.check(B.load().muse_batch_last_run_path(self._h, ctypes.byref(path)))
        return int(path.value)

    def screen_estimates(self, max_lag=10):
        """test hook: (estimates, flags, E) of the fp32 screening pass (include/muse_hip_test.h)"""
        M = self.dgroup.M
        est = np.zeros(M)
        flags = np.zeros(M, dtype=np.uint32)
        E = ctypes.c_double(0)
        B.check(B.load().muse_batch_screen_estimates(self._h, int(max_lag), B.dptr(est),
                                                      flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.byref(E)))
        return est, flags, float(E.value)

    def run(self, group_id=None, G=0, max_lag=10, top_n=20, threshold=0.0, sign_filter=0, abs_scores=True):
        cap = max(int(top_n), 1)
        o_s = np.zeros(cap, dtype=np.int64)
        o_l = np.zeros(cap, dtype=np.int32)
        o_v = np.zeros(cap)
        cnt, mean = ctypes.c_int32(0), ctypes.c_double(0)
        gid = None
        if group_id is not None:
            gid = np.ascontiguousarray(group_id, dtype=np.int32)
        B.check(B.load().muse_batch_run(
            self._h, B.i32ptr(gid) if gid is not None else None, int(G), int(max_lag), int(top_n),
            float(threshold), int(sign_filter), 1 if abs_scores else 0,
            B.i64ptr(o_s), B.i32ptr(o_l), B.dptr(o_v), ctypes.byref(cnt), ctypes.byref(mean)))
        c = cnt.value
        return o_s[:c].copy(), o_l[:c].copy(), o_v[:c].copy(), float(mean.value)

    def run_shard(self, group_id=None, G=0, series_offset=0, max_lag=10, top_n=20, threshold=0.0,
                  sign_filter=0, abs_scores=True):
        """this shard's top-N candidates as a RECORD_DTYPE array (<= top_n)"""
        rec = np.zeros(max(int(top_n), 1), dtype=B.RECORD_DTYPE)
        cnt = ctypes.c_int32(0)
        gid = None
        if group_id is not None:
            gid = np.ascontiguousarray(group_id, dtype=np.int32)
        B.check(B.load().muse_batch_run_shard(
            self._h, B.i32ptr(gid) if gid is not None else None, int(G), int(series_offset), int(max_lag),
            int(top_n), float(threshold), int(sign_filter), 1 if abs_scores else 0,
            B.recptr(rec), ctypes.byref(cnt)))
        return rec[:cnt.value].copy()

    def run_rows(self, rows, abs_scores=False):
        """muse_batch_run_rows: Muse.Run (muse.go:46-92) in one call -- `rows` (M x N, host) form ONE label group scored against
        this batch's reference (the batch's own group is not touched); -> (winner record, state) as run_groups reports them"""
        rows = np.asarray(rows, dtype=np.float64)
        if rows.ndim != 2:
            raise ValueError("rows must be 2-D")
        if rows.shape[0] and rows.strides[1] != 8:
            rows = np.ascontiguousarray(rows)
        stride = rows.strides[0] // 8 if rows.shape[0] > 1 else rows.shape[1]
        rec = np.zeros(1, dtype=B.RECORD_DTYPE)
        state = ctypes.c_uint8(0)
        B.check(B.load().muse_batch_run_rows(self._h, rows.ctypes.data_as(B._dp), rows.shape[0], stride,
                                             1 if abs_scores else 0, B.recptr(rec), ctypes.byref(state)))
        return rec[0], int(state.value)

    def run_row_ptrs(self, series, abs_scores=False):
        """muse_batch_run_row_ptrs: the same with one pointer per series (a list of separate contiguous float64 arrays of the
        reference's length): gathered straight into the call's pinned buffer"""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in series]
        for a in arrs:
            if a.ndim != 1 or a.shape[0] != self.dgroup.N:
                raise MuseError(B.MUSE_ERR_LENGTH, "Encountered a comparison graph with differing length than the reference")
        ptrs = (B._dp * max(len(arrs), 1))(*[a.ctypes.data_as(B._dp) for a in arrs])
        rec = np.zeros(1, dtype=B.RECORD_DTYPE)
        state = ctypes.c_uint8(0)
        B.check(B.load().muse_batch_run_row_ptrs(self._h, ptrs, len(arrs), 1 if abs_scores else 0, B.recptr(rec), ctypes.byref(state)))
        return rec[0], int(state.value)

    def run_groups(self, group_id, G, series_offset=0, abs_scores=True):
        """this shard's winner per label group, unfiltered (muse_batch_run_groups): (records[G], state[G])"""
        rec = np.zeros(max(int(G), 1), dtype=B.RECORD_DTYPE)
        state = np.zeros(max(int(G), 1), dtype=np.uint8)
        gid = np.ascontiguousarray(group_id, dtype=np.int32)
        B.check(B.load().muse_batch_run_groups(self._h, B.i32ptr(gid), int(G), int(series_offset), 1 if abs_scores else 0,
                                               B.recptr(rec), state.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
        return rec[:int(G)], state[:int(G)]

    def close(self):
        if self._h:
            B.load().muse_batch_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            if not sys.is_finalizing():   # at exit the HIP runtime tears itself down
                self.close()
        except Exception:
            pass


def _batch_handles(batches):
    batches = list(batches)
    if not batches:
        raise ValueError("empty batch list")
    arr = (ctypes.c_void_p * len(batches))(*[b._h for b in batches])
    return batches, arr


def score_many(batches):
    """muse_batch_score_many: R references against one resident group in ONE pass over the rows
    (asynchronous; results land in each DeviceBatch's own buffers)."""
    batches, arr = _batch_handles(batches)
    B.check(B.load().muse_batch_score_many(arr, len(batches)))


def scores_many(batches):
    """score_many + copy back: list of (lag, mv), one per batch"""
    batches, arr = _batch_handles(batches)
    B.check(B.load().muse_batch_score_many(arr, len(batches)))
    out = []
    lib = B.load()
    for b in batches:
        M = b.dgroup.M
        lag = np.zeros(M, dtype=np.int32)
        mv = np.zeros(M)
        if M:
            B.check(lib.muse_batch_read_scores(b._h, B.i32ptr(lag), B.dptr(mv)))
        out.append((lag, mv))
    return out


def run_many(batches, group_id=None, G=0, max_lag=10, top_n=20, threshold=0.0, sign_filter=0, abs_scores=True):
    """muse_batch_run_many: Batch.Run for every reference with one pass over the rows;
    returns a list of (series, lag, score, mean_abs), one per batch"""
    batches, arr = _batch_handles(batches)
    R = len(batches)
    cap = max(int(top_n), 1)
    o_s = np.zeros(R * cap, dtype=np.int64)
    o_l = np.zeros(R * cap, dtype=np.int32)
    o_v = np.zeros(R * cap)
    cnt = np.zeros(R, dtype=np.int32)
    mean = np.zeros(R)
    gid = None
    if group_id is not None:
        gid = np.ascontiguousarray(group_id, dtype=np.int32)
    B.check(B.load().muse_batch_run_many(
        arr, R, B.i32ptr(gid) if gid is not None else None, int(G), int(max_lag), int(top_n),
        float(threshold), int(sign_filter), 1 if abs_scores else 0,
        B.i64ptr(o_s), B.i32ptr(o_l), B.dptr(o_v), B.i32ptr(cnt), B.dptr(mean)))
    tn = max(int(top_n), 0)
    res = []
    for r in range(R):
        c = int(cnt[r])
        res.append((o_s[r * tn:r * tn + c].copy(), o_l[r * tn:r * tn + c].copy(), o_v[r * tn:r * tn + c].copy(),
                    float(mean[r])))
    return res


def device_count():
    n = ctypes.c_int32(0)
    B.check(B.load().muse_device_count(ctypes.byref(n)))
    return int(n.value)


def merge_group_records(records, state, max_lag, top_n, threshold, sign_filter):
    """muse_merge_group_records: records / state are (n_shards, G) arrays, shards in ascending row order"""
    records = np.ascontiguousarray(records, dtype=B.RECORD_DTYPE)
    state = np.ascontiguousarray(state, dtype=np.uint8)
    W, G = records.shape
    cap = max(int(top_n), 1)
    o_s, o_l, o_v = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int32), np.zeros(cap)
    cnt, mean = ctypes.c_int32(0), ctypes.c_double(0)
    B.check(B.load().muse_merge_group_records(B.recptr(records), state.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), W, G,
                                              int(max_lag), int(top_n), float(threshold), int(sign_filter),
                                              B.i64ptr(o_s), B.i32ptr(o_l), B.dptr(o_v), ctypes.byref(cnt), ctypes.byref(mean)))
    c = cnt.value
    return o_s[:c].copy(), o_l[:c].copy(), o_v[:c].copy(), float(mean.value)


def merge_group_winners(records, state):
    """muse_merge_group_winners: (n_shards, G) records / states -> the G label groups' winners and states (0 no member, 1 a
    Score, 2 the group's score is NaN): what Batch.Run feeds through Results.Update in group order"""
    records = np.ascontiguousarray(records, dtype=B.RECORD_DTYPE)
    state = np.ascontiguousarray(state, dtype=np.uint8)
    W, G = records.shape
    out, ost = np.zeros(max(G, 1), dtype=B.RECORD_DTYPE), np.zeros(max(G, 1), dtype=np.uint8)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    B.check(B.load().muse_merge_group_winners(B.recptr(records), state.ctypes.data_as(u8), W, G, B.recptr(out), ost.ctypes.data_as(u8)))
    return out[:G], ost[:G]


def merge_records(records, top_n):
    """Results.Update/Fetch over gathered shard candidates (host only)."""
    records = np.ascontiguousarray(records, dtype=B.RECORD_DTYPE)
    cap = max(int(top_n), 1)
    o_s = np.zeros(cap, dtype=np.int64)
    o_l = np.zeros(cap, dtype=np.int32)
    o_v = np.zeros(cap)
    cnt, mean = ctypes.c_int32(0), ctypes.c_double(0)
    B.check(B.load().muse_merge_records(B.recptr(records), len(records), int(top_n), B.i64ptr(o_s),
                                        B.i32ptr(o_l), B.dptr(o_v), ctypes.byref(cnt), ctypes.byref(mean)))
    c = cnt.value
    return o_s[:c].copy(), o_l[:c].copy(), o_v[:c].copy(), float(mean.value)


# --------------------------------------------------- labels.go / series.go
class Labels:
    def __init__(self, label_map):
        self.labels = dict(label_map)
        self.keys = sorted(self.labels)          # labels.go:21-29

    def Len(self):
        return len(self.labels)

    def Keys(self):
        return self.keys

    def Get(self, key):                           # labels.go:44-49
        if key in self.labels:
            return self.labels[key], True
        return "", False

    def ID(self, labels=None):                    # labels.go:54-73
        if not labels:
            labels = self.Keys()
        else:
            labels.sort()                         # labels.go:59 sorts the caller's slice
        return ",".join("%s:%s" % (k, self.labels[k]) for k in labels if k in self.labels)

    def __eq__(self, other):
        return isinstance(other, Labels) and self.labels == other.labels

    def __repr__(self):
        return "Labels(%r)" % (self.labels,)


def NewLabels(label_map):
    return Labels(label_map)


class Series:
    def __init__(self, y, labels=None):           # series.go:15-21
        if labels is None or labels.Len() == 0:
            labels = NewLabels({DefaultLabel: str(uuid.uuid4())})
        self.y = np.asarray(y, dtype=np.float64)
        self.labels = labels

    def Length(self):
        return int(self.y.shape[0])

    def Values(self):
        return self.y

    def Labels(self):
        return self.labels

    def UID(self):                                # series.go:40-42
        return self.labels.ID(self.labels.Keys())


def NewSeries(y, labels=None):
    return Series(y, labels)


# ---------------------------------------------------------------- group.go
class _Registry(dict):
    """Group.registry: uid -> Series, insertion ordered, and PUBLIC (the reference's field).  Every mutation bumps `version`,
    which is what the cached partition of indexLabelValues (and the device copy of the rows) is keyed by: a series replaced or
    removed and re-added at the same count is seen.  (A Series' labels and values are taken as immutable once it is added.)"""

    def __init__(self):
        super().__init__()
        self.version = 0

    def _bump(name):                              # noqa: N805 (a decorator factory evaluated in the class body)
        def method(self, *a, **kw):
            self.version += 1
            return getattr(dict, name)(self, *a, **kw)
        method.__name__ = name
        return method
    for _m in ("__setitem__", "__delitem__", "pop", "popitem", "clear", "update", "setdefault", "__ior__"):
        locals()[_m] = _bump(_m)
    del _m, _bump


class Group:
    def __init__(self, name):
        self.Name = name
        self.n = 0
        self.index = {}
        self.registry = _Registry()               # uid -> Series, insertion ordered
        self._dev = None                          # (engine, DeviceGroup, rows uploaded)

    def Length(self):
        return self.n

    def Add(self, *series):                       # group.go:31-56
        for s in series:
            if len(s.labels.Keys()) == 0:
                raise ValueError("Invalid Series with no labels, %r" % (s,))
            uid = s.UID()
            if uid in self.registry:
                raise ValueError("Series with label:values, %s, already exists within group, %s"
                                 % (uid, self.Name))
            if len(self.registry) == 0:
                self.n = s.Length()
            elif s.Length() != self.n:
                raise ValueError("Timeseries has length %d, but current group has length %d"
                                 % (s.Length(), self.n))
            self.registry[uid] = s

    def FilterByLabelValues(self, labels):        # group.go:60-71
        guid = labels.ID(labels.Keys())
        return [self.registry[u] for u in self.index.get(guid, [])]

    def indexLabelValues(self, groupByLabels):    # group.go:76-104
        groupByLabels = list(groupByLabels) if groupByLabels else []
        # the partition depends on the label names and the series alone, and series are only ever added: a repeated Run with the
        # same grouping over the same series reuses it (and the group ids derived from it: _group_ids)
        key = (getattr(self.registry, "version", None), len(self.registry), tuple(groupByLabels))
        if key[0] is not None and getattr(self, "_index_key", None) == key and self.registry:   # (a plain dict put in its place: no cache)
            return self._index_distinct
        distinct = []
        self.index = {}
        self._gid_cache = None
        for uid, s in self.registry.items():
            if len(groupByLabels) != 0:
                guid = s.labels.ID(groupByLabels)
            else:
                guid = uid
                groupByLabels = list(s.Labels().Keys())   # group.go:88
            if guid not in self.index:
                lv = {}
                for name in groupByLabels:
                    v, ok = s.labels.Get(name)
                    if ok:
                        lv[name] = v
                distinct.append(NewLabels(lv))
                self.index[guid] = []
            self.index[guid].append(uid)
        self._index_key, self._index_distinct = key, distinct
        return distinct

    def _group_ids(self):
        """int32 group id of every series (insertion order) for the partition indexLabelValues built last"""
        if getattr(self, "_gid_cache", None) is None:
            uid_pos = {uid: i for i, uid in enumerate(self.registry)}
            gid = np.zeros(len(uid_pos), dtype=np.int32)
            for g, uids in enumerate(self.index.values()):   # same order as the distinct label values
                for u in uids:
                    gid[uid_pos[u]] = g
            self._gid_cache = gid
        return self._gid_cache

    # --- device residency: the matrix is uploaded once and appended to
    def _series_list(self):
        return list(self.registry.values())

    def _appended_only(self, seen_version, seen_count):
        """True when everything that happened to the (public) registry since (seen_version, seen_count) was the adding of new series"""
        v = getattr(self.registry, "version", None)
        return v is not None and len(self.registry) >= seen_count and v - seen_version == len(self.registry) - seen_count

    def _device_group(self, engine):
        ser = self._series_list()
        stale = self._dev is not None and not self._appended_only(self._dev[3], self._dev[2])
        if self._dev is None or self._dev[0] is not engine or stale:
            if self._dev is not None and stale:
                self._dev[1].close()              # a series was replaced or removed: the resident rows are rebuilt
            self._dev = [engine, DeviceGroup(engine, self.n, len(ser)), 0, getattr(self.registry, "version", 0) - len(ser)]
        _, dg, done, _ = self._dev
        if done < len(ser):
            dg.append(np.stack([s.y for s in ser[done:]]))
        self._dev[2] = len(ser)
        self._dev[3] = getattr(self.registry, "version", 0)
        return dg


    def _device_shards(self, engines):
        """one contiguous row range per engine (dist.shard_bounds: equal shares rounded up to an even row count), cut once at
        the first sharded Run; series added later extend the last shard.  Returns [(engine, DeviceGroup, lo, hi)]."""
        from .dist import shard_bounds
        ser = self._series_list()
        sh = getattr(self, "_shards", None)
        if sh is None or len(sh) != len(engines) or any(s[0] is not e for s, e in zip(sh, engines)):
            sh = []
            for r, e in enumerate(engines):
                lo, hi = shard_bounds(len(ser), len(engines), r)
                sh.append([e, DeviceGroup(e, self.n, hi - lo), lo, hi, 0])
            self._shards = sh
        if sh:
            sh[-1][3] = len(ser)
        for s in sh:
            e, dg, lo, hi, done = s
            if lo + done < hi:
                dg.append(np.stack([x.y for x in ser[lo + done:hi]]))
                s[4] = hi - lo
        return [(s[0], s[1], s[2], s[3]) for s in sh]


def NewGroup(name):
    return Group(name)


# --------------------------------------------------- scores.go / results.go
class Score:
    __slots__ = ("Labels", "Lag", "PercentScore")

    def __init__(self, Labels=None, Lag=0, PercentScore=0.0):
        self.Labels, self.Lag, self.PercentScore = Labels, Lag, PercentScore

    def __repr__(self):
        return "Score(%r, Lag=%d, PercentScore=%.6f)" % (self.Labels, self.Lag, self.PercentScore)


class Results:
    """results.go:11-87.  The heap is Go's container/heap on |PercentScore|."""

    def __init__(self, maxLag, topN, threshold, signFilter):
        self.MaxLag, self.TopN, self.Threshold, self.SignFilter = maxLag, topN, threshold, signFilter
        self.scores = []
        self._mu = threading.Lock()               # results.go:12: Muse.Run is called from many goroutines

    def _less(self, i, j):                        # scores.go:25-27
        return abs(self.scores[i].PercentScore) < abs(self.scores[j].PercentScore)

    def _up(self, j):
        while True:
            i = (j - 1) // 2 if j > 0 else 0
            if i == j or not self._less(j, i):
                break
            self.scores[i], self.scores[j] = self.scores[j], self.scores[i]
            j = i

    def _down(self, i0, n):
        i = i0
        while True:
            j1 = 2 * i + 1
            if j1 >= n:
                break
            j = j1
            if j1 + 1 < n and self._less(j1 + 1, j1):
                j = j1 + 1
            if not self._less(j, i):
                break
            self.scores[i], self.scores[j] = self.scores[j], self.scores[i]
            i = j

    def _push(self, s):
        self.scores.append(s)
        self._up(len(self.scores) - 1)

    def _pop(self):
        n = len(self.scores) - 1
        self.scores[0], self.scores[n] = self.scores[n], self.scores[0]
        self._down(0, n)
        return self.scores.pop()

    def passed(self, s):                          # results.go:46-52
        return (abs(float(s.Lag)) <= float(self.MaxLag)
                and abs(s.PercentScore) >= self.Threshold
                and (self.SignFilter == SignFilter_ANY
                     or (s.PercentScore > 0 and self.SignFilter == SignFilter_POS)
                     or (s.PercentScore < 0 and self.SignFilter == SignFilter_NEG)))

    def Update(self, s):                          # results.go:55-72
        if s.Labels is None:
            return
        with self._mu:
            if self.passed(s):
                if len(self.scores) == self.TopN:
                    if self.TopN > 0 and abs(s.PercentScore) > abs(self.scores[0].PercentScore):
                        self._pop()
                        self._push(s)
                else:
                    self._push(s)

    def Fetch(self):                              # results.go:75-87
        with self._mu:
            num = len(self.scores)
            out = [None] * num
            total = 0.0
            for i in range(num - 1, -1, -1):
                sc = self._pop()
                total += abs(sc.PercentScore)
                out[i] = sc
            return out, (total / num if num else math.nan)


def NewResults(maxLag, topN, threshold, signFilter):
    return Results(maxLag, topN, threshold, signFilter)


# ----------------------------------------------------------- muse_batch.go
# Batch.Run feeds Results one Score per label group (the reference's feed) up to this many groups; beyond it the device
# pre-selects the TopN candidates
EXACT_FEED_MAX_GROUPS = 65536


def feed_group_winners(results, winners, state, labels_of):
    """Batch.Run's ordered drain (muse_batch.go:124-128): ONE Score per label group, in group order, through the unchanged
    Results.Update -- the reference's heap history.  winners / state: what muse_merge_group_winners returns (state 1 = the
    group's Score; 0 no member; 2 the group's score is NaN, which never passes).  labels_of(series index, group id) -> Labels.
    Returns the number of Scores fed."""
    r = results
    live = (state == 1) & (np.abs(winners["lag"].astype(np.int64)) <= r.MaxLag) & (np.abs(winners["score"]) >= r.Threshold)
    if r.SignFilter == SignFilter_POS:
        live &= winners["score"] > 0
    elif r.SignFilter == SignFilter_NEG:
        live &= winners["score"] < 0
    # A Score that fails Results.passed leaves the heap untouched, and so does one that is not STRICTLY greater than the minimum
    # of a full heap (results.go:62-66) -- a minimum that only rises during a feed.  Neither is constructed: the Scores that do
    # reach Update arrive in group order, so the heap's history (and with it the order among exact ties) is the full feed's.
    cand = np.nonzero(live)[0]
    mag = np.abs(winners["score"][cand])
    fed, k, CH = 0, 0, 8192
    while k < len(cand):
        if len(r.scores) == r.TopN:
            if r.TopN <= 0:
                break                  # (TopN = 0: Update never pushes)
            floor = abs(r.scores[0].PercentScore)
            hit = np.nonzero(mag[k:k + CH] > floor)[0]
            if len(hit) == 0:
                k += CH
                continue
            k += int(hit[0])
        g = cand[k]
        r.Update(Score(labels_of(int(winners["series"][g]), int(g)), int(winners["lag"][g]), float(winners["score"][g])))
        fed += 1
        k += 1
    return fed


class Batch:
    def __init__(self, ref, comp, results, cc, engine=None, engines=None):
        # muse_batch.go:24-28 (length check over the registry)
        for uid, s in comp.registry.items():
            if ref.Length() != s.Length():
                raise MuseError(B.MUSE_ERR_LENGTH, "%s from comparison group series does not have the same "
                                "length as the reference" % uid)
        self.Concurrency = max(int(cc), 1)       # kept for API compatibility; the GPU is the fan-out
        self.Comparison = comp
        self.Results = results
        # engines: a list of Engine objects (one per device; a device may appear more than once) -- the Comparison group is
        # then sharded over them and every Run scores the shards at the same time, one host thread per device
        self._engines = list(engines) if engines and len(engines) > 1 else None
        self._engine = engine or (engines[0] if engines else None) or get_engine()
        self._ref = np.array(ref.Values(), dtype=np.float64)
        self._db = None
        self._shard_db = None
        self.n = next_pow2(float(ref.Length()))   # muse_batch.go:35
        # NewBatch computes the reference spectrum right away and returns
        # "Invalid input query" on sigma == 0 (muse_batch.go:38-41): probe it.
        probe = DeviceGroup(self._engine, ref.Length(), 0)
        try:
            DeviceBatch(self._engine, probe, self._ref).close()
        finally:
            probe.close()

    def _batch(self):
        dg = self.Comparison._device_group(self._engine)
        if self._db is None or self._db.dgroup is not dg:
            self._db = DeviceBatch(self._engine, dg, self._ref)
        return self._db

    def Run(self, groupByLabels):                 # muse_batch.go:99-130
        comp = self.Comparison
        labelValuesSet = comp.indexLabelValues(groupByLabels)
        if not labelValuesSet:
            return None
        series = comp._series_list()
        gid = comp._group_ids()
        r = self.Results
        G = len(labelValuesSet)
        if G <= EXACT_FEED_MAX_GROUPS:
            # the reference's own feed (muse_batch.go:124-128): ONE Score per label group, in group order, through Results.Update --
            # the heap's history, and with it the order Fetch returns exactly tied scores in and which of them survives at the
            # top-N boundary, is the reference's (for insertion-ordered groups), also when the Results already holds the Scores of
            # earlier Runs (results.go:55-72) and whether the Group sits on one device or is cut over several
            if self._engines:
                rec, state = self._group_winners_sharded(gid, G)
            else:
                rec, state = merge_group_winners(*[a[None, :] for a in self._batch().run_groups(gid, G, 0, abs_scores=True)])
            feed_group_winners(r, rec, state, lambda i, g: series[i].Labels())
            return None
        # very many label groups (Run(nil) over a million series): the device pre-selects the TopN candidates, 24 B x TopN cross
        # the host; among EXACTLY tied scores the order / the survivor at the boundary may then differ from a full feed
        if self._engines:
            idx, lag, score = self._run_sharded(gid, G)
        else:
            idx, lag, score, _ = self._batch().run(gid, G, r.MaxLag, r.TopN, r.Threshold, r.SignFilter, abs_scores=True)
        # feed Results in group order, as the ordered drain does (muse_batch.go:124-128)
        order = np.argsort(gid[idx], kind="stable")
        for k in order:
            r.Update(Score(series[int(idx[k])].Labels(), int(lag[k]), float(score[k])))
        return None

    def _shard_batches(self):
        shards = self.Comparison._device_shards(self._engines)
        if self._shard_db is None or len(self._shard_db) != len(shards) or \
                any(db.dgroup is not sh[1] for db, sh in zip(self._shard_db, shards)):
            self._shard_db = [DeviceBatch(e, dg, self._ref) for e, dg, _, _ in shards]
        return shards

    def _group_winners_sharded(self, gid, G):
        """every shard's winner per label group at the same time (one host thread per device), merged per group"""
        import threading
        shards = self._shard_batches()
        out, err = [None] * len(shards), [None] * len(shards)

        def work(k):
            _, _, lo, hi = shards[k]
            try:
                out[k] = self._shard_db[k].run_groups(gid[lo:hi], G, lo, abs_scores=True)
            except Exception as e:   # reported by the calling thread
                err[k] = e
        threads = [threading.Thread(target=work, args=(k,)) for k, sh in enumerate(shards) if sh[3] > sh[2]]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in err:
            if e is not None:
                raise e
        empty = (np.zeros(G, dtype=B.RECORD_DTYPE), np.zeros(G, dtype=np.uint8))
        empty[0]["series"] = -1
        parts = [o if o is not None else empty for o in out]
        return merge_group_winners(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))


    def _run_sharded(self, gid, G):
        """the sharded Run (muse.hpp Batch::run_sharded is the same code): top-N candidates per shard when every label group
        lives on one shard, per-group winners merged BEFORE filtering when groups straddle shards"""
        import threading
        shards = self._shard_batches()
        owner = np.full(G, -1, dtype=np.int64)
        straddle = False
        for k, (_, _, lo, hi) in enumerate(shards):
            g = np.unique(gid[lo:hi])
            straddle = straddle or bool(np.any((owner[g] >= 0) & (owner[g] != k)))
            owner[g] = k
        r = self.Results
        out = [None] * len(shards)
        err = [None] * len(shards)

        def work(k):
            _, _, lo, hi = shards[k]
            try:
                if straddle:
                    out[k] = self._shard_db[k].run_groups(gid[lo:hi], G, lo, abs_scores=True)
                else:
                    out[k] = self._shard_db[k].run_shard(gid[lo:hi], G, lo, r.MaxLag, r.TopN, r.Threshold, r.SignFilter, True)
            except Exception as e:   # reported by the calling thread
                err[k] = e
        threads = [threading.Thread(target=work, args=(k,)) for k, sh in enumerate(shards) if sh[3] > sh[2]]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in err:
            if e is not None:
                raise e
        if straddle:
            empty = (np.zeros(G, dtype=B.RECORD_DTYPE), np.zeros(G, dtype=np.uint8))
            empty[0]["series"] = -1
            parts = [o if o is not None else empty for o in out]
            idx, lag, score, _ = merge_group_records(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                                                     r.MaxLag, r.TopN, r.Threshold, r.SignFilter)
        else:
            recs = [o for o in out if o is not None and len(o)]
            allrec = np.concatenate(recs) if recs else np.zeros(0, dtype=B.RECORD_DTYPE)
            idx, lag, score, _ = merge_records(allrec, r.TopN)
        return idx, lag, score


def NewBatch(ref, comp, results, cc, engine=None, engines=None):
    return Batch(ref, comp, results, cc, engine, engines)


def RunMany(batches, groupByLabels):
    """Batch.Run for several batches that share one Comparison group and the same Results settings
    (README.md:10-13: many references against one set of series): the resident rows are read and
    transformed once for all references (muse_batch_run_many).  Not part of the reference's API;
    batches that do not qualify are run one after the other."""
    batches = list(batches)
    if not batches:
        return None
    b0 = batches[0]
    r0 = b0.Results
    same = all(b.Comparison is b0.Comparison and b._engine is b0._engine and
               (b.Results.MaxLag, b.Results.TopN, b.Results.Threshold, b.Results.SignFilter) ==
               (r0.MaxLag, r0.TopN, r0.Threshold, r0.SignFilter) for b in batches)
    if not same:
        for b in batches:
            b.Run(groupByLabels)
        return None
    comp = b0.Comparison
    labelValuesSet = comp.indexLabelValues(groupByLabels)
    if not labelValuesSet:
        return None
    series = comp._series_list()
    gid = comp._group_ids()
    res = run_many([b._batch() for b in batches], gid, len(labelValuesSet), r0.MaxLag, r0.TopN, r0.Threshold,
                   r0.SignFilter, abs_scores=True)
    for b, (idx, lag, score, _) in zip(batches, res):
        order = np.argsort(gid[idx], kind="stable")
        for k in order:
            b.Results.Update(Score(series[int(idx[k])].Labels(), int(lag[k]), float(score[k])))
    return None


# ----------------------------------------------------------------- muse.go
class Muse:
    def __init__(self, ref, results, engine=None):
        if ref.Length() < 1:                      # muse.go:24-26
            raise MuseError(B.MUSE_ERR_EMPTY, "Reference series length must be greater than zero")
        self.refN = ref.Length()
        self.n = next_pow2(float(self.refN))
        self.Results = results
        self._engine = engine or get_engine()
        self._ref = np.array(ref.Values(), dtype=np.float64)
        # the reference spectrum is computed once, as New does (muse.go:29-39: sigma(ref) == 0 is an error here);
        # every Run shares it through muse_batch_run_rows
        self._probe = DeviceGroup(self._engine, self.refN, 0)
        try:
            self._template = DeviceBatch(self._engine, self._probe, self._ref)
        except Exception:
            self._probe.close()
            raise

    def Run(self, compGraphs):                    # muse.go:46-92
        if len(compGraphs) == 0:
            return None
        for s in compGraphs:                      # muse.go:68-70
            if s.Length() != self.refN:
                raise MuseError(B.MUSE_ERR_LENGTH, "Encountered a comparison graph with differing length "
                                "than the reference, %r" % (s.Labels(),))
        rows = np.stack([s.y for s in compGraphs])
        win, state = self._template.run_rows(rows, abs_scores=False)      # one ABI call (muse_batch_run_rows)
        if state == 1 and win["series"] >= 0:
            # the group's Score through the unchanged Update, which applies passed() as the reference does (results.go:55-72)
            self.Results.Update(Score(compGraphs[int(win["series"])].Labels(), int(win["lag"]), float(win["score"])))
        return None


def New(ref, results, engine=None):
    return Muse(ref, results, engine)
