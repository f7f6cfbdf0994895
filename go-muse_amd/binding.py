"""ctypes binding of libmuse_hip.so -- every symbol include/muse_hip.h and include/muse_hip_test.h declare.

Loading the library does not need a GPU (the symbol-export test runs on CPU);
every compute call does, and fails loudly (MuseError) when the device or the
built library is missing.  There is no fallback path.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f64 = ctypes.c_double
_vp = ctypes.c_void_p
_dp = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)

MUSE_OK = 0
MUSE_ERR_INVALID = -1
MUSE_ERR_LENGTH = -2
MUSE_ERR_ZERO_STD = -3
MUSE_ERR_NO_DEVICE = -4
MUSE_ERR_HIP = -5
MUSE_ERR_UNSUPPORTED = -6
MUSE_ERR_NOMEM = -7
MUSE_ERR_EMPTY = -8


class MuseRecord(ctypes.Structure):
    _fields_ = [("series", _i64), ("score", _f64), ("lag", _i32), ("group", _i32)]


RECORD_DTYPE = np.dtype([("series", "<i8"), ("score", "<f8"), ("lag", "<i4"), ("group", "<i4")])
_recp = ctypes.POINTER(MuseRecord)

# name -> (restype, argtypes): the complete ABI of include/muse_hip.h (+ the test hooks of include/muse_hip_test.h)
SIGNATURES = {
    "muse_abi_version": (ctypes.c_int, []),
    "muse_last_error": (ctypes.c_char_p, []),
    "muse_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "muse_ctx_create": (ctypes.c_int, [_i32, ctypes.POINTER(_vp)]),
    "muse_device_count": (ctypes.c_int, [_i32p]),
    "muse_ctx_destroy": (ctypes.c_int, [_vp]),
    "muse_ctx_synchronize": (ctypes.c_int, [_vp]),
    "muse_ctx_device_info": (ctypes.c_int, [_vp, ctypes.c_char_p, _i32, _i32p, _i64p]),
    "muse_ctx_set_kernel": (ctypes.c_int, [_vp, _i32]),
    "muse_ctx_set_screening": (ctypes.c_int, [_vp, _i32]),
    "muse_ctx_kernel_timing": (ctypes.c_int, [_vp, _i32]),
    "muse_ctx_kernel_time": (ctypes.c_int, [_vp, _dp, _i64p]),
    "muse_ctx_redo_time": (ctypes.c_int, [_vp, _dp, _i64p]),
    "muse_ctx_device_pci_bus_id": (ctypes.c_int, [_vp, ctypes.c_char_p, _i32]),
    "muse_group_create": (ctypes.c_int, [_vp, _i64, _i32, ctypes.POINTER(_vp)]),
    "muse_group_create_f32": (ctypes.c_int, [_vp, _i64, _i32, ctypes.POINTER(_vp)]),
    "muse_group_append": (ctypes.c_int, [_vp, _dp, _i64, _i64]),
    "muse_group_upload": (ctypes.c_int, [_vp, _dp, _i64, _i32, _i64, ctypes.POINTER(_vp)]),
    "muse_group_fill_synthetic": (ctypes.c_int, [_vp, _i64, _i64, _i64, ctypes.c_uint64, ctypes.c_uint32, _dp]),
    "muse_group_shape": (ctypes.c_int, [_vp, _i64p, _i32p]),
    "muse_group_read": (ctypes.c_int, [_vp, _i64, _i64, _dp]),
    "muse_group_free": (ctypes.c_int, [_vp]),
    "muse_batch_create": (ctypes.c_int, [_vp, _vp, _dp, _i32, ctypes.POINTER(_vp)]),
    "muse_batch_create_like": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp)]),
    "muse_batch_run_rows": (ctypes.c_int, [_vp, _dp, _i64, _i64, _i32, _recp, ctypes.POINTER(ctypes.c_uint8)]),
    "muse_batch_run_row_ptrs": (ctypes.c_int, [_vp, ctypes.POINTER(_dp), _i64, _i32, _recp, ctypes.POINTER(ctypes.c_uint8)]),
    "muse_group_stage": (ctypes.c_int, [_vp, _i64, ctypes.POINTER(_dp), _i64p]),
    "muse_group_commit": (ctypes.c_int, [_vp, _i64, _i64]),
    "muse_ctx_trim": (ctypes.c_int, [_vp]),
    "muse_batch_fft_len": (ctypes.c_int, [_vp, _i32p]),
    "muse_batch_spectrum": (ctypes.c_int, [_vp, _dp]),
    "muse_batch_score": (ctypes.c_int, [_vp]),
    "muse_batch_scores": (ctypes.c_int, [_vp, _i32p, _dp]),
    "muse_batch_run": (ctypes.c_int, [_vp, _i32p, _i32, _i32, _i32, _f64, _i32, _i32,
                                      _i64p, _i32p, _dp, _i32p, _dp]),
    "muse_batch_run_shard": (ctypes.c_int, [_vp, _i32p, _i32, _i64, _i32, _i32, _f64, _i32, _i32,
                                            _recp, _i32p]),
    "muse_batch_run_groups": (ctypes.c_int, [_vp, _i32p, _i32, _i64, _i32, _recp, ctypes.POINTER(ctypes.c_uint8)]),
    "muse_merge_group_records": (ctypes.c_int, [_recp, ctypes.POINTER(ctypes.c_uint8), _i32, _i32, _i32, _i32, _f64, _i32,
                                                _i64p, _i32p, _dp, _i32p, _dp]),
    "muse_merge_group_winners": (ctypes.c_int, [_recp, ctypes.POINTER(ctypes.c_uint8), _i32, _i32, _recp, ctypes.POINTER(ctypes.c_uint8)]),
    "muse_merge_records": (ctypes.c_int, [_recp, _i64, _i32, _i64p, _i32p, _dp, _i32p, _dp]),
    "muse_batch_score_many": (ctypes.c_int, [ctypes.POINTER(_vp), _i32]),
    "muse_batch_read_scores": (ctypes.c_int, [_vp, _i32p, _dp]),
    "muse_batch_last_run_info": (ctypes.c_int, [_vp, _i32p, _i64p]),
    "muse_batch_last_run_path": (ctypes.c_int, [_vp, _i32p]),
    "muse_batch_kernel_name": (ctypes.c_int, [_vp, ctypes.c_char_p, _i32]),
    "muse_test_set_screen_bound_scale": (ctypes.c_int, [_vp, _f64]),
    "muse_test_screen_bound": (ctypes.c_int, [_i32, _f64, _dp]),
    "muse_test_wave_argmax": (ctypes.c_int, [_vp, _dp, _dp, _dp]),
    "muse_test_rows_always_copy": (ctypes.c_int, [_vp, _i32]),
    "muse_test_pool_stats": (ctypes.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "muse_test_xcorr_repeat": (ctypes.c_int, [_vp, _i32]),
    "muse_test_huge_batch_mb": (ctypes.c_int, [_vp, _i32]),
    "muse_test_clock_probe_start": (ctypes.c_int, [_vp, _f64, _f64]),
    "muse_test_clock_probe_stop": (ctypes.c_int, [_vp]),
    "muse_test_clock_probe_read": (ctypes.c_int, [_vp, _dp, _i32, _i32p]),
    "muse_batch_screen_estimates": (ctypes.c_int, [_vp, _i32, _dp, ctypes.POINTER(ctypes.c_uint32), _dp]),
    "muse_batch_run_many": (ctypes.c_int, [ctypes.POINTER(_vp), _i32, _i32p, _i32, _i32, _i32, _f64, _i32, _i32,
                                           _i64p, _i32p, _dp, _i32p, _dp]),
    "muse_batch_free": (ctypes.c_int, [_vp]),
    "muse_xcorr_with_x": (ctypes.c_int, [_vp, _dp, _dp, _i32, _i32, _dp, _i32p, _dp, _i32p]),
    "muse_xcorr": (ctypes.c_int, [_vp, _dp, _i32, _dp, _i32, _i32, _i32, _dp, _i32p, _dp, _i32p]),
    "muse_xcorr_groups": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32p, _dp, _i32p, _dp]),
    "muse_xcorr_batch": (ctypes.c_int, [_vp, _dp, _dp, _i64, _i32, _i32, _i32, _i32, _i32p, _dp, _i32p, _dp]),
    "muse_next_pow2": (_i64, [_f64]),
}


class MuseError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("muse_hip status %d: %s" % (status, message))
        self.status = status
        self.message = message


_lib = None


def lib_path():
    return _build.LIB


def load():
    """dlopen the in-tree library (built by __graft_entry__.build()).  Raises if
    it is missing -- the product never substitutes another implementation."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise MuseError(MUSE_ERR_NO_DEVICE,
                            "%s is not built; run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        L = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != MUSE_OK:
        msg = load().muse_last_error()
        raise MuseError(status, msg.decode("utf-8", "replace") if msg else "")


def as_f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def dptr(a):
    return a.ctypes.data_as(_dp)


def i32ptr(a):
    return a.ctypes.data_as(_i32p)


def i64ptr(a):
    return a.ctypes.data_as(_i64p)


def recptr(a):
    return a.ctypes.data_as(_recp)
