#!/usr/bin/env python3
"""All-scores pass per series length from 65 536 up (the lengths of xcorr_huge.hip against n = 65536, the longest length the
one-workgroup-per-series kernels take): HIP-event time of the pass, series-pairs/s, picoseconds per sample, fraction of the
8 TB/s roofline on 8 N + 16 bytes per series.
usage: huge_bench.py [group GB] [N ...]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
if os.environ.get("MUSE_AB_LIB"):  # another build of the library (an A/B on one box)
    import ctypes
    pkg.build.LIB = os.path.abspath(os.environ["MUSE_AB_LIB"])
    pkg.build.stale = lambda: False
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
Ns = [int(a) for a in sys.argv[2:]] or [65536, 131072, 262144, 524288, 1048576, 100000, 600000]
eng = pkg.get_engine(0)
if os.environ.get("MUSE_HUGE_BATCH_MB"):  # the batch's work buffer (measurement hook; default 128 MB)
    eng.huge_batch_mb(int(os.environ["MUSE_HUGE_BATCH_MB"]))
for N in Ns:
    M = max(8, int(GB * (1 << 30) / (8 * N)) // 2 * 2)
    dg, ref = pkg.DeviceGroup.synthetic(eng, M, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    for _ in range(3):
        db.score()
    eng.synchronize()
    eng.kernel_time()
    eng.kernel_timing(True)
    reps = 6
    for _ in range(reps):
        db.score()
    eng.synchronize()
    eng.kernel_timing(False)
    ms, cnt = eng.kernel_time()
    t = ms / reps * 1e-3
    print("N=%8d n=%8d M=%7d: pass %9.3f ms  %.3e series/s  %6.2f ps/sample  %.3f of 8 TB/s  (kernel %s)" % (
        N, db.n, M, t * 1e3, M / t, t / (M * float(N)) * 1e12, M * (8.0 * N + 16) / t / 8e12, eng.kernel_name(db)), flush=True)
    db.close()
    dg.close()
