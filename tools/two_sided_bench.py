#!/usr/bin/env python3
"""Batched two-sided xCorr (muse_xcorr_groups) over resident groups: kernel time per call.
usage: two_sided_bench.py [pairs] [N ...]   (n = N)"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
if os.environ.get("MUSE_AB_LIB"):  # another build of the library (an A/B on one box: boxes differ by several per cent)
    pkg.build.LIB = os.path.abspath(os.environ["MUSE_AB_LIB"])
    pkg.build.stale = lambda: False
P = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
Ns = [int(a) for a in sys.argv[2:]] or [4096]
eng = pkg.get_engine(0)
eng.set_kernel(int(os.environ.get("MUSE_TEST_KERNEL", "0")))  # a test hook (muse_hip_test.h) instead of automatic selection
for N in Ns:
    rows = max(1024, min(P, (6 << 30) // (8 * N)))
    gx, _ = pkg.DeviceGroup.synthetic(eng, rows, N, seed=0x78636F72)
    gy, _ = pkg.DeviceGroup.synthetic(eng, rows, N, seed=0x6D757365)
    for normalize in (True, False):
        for _ in range(8):  # (past the clock ramp: the first launches of a process run at a boost clock the part does not hold)
            pkg.xcorr_groups(gx, gy, N, normalize)
        eng.synchronize()
        eng.kernel_time()
        eng.kernel_timing(True)
        for _ in range(8):
            pkg.xcorr_groups(gx, gy, N, normalize)
        eng.synchronize()
        eng.kernel_timing(False)
        ms, cnt = eng.kernel_time()
        k = ms / max(cnt, 1)
        print("N=%5d pairs=%6d normalize=%d: kernel %8.3f ms  %.3e pairs/s  (%.1f %% of 8 TB/s on 16 N + 16 B per pair)" % (
            N, rows, normalize, k, rows / (k * 1e-3), rows * (16.0 * N + 16) / (k * 1e-3) / 8e12 * 100), flush=True)
    gx.close()
    gy.close()
