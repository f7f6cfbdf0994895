#!/usr/bin/env python3
"""Fused-kernel time per FFT length (generic LDS kernel for n != 4096).  usage: sizes_bench.py [rows]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
eng = pkg.get_engine(0)
for N in (480, 512, 1000, 2048, 4096, 5000, 8192):
    dg, ref = pkg.DeviceGroup.synthetic(eng, rows, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    db.score(); eng.synchronize()
    eng.kernel_timing(True)
    for _ in range(3):
        db.score()
    eng.synchronize()
    ms, cnt = eng.kernel_time()
    eng.kernel_timing(False)
    t = ms / cnt * 1e-3
    print("N=%5d n=%5d: %8.3f ms per %d series -> %.3e series/s, %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        N, db.n, t * 1e3, rows, rows / t, rows * (8 * N + 16) / t / 1e9, rows * (8 * N + 16) / t / 8e12 * 100))
    db.close(); dg.close()
