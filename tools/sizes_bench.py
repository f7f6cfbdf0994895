#!/usr/bin/env python3
"""Fused-kernel time per FFT length: automatic kernel selection against the radix-2 generic kernel.
usage: sizes_bench.py [bytes_per_group] [N ...]   (SIZES_AUTO_ONLY=1: skip the generic kernel; SIZES_VARIANT=k: test-hook kernel k instead of automatic selection)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
Ns = [int(a) for a in sys.argv[2:]] or [480, 512, 1000, 2048, 4096, 5000, 8192, 16384, 40000, 65536]
eng = pkg.get_engine(0)
for N in Ns:
    rows = max(2048, min(400_000, budget // (8 * N)))
    dg, ref = pkg.DeviceGroup.synthetic(eng, rows, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    out = []
    first = int(os.environ.get("SIZES_VARIANT", "0"))  # a test-hook kernel (muse_hip_test.h) instead of automatic selection
    for variant in ((first,) if os.environ.get("SIZES_AUTO_ONLY") else (first, 1)):
        if variant == 1 and db.n == 4096:
            continue
        eng.set_kernel(variant)
        db.score(); eng.synchronize()
        eng.kernel_timing(True)
        for _ in range(3):
            db.score()
        eng.synchronize()
        ms, cnt = eng.kernel_time()
        eng.kernel_timing(False)
        t = ms / cnt * 1e-3
        out.append("%s %8.3f ms %.3e series/s %5.0f GB/s (%4.1f%%)" % (
            "auto   " if variant == 0 else "generic" if variant == 1 else "kernel%2d" % variant, t * 1e3, rows / t, rows * (8 * N + 16) / t / 1e9,
            rows * (8 * N + 16) / t / 8e12 * 100))
    eng.set_kernel(0)
    print("N=%5d n=%5d rows=%6d: %s" % (N, db.n, rows, " | ".join(out)), flush=True)
    db.close(); dg.close()
eng.close()  # (diagnostic builds dump their phase stamps when the context goes)
