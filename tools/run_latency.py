#!/usr/bin/env python3
"""Wall time of one Batch.Run (fused pass + group max + top-N + copy back) per group size: BASELINE config 2 (10 000 x 4096)
and neighbours.  usage: run_latency.py [N] [M ...]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ms = [int(a) for a in sys.argv[2:]] or [100, 1000, 10_000, 100_000, 1_000_000]
eng = pkg.get_engine(0)
for M in Ms:
    dg, ref = pkg.DeviceGroup.synthetic(eng, M, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    db.run(None, 0, 15, 20, 0.0, 0, True)
    reps = 50 if M <= 100_000 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        db.run(None, 0, 15, 20, 0.0, 0, True)
    dt = (time.perf_counter() - t0) / reps
    eng.kernel_time()
    eng.kernel_timing(True)
    for _ in range(5):
        db.score()
    eng.synchronize()
    eng.kernel_timing(False)
    ms, cnt = eng.kernel_time()
    print("M=%8d N=%d: Run %.3f ms (%.3e series/s); fused kernel alone %.3f ms" % (M, N, dt * 1e3, M / dt, ms / max(cnt, 1)), flush=True)
    db.close(); dg.close()
