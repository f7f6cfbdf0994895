#!/usr/bin/env python3
"""Run(nil) per FFT length (2 GB groups): filter-and-refine against the all-fp64 path."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
for N in [int(a) for a in sys.argv[1:]] or (512, 700, 1024, 2048, 4096, 8192):
    M = int(float(os.environ.get("GROUP_GB", "2")) * 1e9 // (8 * N))
    dg, ref = pkg.DeviceGroup.synthetic(eng, M, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    res = {}
    for screening in (True, False):
        eng.set_screening(screening)
        for _ in range(2):
            db.run(None, 0, 15, 20, 0.0, 0, True)
        eng.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            db.run(None, 0, 15, 20, 0.0, 0, True)
        res[screening] = ((time.perf_counter() - t0) / reps * 1e3, db.last_run_info())
    byts = M * (8 * N + 16)
    print("N=%5d M=%7d  filter-and-refine %.3f ms (%.1f%% of 8 TB/s; screened=%s, %d pairs re-evaluated)   all fp64 %.3f ms (%.1f%%)" % (
        N, M, res[True][0], byts / (res[True][0] * 1e-3) / 8e12 * 100, res[True][1][0], res[True][1][1],
        res[False][0], byts / (res[False][0] * 1e-3) / 8e12 * 100), flush=True)
    eng.set_screening(True)
    db.close()
    del dg
