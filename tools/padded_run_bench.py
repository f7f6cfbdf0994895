#!/usr/bin/env python3
"""Run(nil) over 1 M zero-padded series (2048 < N < 4096): filter-and-refine against the all-fp64 path."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
M = 1_000_000
for N in (3000, 4000, 4096):
    dg, ref = pkg.DeviceGroup.synthetic(eng, M, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    for screening in (True, False):
        eng.set_screening(screening)
        db.run(None, 0, 15, 20, 0.0, 0, True)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = db.run(None, 0, 15, 20, 0.0, 0, True)
        dt = (time.perf_counter() - t0) / 5
        scr, pairs = db.last_run_info()
        print("N=%d  %-18s %.3f ms per Run  (%s)" % (N, "filter-and-refine" if scr else "all fp64", dt * 1e3,
                                                    "%d pairs re-evaluated" % pairs if scr else "-"), flush=True)
    eng.set_screening(True)
    db.close()
    del dg
