#!/usr/bin/env python3
"""Run(nil) time against the row count with the filter-and-refine path on / off (where does screening start to pay?)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
for M in (16384, 32768, 65536, 131072, 262144, 524288):
    dg, ref = pkg.DeviceGroup.synthetic(eng, M, 4096)
    db = pkg.DeviceBatch(eng, dg, ref)
    res = {}
    for screening in (True, False):
        eng.set_screening(screening)
        for _ in range(3):
            db.run(None, 0, 15, 20, 0.0, 0, True)
        eng.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            db.run(None, 0, 15, 20, 0.0, 0, True)
        res[screening] = (time.perf_counter() - t0) / reps * 1e3
        scr = db.last_run_info()[0]
        assert scr == screening, (M, scr)
    print("M=%7d  filter-and-refine %.3f ms   all fp64 %.3f ms   ratio %.2f" % (M, res[True], res[False], res[False] / res[True]), flush=True)
    eng.set_screening(True)
    db.close()
    del dg
