#!/usr/bin/env python3
"""Long series (n = 32768, 65536): the default four-step kernel (xcorr_long.hip, one slice per workgroup) against variant 14
(xcorr_long_team.hip: tasks of one pair on one XCD, slices in its L2) over geometries (workgroups per CU, slices per XCD).
usage: long_team_bench.py [bytes_per_group] [N ...]     LONG_TEAM_GEOMS="1x3,2x4,..." overrides the geometries"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 32
Ns = [int(a) for a in sys.argv[2:]] or [32768, 65536, 20000, 40000]
geoms = [tuple(int(x) for x in g.split("x")) for g in os.environ.get("LONG_TEAM_GEOMS", "2x4x1,4x8x1,4x5x2,4x8x2,4x7x3,4x8x3").split(",")]
geoms = [g if len(g) == 3 else g + (1,) for g in geoms]
eng = pkg.get_engine(0)


def timed(db, reps=3):
    db.score(); eng.synchronize()
    eng.kernel_time()
    eng.kernel_timing(True)
    for _ in range(reps):
        db.score()
    eng.synchronize()
    ms, cnt = eng.kernel_time()
    eng.kernel_timing(False)
    return ms / cnt * 1e-3


for N in Ns:
    rows = max(2048, min(400_000, budget // (8 * N)))
    dg, ref = pkg.DeviceGroup.synthetic(eng, rows, N)
    db = pkg.DeviceBatch(eng, dg, ref)
    eng.set_kernel(0)
    t0 = timed(db)
    lag0, mv0 = db.scores()
    print("N=%5d n=%5d rows=%6d: default  %8.3f ms %.3e series/s (%4.1f%% of 8 TB/s)" % (
        N, db.n, rows, t0 * 1e3, rows / t0, rows * (8 * N + 16) / t0 / 8e12 * 100), flush=True)
    for wgs, slots, dist in geoms:
        eng.set_kernel(14)
        eng.long_team_config(wgs, slots, dist)
        try:
            t1 = timed(db)
            lag1, mv1 = db.scores()
        except Exception as e:
            print("   team %dx%dx%d: FAILED %s" % (wgs, slots, dist, e), flush=True)
            eng.set_kernel(0)
            continue
        ok = np.isfinite(mv0)
        bad_lag = int(np.sum(lag0 != lag1))
        rel = float(np.max(np.abs(mv1[ok] - mv0[ok]) / np.maximum(np.abs(mv0[ok]), 1e-300))) if ok.any() else 0.0
        nan_same = bool(np.array_equal(np.isnan(mv0), np.isnan(mv1)))
        print("   team %d wgs/CU x %d slices/XCD, distance %d: %8.3f ms %.3e series/s (%4.1f%%)  x%.2f   lags differ %d, max rel %.1e, nan pattern %s" % (
            wgs, slots, dist, t1 * 1e3, rows / t1, rows * (8 * N + 16) / t1 / 8e12 * 100, t0 / t1, bad_lag, rel, nan_same), flush=True)
    eng.set_kernel(0)
    db.close(); dg.close()
