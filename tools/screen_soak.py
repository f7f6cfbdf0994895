#!/usr/bin/env python3
"""Soak check at full size: filter-and-refine Run against the all-fp64 Run on the 1 M synthetic rows under random filters,
with and without label groups (scores must agree; every returned row must carry its exact lag and score)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 30
eng = pkg.get_engine(0)
dg, ref = pkg.DeviceGroup.synthetic(eng, M, 4096, seed=2027)
db = pkg.DeviceBatch(eng, dg, ref)
lag, mv = db.scores()
rng = np.random.default_rng(1)
bad = 0
for trial in range(trials):
    if trial % 3 == 2:
        G = int(rng.choice([100, 20000, 300000]))
        gid = rng.integers(0, G, size=M).astype(np.int32)
    else:
        gid, G = None, 0
    max_lag = int(rng.choice([0, 5, 15, 100, 2048, 4096]))
    top_n = int(rng.choice([1, 5, 20, 100, 256]))
    thr = float(rng.choice([0.0, 0.1, 0.4, 0.9]))
    sign = int(rng.choice([0, 1, -1]))
    absf = bool(rng.random() < 0.5)
    eng.set_screening(False)
    exp = db.run(gid, G, max_lag, top_n, thr, sign, absf)
    eng.set_screening(True)
    got = db.run(gid, G, max_lag, top_n, thr, sign, absf)
    scr, pairs = db.last_run_info()
    ok = len(got[0]) == len(exp[0]) and np.allclose(got[2], exp[2], rtol=1e-12, atol=0)
    rows = got[0]
    ok = ok and np.array_equal(lag[rows], got[1])
    exact = np.clip(np.abs(mv[rows]), None, 1.0) if absf else np.clip(mv[rows], -1.0, 1.0)
    ok = ok and np.allclose(got[2], exact, rtol=1e-12, atol=0)
    ok = ok and (len(rows) == 0 or np.all(np.abs(lag[rows]) <= max_lag))
    bad += 0 if ok else 1
    print("trial %2d G=%6d MaxLag=%4d TopN=%3d thr=%.1f sign=%2d abs=%d: %s  (%s, %d pairs re-evaluated, %d records)" % (
        trial, G, max_lag, top_n, thr, sign, absf, "ok" if ok else "MISMATCH", "screened" if scr else "fp64", pairs, len(rows)), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
