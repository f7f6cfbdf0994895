#!/usr/bin/env python3
"""xcorr_real.hip (test hook 14: n = 32768 as one real series per workgroup) against the CPU checker and the long-series kernel"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
from oracle import oracle_py
eng = pkg.get_engine(0)
rng = np.random.default_rng(5)
for N in [int(a) for a in sys.argv[1:]] or (32768, 20000, 16385, 32767, 24577, 65536, 40000, 32769, 65535):
    M = 13
    t = np.arange(N)
    ref = 1.5 * (np.abs(t - N // 2) <= 5) + 0.1 * rng.standard_normal(N)
    rows = rng.uniform(-2, 2, (M, 1)) * (np.abs(t[None, :] - N // 2 - rng.integers(-300, 301, (M, 1))) <= 6) + 0.3 * rng.standard_normal((M, N)) + rng.normal(size=(M, 1)) * 10
    rows[3] = 2.5
    rows[5, N // 3] = np.nan
    rows[7] = np.roll(ref, 77) * 3 + 1
    rows[8] *= 1e-7
    dg = pkg.DeviceGroup.from_rows(eng, rows)
    db = pkg.DeviceBatch(eng, dg, ref)
    olag, omv, gap = oracle_py.batch_scores(ref, rows, nthreads=8)
    for variant in (0, 14):
        eng.set_kernel(variant)
        lag, mv = db.scores()
        ok = ~np.isnan(omv)
        err = np.max(np.abs(mv[ok] - omv[ok]) / np.maximum(np.abs(omv[ok]), 1e-300) * (np.abs(omv[ok]) > 0) + np.abs(mv[ok]) * (omv[ok] == 0))
        bad = int(np.sum((lag != olag) & (gap >= 1e-12) & ok))
        print("N=%5d variant %2d: worst rel err %.2e, lag mismatches %d, NaN rows agree %s" % (N, variant, err, bad, np.array_equal(np.isnan(mv), np.isnan(omv))), flush=True)
        if bad or err > 1e-6:
            print("   lag", lag.tolist(), "\n   exp", olag.tolist(), "\n   mv ", mv.tolist(), "\n   exp", omv.tolist())
    eng.set_kernel(0)
    db.close(); dg.close()
