#!/usr/bin/env python3
"""xcorr_small.hip (variant 12) against the round-1 Stockham kernels (11) and the oracle on small groups, per length."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
from oracle import oracle_py
eng = pkg.get_engine(0)
rng = np.random.default_rng(3)
for N in (512, 480, 300, 1024, 1000, 2048, 1500):
    for M in (2, 5, 16, 37):
        ref = rng.standard_normal(N)
        rows = rng.standard_normal((M, N)) + rng.uniform(-2, 2, size=(M, 1)) * np.roll(ref, 3)
        db = pkg.DeviceBatch(eng, pkg.DeviceGroup.from_rows(eng, rows), ref)
        olag, omv, gap = oracle_py.batch_scores(ref, rows)
        out = []
        for v in (11, 12):
            eng.set_kernel(v)
            lag, mv = db.scores()
            bad = np.nonzero((np.abs(mv - omv) > 1e-9 * np.maximum(np.abs(omv), 1e-12)) | (lag != olag))[0]
            out.append("v%d: %d bad %s" % (v, len(bad), bad[:8].tolist()))
            if v == 12 and len(bad):
                i = bad[0]
                out.append("row %d: got (%d, %.6g) want (%d, %.6g)" % (i, lag[i], mv[i], olag[i], omv[i]))
        eng.set_kernel(0)
        print("N=%d M=%d  %s" % (N, M, " | ".join(out)), flush=True)
        db.close()
