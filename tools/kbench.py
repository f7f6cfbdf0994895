#!/usr/bin/env python3
"""A/B timing of the fused-kernel variants in ONE process on the same resident
matrix (interleaved rounds), with a cross-check of their outputs.
usage: python tools/kbench.py [rows] [rounds] [variants...]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variants = [int(v) for v in sys.argv[3:]] or [7, 10]
eng = pkg.get_engine(0)
dg, ref = pkg.DeviceGroup.synthetic(eng, rows, 4096)
db = pkg.DeviceBatch(eng, dg, ref)
base = None
times = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        eng.set_kernel(v)
        eng.kernel_timing(True)
        db.score()
        eng.synchronize()
        ms, cnt = eng.kernel_time()
        eng.kernel_timing(False)
        if r > 0:
            times[v].append(ms / cnt)
        if r == 0:
            lag, mv = db.scores()
            if base is None:
                base = (lag, mv)
            else:
                same_lag = int((lag != base[0]).sum())
                rel = np.nanmax(np.abs(mv - base[1]) / np.maximum(np.abs(base[1]), 1e-300))
                print("variant %d vs %d: lag diffs %d, max rel score diff %.3e" % (v, variants[0], same_lag, rel))
                with np.errstate(all="ignore"):
                    bad = np.nonzero((lag != base[0]) | (np.abs(mv - base[1]) > 1e-9 * np.maximum(np.abs(base[1]), 1e-300))
                                     | (np.isnan(mv) != np.isnan(base[1])))[0]
                for i in bad[:8]:
                    print("   row %d: lag %d vs %d, mv %.17g vs %.17g" % (i, lag[i], base[0][i], mv[i], base[1][i]))
for v in variants:
    t = np.array(times[v])
    print("variant %d: median %.3f ms  min %.3f ms  -> %.3e series/s, %.1f%% of 8 TB/s" % (
        v, np.median(t), t.min(), rows / (np.median(t) * 1e-3), rows * 32784 / (np.median(t) * 1e-3) / 8e12 * 100))
