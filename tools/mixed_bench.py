#!/usr/bin/env python3
"""Mixed-unit group (every second series scaled by 1e30): first pass (default kernel + full hand-off)
against later passes (automatic selection goes to the rescaling kernel directly)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
M, N = 200_000, 4096
dg, ref = pkg.DeviceGroup.synthetic(eng, M, N)
rows = dg.read(0, 2048)
big = np.tile(rows, (8, 1))
big[1::2] *= 1e30
dg2 = pkg.DeviceGroup.from_rows(eng, big)
db = pkg.DeviceBatch(eng, dg2, ref)
for i in range(4):
    eng.synchronize(); t0 = time.perf_counter(); db.score(); eng.synchronize()
    print("pass %d: %.3f ms for %d series" % (i, (time.perf_counter() - t0) * 1e3, len(big)), flush=True)
