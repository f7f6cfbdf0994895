#!/usr/bin/env python3
"""Debug aid: screening estimates against fp64 scores for one series length (python tools/screen_len_debug.py N [M])."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
N = int(sys.argv[1]); M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = pkg.get_engine(0)
rng = np.random.default_rng(3)
ref = rng.standard_normal(N)
rows = rng.standard_normal((M, N))
for i in range(0, M, 2):
    rows[i] += 2.0 * np.roll(ref, int(rng.integers(-N // 2, N // 2)))
dg = pkg.DeviceGroup.from_rows(eng, rows)
db = pkg.DeviceBatch(eng, dg, ref)
lag, mv = db.scores()
est, flags, E = db.screen_estimates(15)
print("n", db.n, "E", E)
refined = (flags >> 31) & 1
print("refined", int(refined.sum()), "flags hist", {int(k): int(v) for k, v in zip(*np.unique(flags & 63, return_counts=True))})
chk = refined == 0
err = np.abs(np.abs(est[chk]) - np.abs(mv[chk]))
print("max err", err.max() if chk.any() else None, "ratio", (err.max() / E) if chk.any() else None)
print("est", est[:8]); print("mv ", mv[:8])
