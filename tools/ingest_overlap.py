#!/usr/bin/env python3
"""host -> HBM ingestion beside a running score pass.  A resident group of 400 000 x 4096 rows is scored
(about 4 ms) while a 1 GB slab is appended; compared with the two done one after the other."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
N, M0, MS = 4096, 400_000, 32_768
dg, ref = pkg.DeviceGroup.synthetic(eng, M0, N)
slab = np.random.default_rng(0).standard_normal((MS, N))
db = pkg.DeviceBatch(eng, dg, ref)
import ctypes
B = pkg.binding
# capacity for several slabs up front (no reallocation inside the timed region)
big = pkg.DeviceGroup(eng, N, M0 + 8 * MS)
B.check(B.load().muse_group_fill_synthetic(big._h, 0, M0, 0, ctypes.c_uint64(1), ctypes.c_uint32(0), None))
db = pkg.DeviceBatch(eng, big, ref)
db.score(); eng.synchronize()
def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        eng.synchronize(); t0 = time.perf_counter(); f(); eng.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
t_score = timed(lambda: db.score())
t_app = timed(lambda: big.append(slab))
def both():
    db.score()
    big.append(slab)
t_both = timed(both)
print("score alone %.2f ms | append of %.2f GB alone %.2f ms (%.1f GB/s) | score + append overlapped %.2f ms (sum %.2f ms)"
      % (t_score * 1e3, slab.nbytes / 1e9, t_app * 1e3, slab.nbytes / t_app / 1e9, t_both * 1e3, (t_score + t_app) * 1e3))
