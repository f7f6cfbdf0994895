#!/usr/bin/env python3
"""Shader clock over time around a burst of all-scores passes (muse_test_clock_probe_*): prints the median clock of every
10 ms of a probe that runs before, during and after `steps` back-to-back passes over a resident 1 M x 4096 group.
usage: clock_trace.py [steps] [rows]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
eng = pkg.get_engine(0)
dg, ref = pkg.DeviceGroup.synthetic(eng, rows, 4096)
db = pkg.DeviceBatch(eng, dg, ref)
db.score()
eng.synchronize()
total = 100 + steps * 10.5 + 150
eng.clock_probe_start(1.0, total)
time.sleep(0.1)
eng.kernel_timing(True)
t0 = time.perf_counter()
for _ in range(steps):
    db.score()
eng.synchronize()
dt = time.perf_counter() - t0
eng.kernel_timing(False)
ms, cnt = eng.kernel_time()
mhz = eng.clock_probe_read()
print("%d passes in %.1f ms (kernel avg %.3f ms); probe windows: %d" % (steps, dt * 1e3, ms / max(cnt, 1), len(mhz)))
for i in range(0, len(mhz), 10):
    w = mhz[i:i + 10]
    print("  t = %4d ms: median %6.0f MHz  min %6.0f  max %6.0f" % (i, np.median(w), w.min(), w.max()))
