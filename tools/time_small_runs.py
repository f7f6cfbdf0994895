#!/usr/bin/env python3
"""Latency of small Runs through the C ABI (muse_batch_run): Run(nil), Run over 100 label groups and Run with the identity label
map (every series its own group, through the general reduction when M > 2048) on 100 000 / 10 000 x 4096, 5 000 x 480 and 6 x 8."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
for M, N in ((100000, 4096), (10000, 4096), (5000, 480), (6, 8)):
    dg, ref = pkg.DeviceGroup.synthetic(eng, M, N, copies=False, constants=False) if M > 100 else (None, None)
    if dg is None:
        rng = np.random.default_rng(1); rows = rng.standard_normal((M, N)); ref = rng.standard_normal(N)
        dg = pkg.DeviceGroup.from_rows(eng, rows)
    db = pkg.DeviceBatch(eng, dg, ref)
    G = min(100, M)
    for tag, args in (("Run(nil)", (None, 0)), ("Run(graph) G=%d" % G, ((np.arange(M) % G).astype(np.int32), G)), ("identity map", (np.arange(M, dtype=np.int32), M))):
        for _ in range(20): db.run(*args)
        t0 = time.perf_counter()
        for _ in range(300): db.run(*args)
        dt = (time.perf_counter() - t0) / 300
        print("%6d x %4d %-18s %.1f us per Run, %.3e series-pairs/s" % (M, N, tag, dt * 1e6, M / dt), flush=True)
    db.close(); dg.close()
