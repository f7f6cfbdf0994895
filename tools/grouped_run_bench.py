#!/usr/bin/env python3
"""Batch.Run with label groups (config 5 style: G = M/50 groups) against Run(nil) on the same resident matrix."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
eng = pkg.get_engine(0)
M, N = 1_000_000, 4096
dg, ref = pkg.DeviceGroup.synthetic(eng, M, N)
db = pkg.DeviceBatch(eng, dg, ref)
gid_blocks = (np.arange(M) // 50).astype(np.int32)
gid_mixed = (np.arange(M) % (M // 50)).astype(np.int32)
for screening in (True, False):
    eng.set_screening(screening)
    for name, gid, G in (("Run(nil)", None, 0), ("groups of 50, contiguous", gid_blocks, M // 50), ("groups of 50, interleaved", gid_mixed, M // 50)):
        db.run(gid, G, 15, 20, 0.0, 0, True)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = db.run(gid, G, 15, 20, 0.0, 0, True)
        dt = (time.perf_counter() - t0) / 5
        scr, pairs = db.last_run_info()
        print("%-28s %.3f ms per Run (%d results, top %.4f)  %s" % (
            name, dt * 1e3, len(out[0]), out[2][0] if len(out[2]) else float("nan"),
            "filter-and-refine, %d pairs re-evaluated" % pairs if scr else "all fp64"), flush=True)
eng.set_screening(True)
