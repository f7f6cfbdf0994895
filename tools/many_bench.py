#!/usr/bin/env python3
"""R references against one resident group: one pass (muse_batch_score_many) against R single-reference passes.
usage: many_bench.py [bytes_per_group] [R] [N ...]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
if os.environ.get("MUSE_AB_LIB"):  # another build of the library (an A/B on one box: boxes differ by several per cent)
    import ctypes
    pkg.build.LIB = os.path.abspath(os.environ["MUSE_AB_LIB"])
    pkg.build.stale = lambda: False
    _L = ctypes.CDLL(pkg.build.LIB)
    pkg.binding.SIGNATURES = {k: v for k, v in pkg.binding.SIGNATURES.items() if hasattr(_L, k)}
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Ns = [int(a) for a in sys.argv[3:]] or [512, 1024, 2048, 4096, 8192, 16384]
eng = pkg.get_engine(0)
for N in Ns:
    rows = max(2048, min(400_000, budget // (8 * N)))
    dg, ref = pkg.DeviceGroup.synthetic(eng, rows, N)
    refs = [ref] + [dg.read(101 * r + 1, 1)[0] for r in range(1, R)]
    bs = [pkg.DeviceBatch(eng, dg, x) for x in refs]

    def timed(fn, reps=3):
        fn(); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.synchronize()
        return (time.perf_counter() - t0) / reps

    t_many = timed(lambda: pkg.score_many(bs))
    t_each = timed(lambda: [b.score() for b in bs])
    print("N=%5d n=%5d rows=%6d R=%d: one pass %8.3f ms (%.3e series-refs/s) | R passes %8.3f ms (%.3e) | x%.2f" % (
        N, bs[0].n, rows, R, t_many * 1e3, rows * R / t_many, t_each * 1e3, rows * R / t_each, t_each / t_many), flush=True)
    for b in bs:
        b.close()
    dg.close()
