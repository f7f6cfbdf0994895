#!/usr/bin/env python3
"""Throughput of the many-references pass (muse_batch_score_many) against R single-reference passes
on the same resident matrix.  usage: python tools/many_bench.py [rows] [rounds] [R ...]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Rs = [int(v) for v in sys.argv[3:]] or [1, 2, 4, 8]
eng = pkg.get_engine(0)
dg, ref0 = pkg.DeviceGroup.synthetic(eng, rows, 4096)
refs = [ref0] + [dg.read(997 * r + 1, 1)[0] for r in range(1, max(Rs))]
batches = [pkg.DeviceBatch(eng, dg, ref) for ref in refs]


def timed(fn):
    eng.synchronize()
    best = 1e30
    for _ in range(rounds):
        t0 = time.perf_counter()
        fn()
        eng.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for R in Rs:
    bs = batches[:R]
    pkg.score_many(bs)
    eng.synchronize()
    t_many = timed(lambda: pkg.score_many(bs))
    t_single = timed(lambda: [b.score() for b in bs])
    print("R=%d: one pass %.2f ms (%.3e series-pairs/s), %d single passes %.2f ms (%.3e /s): x%.2f" % (
        R, t_many * 1e3, R * rows / t_many, R, t_single * 1e3, R * rows / t_single, t_single / t_many), flush=True)
