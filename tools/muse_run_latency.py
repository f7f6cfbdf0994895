#!/usr/bin/env python3
"""Latency of one Muse.Run call (muse.go:46-92: one small label group per call): the ABI call alone (muse_batch_run_rows on
rows that are already one matrix) and the whole Python mirror (np.stack of the Series + the call + Results.Update)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
muse = importlib.import_module("go-muse_amd")
rng = np.random.default_rng(0)
for N, K in ((8, 2), (12, 1), (480, 5), (480, 50), (4096, 5), (4096, 200), (32768, 5)):
    ref = muse.NewSeries(rng.standard_normal(N), muse.NewLabels({"graph": "ref"}))
    comp = [muse.NewSeries(rng.standard_normal(N), muse.NewLabels({"graph": "g", "host": "h%d" % k})) for k in range(K)]
    m = muse.New(ref, muse.NewResults(N, 20, 0.0, muse.SignFilter_ANY))
    m.Run(comp)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        m.Run(comp)
    dt = (time.perf_counter() - t0) / reps
    rows = np.stack([s.y for s in comp])
    t0 = time.perf_counter()
    for _ in range(reps):
        m._template.run_rows(rows, abs_scores=False)
    da = (time.perf_counter() - t0) / reps
    muse.get_engine().rows_always_copy(True)
    m._template.run_rows(rows, abs_scores=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        m._template.run_rows(rows, abs_scores=False)
    dc = (time.perf_counter() - t0) / reps
    muse.get_engine().rows_always_copy(False)
    print("N=%6d, %3d series per call: %6.1f us per Muse.Run (mirror), %6.1f us per muse_batch_run_rows, %6.1f us with the copy forced" % (N, K, dt * 1e6, da * 1e6, dc * 1e6), flush=True)
