#!/usr/bin/env python3
"""Latency of one Muse.Run call (muse.go:46-92: one small label group per call) through the Python mirror."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
muse = importlib.import_module("go-muse_amd")
rng = np.random.default_rng(0)
for N, K in ((12, 1), (480, 5), (4096, 5), (4096, 200), (32768, 5)):
    ref = muse.NewSeries(rng.standard_normal(N), muse.NewLabels({"graph": "ref"}))
    comp = [muse.NewSeries(rng.standard_normal(N), muse.NewLabels({"graph": "g", "host": "h%d" % k})) for k in range(K)]
    m = muse.New(ref, muse.NewResults(N, 20, 0.0, muse.SignFilter_ANY))
    m.Run(comp)
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        m.Run(comp)
    dt = (time.perf_counter() - t0) / reps
    print("N=%6d, %3d series per call: %.1f us per Muse.Run" % (N, K, dt * 1e6), flush=True)
