#!/usr/bin/env python3
"""Random-shape soak of the kernels round 4 touched, against the CPU oracle (tests/ use fixed seeds; this draws new ones):
xCorrWithX all-scores passes at random lengths 4097 ... 65536 (n = 8192 ... 65536, padded and not, NaN / constant / huge-scale rows
mixed in, odd row counts) and the batched two-sided xCorr at random (Nx, Ny, n) with n = 32768 / 65536.
usage: soak_lengths.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("go-muse_amd")
from oracle import oracle_py as oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
eng = pkg.get_engine(0)
t_end = time.time() + budget
cases = bad = 0
print("seed", seed, flush=True)


def check(lag, mv, olag, omv, gap, what):
    global bad
    nan_o = np.isnan(omv)
    ok = ~nan_o
    good = np.array_equal(np.isnan(mv), nan_o)
    err = np.abs(mv[ok] - omv[ok])
    good = good and bool(np.all(err <= 1e-6 * np.abs(omv[ok]) + 1e-12))
    tie = (gap < 1e-12) & ok
    good = good and not np.any((lag != olag) & ~tie & ok)
    if not good:
        bad += 1
        print("MISMATCH", what, flush=True)


t_said = time.time()
while time.time() < t_end:
    if time.time() - t_said > 60.0:  # (a GPU box takes a run that prints nothing for minutes to be hung)
        t_said = time.time()
        print("... %d cases, %d mismatches so far" % (cases, bad), flush=True)
    # ---- xCorrWithX
    n = int(rng.choice([8192, 16384, 32768, 65536]))
    N = n if rng.random() < 0.4 else int(rng.integers(n // 2 + 1, n))
    M = int(rng.integers(3, 40))
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N)) * rng.uniform(0.01, 100.0, size=(M, 1)) + rng.standard_normal((M, 1)) * 10.0
    rows[0] = np.roll(ref, int(rng.integers(-50, 50))) * 3.0 + 0.1 * rng.standard_normal(N)
    if M > 4:
        rows[1] = 7.5                                   # sigma == 0
        rows[2, int(rng.integers(0, N))] = np.nan       # NaN row (must not poison its pair partner)
        rows[3] *= 1e25                                 # sigma spread inside a pair
    dg = pkg.DeviceGroup.from_rows(eng, rows)
    db = pkg.DeviceBatch(eng, dg, ref)
    lag, mv = db.scores()
    olag, omv, gap = oracle.batch_scores(ref, rows, nthreads=16)
    check(lag, mv, olag, omv, gap, "xCorrWithX N=%d n=%d M=%d" % (N, n, M))
    lag2, mv2 = db.scores()
    if not (np.array_equal(lag, lag2) and np.array_equal(mv, mv2, equal_nan=True)):
        bad += 1
        print("NOT REPRODUCIBLE xCorrWithX N=%d M=%d" % (N, M), flush=True)
    db.close()
    dg.close()
    cases += 1
    # ---- two-sided xCorr, long n
    n = int(rng.choice([32768, 65536]))
    if rng.random() < 0.5:
        Nx = Ny = n
    else:
        Nx, Ny = int(rng.integers(2, n + 1)), int(rng.integers(2, n + 1))
    P = int(rng.integers(1, 7))
    X = rng.standard_normal((P, Nx)) * rng.uniform(0.1, 10.0, size=(P, 1)) + 1.0
    Y = rng.standard_normal((P, Ny)) * rng.uniform(0.1, 10.0, size=(P, 1))
    if P > 2:
        X[1] *= 10.0 ** rng.uniform(-30, 30)            # a listed pair now and then
    normalize = bool(rng.integers(0, 2))
    glag, gmv, gnil = eng.xcorr_batch(X, Y, n, normalize)
    for i in range(P):
        cc, ol, om = oracle.xcorr(X[i], Y[i], n, normalize)
        if cc is None:
            okk = gnil[i] == 1 and glag[i] == 0 and gmv[i] == 0.0
        else:
            a = np.sort(np.abs(cc))[::-1]
            tie = a[0] > 0 and (a[0] - a[1]) / a[0] < 1e-12
            okk = gnil[i] == 0 and abs(gmv[i] - om) <= 1e-6 * abs(om) + 1e-12 and (glag[i] == ol or tie)
        if not okk:
            bad += 1
            print("MISMATCH two-sided n=%d Nx=%d Ny=%d normalize=%d pair %d: %s %s vs %s %s" % (n, Nx, Ny, normalize, i, glag[i], gmv[i], ol, om), flush=True)
    cases += 1
print("soak: %d cases, %d mismatches (seed %d)" % (cases, bad, seed))
sys.exit(1 if bad else 0)
