#!/usr/bin/env python3
"""Random-seed soak of what round 5 added, against the CPU oracle (tests/ use fixed seeds; this draws new ones):
  * muse_batch_run_rows (Muse.Run as one call) at random lengths 2 ... 20000 and group sizes 1 ... 300, signed and abs scores, NaN /
    constant / exactly tied members, through the copy and (small groups) straight out of pinned memory;
  * the batched two-sided xCorr at random FFT lengths 512 ... 65536 with magnitudes between 1e-200 and 1e200 mixed into the pairs
    (the statistics that leave the float64 range: NaN stands / every cc zero / recomputed on rescaled copies);
  * the all-scores pass at random lengths around the real-transform kernels' ranges (4097 ... 65536: n = 8192, 16384, 32768, 65536), odd and even pads.
usage: soak_round5.py [seconds] [seed]"""
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
cases = {"run_rows": 0, "xcorr": 0, "scores": 0}
bad = 0
print("seed", seed, flush=True)


def clamp(mv, abs_scores):
    return np.minimum(np.abs(mv), 1.0) if abs_scores else np.clip(mv, -1.0, 1.0)


def soak_run_rows():
    global bad
    N = int(rng.choice([2, 8, 12, 100, 480, 1000, 4096, 5000, 20000, int(rng.integers(2, 9000))]))
    M = int(rng.choice([1, 2, 5, 50, int(rng.integers(1, 300))]))
    if M * N > 3_000_000:
        M = max(1, 3_000_000 // N)
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N)) + rng.uniform(-3, 3, (M, 1)) * np.roll(ref, int(rng.integers(-N // 3 - 1, N // 3 + 1)))[None, :]
    if M > 3 and rng.random() < 0.5:
        rows[2] = rows[0]
    if M > 1 and rng.random() < 0.3:
        rows[int(rng.integers(0, M))] = 1.5
    nan_first = rng.random() < 0.15
    if nan_first:
        rows[0, int(rng.integers(0, N))] = np.nan
    elif M > 2 and rng.random() < 0.2:
        rows[1, 0] = np.nan
    probe = pkg.DeviceGroup(eng, N, 0)
    tmpl = pkg.DeviceBatch(eng, probe, ref)
    clean = np.nan_to_num(rows, nan=0.0)
    olag, omv, gap = oracle.batch_scores(ref, clean)
    isnan = np.isnan(rows).any(axis=1)
    for abs_scores in (False, True):
        eng.rows_always_copy(bool(rng.integers(0, 2)))
        win, st = tmpl.run_rows(rows, abs_scores=abs_scores)
        sc = clamp(omv, abs_scores)
        sc[isnan] = np.nan
        exp_state = 2 if isnan[0] else 1
        best = -1
        for i in range(M):
            if np.isnan(sc[i]):
                continue
            if best < 0 or abs(sc[i]) > abs(sc[best]):
                best = i
        ok = st == exp_state
        if best >= 0:
            w = int(win["series"])
            # (the winner itself may differ only where two members tie to rounding -- then the sign may differ too; N = 2: every
            # series scores exactly +-1)
            ok = ok and 0 <= w < M and abs(abs(win["score"]) - abs(sc[best])) <= 1e-6 * abs(sc[best]) + 1e-12
            ok = ok and abs(win["score"] - sc[w]) <= 1e-6 * abs(sc[w]) + 1e-12
            if w != best:
                ok = ok and abs(abs(sc[w]) - abs(sc[best])) <= 1e-9
            elif gap[best] >= 1e-12:
                ok = ok and int(win["lag"]) == int(olag[best])
        else:
            ok = ok and int(win["series"]) == -1
        if not ok:
            bad += 1
            print("MISMATCH run_rows", N, M, abs_scores, win, st, best, flush=True)
    eng.rows_always_copy(False)
    tmpl.close()
    probe.close()
    cases["run_rows"] += 1


def soak_xcorr():
    global bad
    n = int(rng.choice([512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]))
    M = int(rng.integers(2, 7))
    Nx = n if rng.random() < 0.5 else int(rng.integers(n // 2 + 1, n + 1))
    Ny = n if rng.random() < 0.5 else int(rng.integers(n // 2 + 1, n + 1))
    X = rng.standard_normal((M, Nx)) * rng.uniform(0.1, 10, (M, 1)) + rng.normal(size=(M, 1))
    Y = rng.standard_normal((M, Ny)) * rng.uniform(0.1, 10, (M, 1))
    for i in range(M):
        r = rng.random()
        if r < 0.25:
            X[i] *= 10.0 ** float(rng.integers(-200, 201))
        elif r < 0.5:
            Y[i] *= 10.0 ** float(rng.integers(-200, 201))
        elif r < 0.6:
            X[i] *= 10.0 ** float(rng.integers(100, 201))
            Y[i] *= 10.0 ** float(rng.integers(-200, -99))
        elif r < 0.65:
            X[i] = 0.0
    normalize = bool(rng.integers(0, 2))
    cc, lag, mv, nil = eng.xcorr_batch(X, Y, n, normalize, want_cc=True)
    for i in range(M):
        occ, olag, omv = oracle.xcorr(X[i], Y[i], n, normalize)
        ok = bool(nil[i]) == (occ is None)
        if occ is not None:
            mx, my = np.max(np.abs(X[i])), np.max(np.abs(Y[i]))
            # |x| < 1e-150: the squares are denormal, sigma has a few bits in the reference as well as here -- no digits to compare;
            # normalized and |x| > 1e154: sum d^2 is Inf, and whether (sum d)^2 overflows too (NaN) or not (sigma = Inf: zeros)
            # hangs on the rounding residue of the summation ORDER (gonum sums with SIMD accumulators, the checker in sequence,
            # the device in a tree): either outcome is the reference's arithmetic
            denormal = (0 < mx < 1e-150) or (0 < my < 1e-150)
            edge = normalize and (mx > 1e154 or my > 1e154)
            if edge:
                ok = ok and (np.isnan(mv[i]) or mv[i] == 0.0)
            elif denormal and normalize:
                ok = ok and (np.isnan(mv[i]) or abs(mv[i]) <= 1.0 + 1e-9)   # (sigma out of denormal squares: a few bits at best, here and there)
            elif np.all(np.isfinite(occ)):
                scale = max(np.max(np.abs(occ)), 1e-300)
                ok = ok and np.all(np.isfinite(cc[i])) and np.max(np.abs(cc[i] - occ)) <= 1e-9 * scale + 1e-12
                ok = ok and abs(mv[i] - omv) <= 1e-6 * abs(omv) + 1e-12
            else:
                # the reference's own arithmetic left the float64 range somewhere: NaN here, or the finite values it still has
                ok = ok and (np.isnan(mv[i]) or np.isfinite(mv[i]))
        if not ok:
            bad += 1
            print("MISMATCH xcorr n=%d Nx=%d Ny=%d normalize=%s pair %d: max|x| %.3g max|y| %.3g got mv %r exp %r" % (
                n, Nx, Ny, normalize, i, np.max(np.abs(X[i])), np.max(np.abs(Y[i])), mv[i], omv), flush=True)
    cases["xcorr"] += 1


def soak_scores():
    global bad
    lo, hi = [(4097, 8192), (8193, 16384), (16385, 32768), (32769, 65536)][int(rng.integers(0, 4))]
    N = hi if rng.random() < 0.3 else int(rng.integers(lo, hi + 1))
    M = int(rng.integers(1, 12))
    t = np.arange(N)
    ref = 1.5 * (np.abs(t - N // 2) <= 5) + 0.1 * rng.standard_normal(N)
    rows = rng.uniform(-2, 2, (M, 1)) * (np.abs(t[None, :] - N // 2 - rng.integers(-300, 301, (M, 1))) <= 6) + 0.3 * rng.standard_normal((M, N)) + rng.normal(size=(M, 1)) * 10
    if M > 2:
        rows[1] = 3.25
        rows[2] *= 10.0 ** float(rng.integers(-100, 101))
    if M > 4:
        rows[4, int(rng.integers(0, N))] = np.nan
    dg = pkg.DeviceGroup.from_rows(eng, rows)
    db = pkg.DeviceBatch(eng, dg, ref)
    lag, mv = db.scores()
    olag, omv, gap = oracle.batch_scores(ref, rows)
    nan_o = np.isnan(omv)
    ok = np.array_equal(np.isnan(mv), nan_o)
    g = ~nan_o
    ok = ok and bool(np.all(np.abs(mv[g] - omv[g]) <= 1e-6 * np.abs(omv[g]) + 1e-12)) and not np.any((lag != olag) & (gap >= 1e-12) & g)
    if not ok:
        bad += 1
        print("MISMATCH scores N=%d M=%d" % (N, M), flush=True)
    db.close()
    dg.close()
    cases["scores"] += 1


t_said = time.time()
while time.time() < t_end:
    if time.time() - t_said > 60.0:  # (a GPU box takes a run that prints nothing for minutes to be hung)
        t_said = time.time()
        print("... %s cases, %d mismatches so far" % (cases, bad), flush=True)
    r = rng.random()
    if r < 0.4:
        soak_run_rows()
    elif r < 0.75:
        soak_xcorr()
    else:
        soak_scores()
print("soak: %s cases, %d mismatches (seed %d)" % (cases, bad, seed), flush=True)
sys.exit(1 if bad else 0)
