#!/usr/bin/env python3
"""Random-seed soak of what round 6 added, against the CPU oracle (tests/ use fixed seeds; this draws new ones):
  * the all-scores pass at random lengths 65 537 ... 1 048 576 (FFT lengths 2^17 ... 2^20, xcorr_huge.hip), odd and even row counts,
    constant / NaN / Inf / 1e+-60-scaled rows inside pairs, random batch sizes of the work buffer (several batches per pass);
  * the two-sided xCorr at those lengths (different lengths per side, normalised and raw, full cc);
  * groups built through staging windows with random piece sizes and commit orders, interleaved with ordinary appends, on a context
    whose allocation cache is warm with blocks of other shapes: rows read back bit for bit, scores equal to a group uploaded at once;
  * muse_batch_run_row_ptrs against muse_batch_run_rows at random lengths and group sizes;
  * small Runs (one-launch reduction, reduce_kernels.hip small_groups_kernel) on both sides of its limits (32 768 series, 2 048 label
    groups): muse_batch_run_groups against a numpy restatement of the per-group winner, muse_batch_run against the general path
    (the same label map padded with empty groups); Run(nil) against the identity label map, also beyond 65 536 series (the
    one-launch per-chunk selection, topn_ungrouped_kernel, against group_final + topn).
usage: soak_round6.py [seconds] [seed]"""
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
cases = {"huge_scores": 0, "huge_xcorr": 0, "windows": 0, "row_ptrs": 0, "small_run": 0}
bad = 0
last_note = time.time()
print("seed", seed, flush=True)


def check_scores(tag, lag, mv, olag, omv, gap, rtol=1e-6):
    global bad
    nan_o = np.isnan(omv)
    ok = ~nan_o
    good = np.array_equal(np.isnan(mv), nan_o)
    good = good and np.all(np.abs(mv[ok] - omv[ok]) <= rtol * np.abs(omv[ok]) + 1e-12)
    good = good and not np.any((lag != olag) & ok & (gap >= 1e-12)) and np.all(lag[nan_o] == 0)
    if not good:
        bad += 1
        print("MISMATCH", tag, "mv", mv[:8], "oracle", omv[:8], "lag", lag[:8], olag[:8], flush=True)


def soak_huge_scores():
    N = int(rng.choice([65537, 131072, 262144, int(rng.integers(65537, 300000)), int(rng.integers(300000, 1048577))]))
    M = int(rng.integers(1, 12 if N < 300000 else 6))
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N))
    for i in range(M):
        if rng.random() < 0.5:
            rows[i] += rng.uniform(-3, 3) * np.roll(ref, int(rng.integers(-N // 2, N // 2)))
        r = rng.random()
        if r < 0.1:
            rows[i] = rng.uniform(-5, 5)
        elif r < 0.18:
            rows[i, int(rng.integers(0, N))] = np.nan if rng.random() < 0.5 else np.inf
        elif r < 0.4:
            rows[i] *= 10.0 ** rng.integers(-60, 61)
    eng.huge_batch_mb(int(rng.choice([0, 16, 48, 256])))
    dg = pkg.DeviceGroup.from_rows(eng, rows)
    db = pkg.DeviceBatch(eng, dg, ref)
    lag, mv = db.scores()
    olag, omv, gap = oracle.batch_scores(ref, rows, nthreads=8)
    check_scores("huge_scores N=%d M=%d" % (N, M), lag, mv, olag, omv, gap)
    db.close()
    dg.close()
    eng.huge_batch_mb(0)
    cases["huge_scores"] += 1


def soak_huge_xcorr():
    global bad
    logn = int(rng.integers(17, 20))
    n = 1 << logn
    Nx = int(rng.integers(n // 2 + 1, n + 1)) if rng.random() < 0.7 else n
    Ny = int(rng.integers(1000, n + 1)) if rng.random() < 0.5 else n
    M = int(rng.integers(1, 4))
    x = rng.standard_normal((M, Nx)) * 10.0 ** rng.integers(-20, 21)
    y = rng.standard_normal((M, Ny))
    L = min(Nx, Ny)
    y[0, Ny - L:] += 2.0 * np.roll(x[0, Nx - L:], int(rng.integers(-50, 50))) / max(np.abs(x[0]).max(), 1e-300)
    if M > 1 and rng.random() < 0.5:
        y[1] = 3.0
    gx, gy = pkg.DeviceGroup.from_rows(eng, x), pkg.DeviceGroup.from_rows(eng, y)
    for normalize in (True, False):
        cc, lag, mv, nil = pkg.xcorr_groups(gx, gy, n, normalize, want_cc=True)
        for i in range(M):
            occ, olag, omv = oracle.xcorr(x[i], y[i], n, normalize)
            if occ is None:
                good = nil[i] == 1 and lag[i] == 0 and mv[i] == 0.0
            else:
                scale = max(np.max(np.abs(occ)), 1e-300)
                srt = np.sort(np.abs(occ))
                good = nil[i] == 0 and np.max(np.abs(cc[i] - occ)) <= 1e-9 * scale and abs(mv[i] - omv) <= 1e-6 * abs(omv) + 1e-300
                if srt[-1] - srt[-2] > 1e-9 * srt[-1]:
                    good = good and lag[i] == olag
            if not good:
                bad += 1
                print("MISMATCH huge_xcorr n=%d Nx=%d Ny=%d normalize=%d pair %d" % (n, Nx, Ny, normalize, i), flush=True)
    gx.close()
    gy.close()
    cases["huge_xcorr"] += 1


def soak_windows():
    global bad
    N = int(rng.choice([12, 480, 1000, 4096, 5000, int(rng.integers(2, 30000))]))
    M = int(rng.integers(1, max(2, min(40000, 60_000_000 // (8 * N)))))
    # warm the allocation cache with another shape first (freed blocks of other sizes / the same class)
    junk = pkg.DeviceGroup.from_rows(eng, rng.standard_normal((int(rng.integers(1, 200)), int(rng.integers(2, 9000)))))
    junk.close()
    rows = rng.standard_normal((M, N))
    ref = rng.standard_normal(N)
    dg = pkg.DeviceGroup(eng, N, 0)
    i = 0
    while i < M:
        if rng.random() < 0.3:
            k = int(rng.integers(1, min(M - i, 50) + 1))
            dg.append(rows[i:i + k])
            i += k
            continue
        win = dg.stage(M - i)
        k = win.shape[0]
        step = int(rng.integers(1, k + 1))
        pieces = [(lo, min(k, lo + step)) for lo in range(0, k, step)]
        for j in rng.permutation(len(pieces)):
            lo, hi = pieces[j]
            win[lo:hi] = rows[i + lo:i + hi]
            dg.commit(lo, hi - lo)
        i += k
    good = dg.M == M
    for _ in range(4):
        f = int(rng.integers(0, M))
        c = int(min(M - f, rng.integers(1, 40)))
        good = good and np.array_equal(dg.read(f, c), rows[f:f + c])
    if N >= 2 and good:
        db = pkg.DeviceBatch(eng, dg, ref)
        lag, mv = db.scores()
        dref = pkg.DeviceGroup.from_rows(eng, rows)
        dbr = pkg.DeviceBatch.like(db, dref)
        lag2, mv2 = dbr.scores()
        good = np.array_equal(lag, lag2) and np.array_equal(mv, mv2, equal_nan=True)
        sub = rng.choice(M, size=min(M, 64), replace=False)
        olag, omv, gap = oracle.batch_scores(ref, rows[sub])
        check_scores("windows N=%d M=%d" % (N, M), lag[sub], mv[sub], olag, omv, gap)
        for h in (dbr, dref, db):
            h.close()
    if not good:
        bad += 1
        print("MISMATCH windows N=%d M=%d" % (N, M), flush=True)
    dg.close()
    cases["windows"] += 1


def soak_row_ptrs():
    global bad
    N = int(rng.choice([2, 8, 480, 1000, 4096, int(rng.integers(2, 20000)), int(rng.integers(65537, 140000))]))
    M = int(rng.integers(1, 120 if N < 30000 else 6))
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N)) + rng.uniform(-2, 2, (M, 1)) * np.roll(ref, 3)[None, :]
    if rng.random() < 0.2:
        rows[0, int(rng.integers(0, N))] = np.nan
    if M > 2 and rng.random() < 0.3:
        rows[2] = rows[1]
    probe = pkg.DeviceGroup(eng, N, 0)
    tmpl = pkg.DeviceBatch(eng, probe, ref)
    for abs_scores in (False, True):
        eng.rows_always_copy(bool(rng.integers(0, 2)))
        a = tmpl.run_rows(rows, abs_scores=abs_scores)
        b = tmpl.run_row_ptrs([rows[i].copy() for i in range(M)], abs_scores=abs_scores)
        if a[1] != b[1] or a[0].tolist() != b[0].tolist():
            bad += 1
            print("MISMATCH row_ptrs N=%d M=%d" % (N, M), a, b, flush=True)
    eng.rows_always_copy(False)
    tmpl.close()
    probe.close()
    cases["row_ptrs"] += 1


def soak_small_run():
    global bad
    N = int(rng.choice([2, 8, 30, 480, int(rng.integers(2, 2000))]))
    M = int(rng.choice([int(rng.integers(1, 60)), int(rng.integers(1, 6000)), int(rng.integers(30000, 36000)), int(rng.integers(65000, 140000))]))
    if M > 40000:  # (beyond the exact feed: Run(nil) selects per chunk on the device, topn_ungrouped_kernel against the general path)
        N = int(rng.choice([2, 8, 30]))
    M = max(1, min(M, 40_000_000 // (8 * N)))
    G = int(rng.choice([1, int(rng.integers(1, 120)), int(rng.integers(1900, 2200)), M]))
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N)) + rng.uniform(-3, 3, (M, 1)) * np.roll(ref, int(rng.integers(-3, 4)))[None, :]
    for _ in range(int(rng.integers(0, 6))):
        i = int(rng.integers(0, M))
        r = rng.random()
        if r < 0.4:
            rows[i] = rng.uniform(-5, 5)
        elif r < 0.7:
            rows[i, int(rng.integers(0, N))] = np.nan
        elif i > 0:
            rows[i] = rows[i - 1]
    gid = rng.integers(0, G, M).astype(np.int32)
    if G == M and rng.random() < 0.5:
        gid = np.arange(M, dtype=np.int32)
    dg = pkg.DeviceGroup.from_rows(eng, rows)
    db = pkg.DeviceBatch(eng, dg, ref)
    lag, mv = db.scores()
    off = int(rng.integers(0, 1000))
    for abs_scores in (True, False):
        rec, st = db.run_groups(gid, G, off, abs_scores=abs_scores)
        v = np.clip(np.abs(mv) if abs_scores else mv, -1.0, 1.0)
        order = np.lexsort((np.arange(M), gid))            # by group, then by index
        gs = gid[order]
        starts = np.flatnonzero(np.r_[True, gs[1:] != gs[:-1]])
        ends = np.r_[starts[1:], M]
        want_st = np.zeros(G, dtype=np.uint8)
        want_series = np.full(G, -1, dtype=np.int64)
        for a, b in zip(starts, ends):
            idx = order[a:b]
            g = int(gs[a])
            want_st[g] = 2 if np.isnan(v[idx[0]]) else 1
            num = idx[~np.isnan(v[idx])]
            if num.size:
                want_series[g] = num[np.argmax(np.abs(v[num]))] + off
        good = np.array_equal(st, want_st) and np.array_equal(rec["series"], want_series)
        has = want_series >= 0
        w = want_series[has] - off
        good = good and np.array_equal(rec["score"][has], v[w]) and np.array_equal(rec["lag"][has], lag[w])
        good = good and not rec["score"][~has].any() and np.array_equal(rec["group"], np.arange(G))
        if not good:
            bad += 1
            print("MISMATCH small_run groups N=%d M=%d G=%d abs=%d" % (N, M, G, abs_scores), flush=True)
    kw = dict(max_lag=int(rng.integers(0, N + 1)), top_n=int(rng.choice([1, 5, 20, 256, 500])), threshold=float(rng.choice([0.0, 0.3, 0.8])),
              sign_filter=int(rng.integers(-1, 2)), abs_scores=bool(rng.integers(0, 2)))
    a = db.run(gid, G, **kw)
    b = db.run(gid, G + 2049, **kw)
    same = all(x.tolist() == y.tolist() for x, y in zip(a[:3], b[:3])) and (a[3] == b[3] or (np.isnan(a[3]) and np.isnan(b[3])))
    if not same:
        bad += 1
        print("MISMATCH small_run run N=%d M=%d G=%d" % (N, M, G), kw, flush=True)
    # Run(nil) (one thread per series, one launch up to 32 768 series) against the same Run with the identity label map
    a = db.run(None, 0, **kw)
    b = db.run(np.arange(M, dtype=np.int32), M, **kw)
    same = all(x.tolist() == y.tolist() for x, y in zip(a[:3], b[:3])) and (a[3] == b[3] or (np.isnan(a[3]) and np.isnan(b[3])))
    if not same:
        bad += 1
        print("MISMATCH small_run ungrouped N=%d M=%d" % (N, M), kw, flush=True)
    db.close()
    dg.close()
    cases["small_run"] += 1


kinds = [soak_huge_scores, soak_huge_xcorr, soak_windows, soak_row_ptrs, soak_small_run]
only = os.environ.get("SOAK_ONLY")
if only:
    kinds = [k for k in kinds if k.__name__ == "soak_" + only]
while time.time() < t_end:
    kinds[int(rng.integers(0, len(kinds)))]()
    if time.time() - last_note > 50:
        last_note = time.time()
        print("...", cases, "mismatches", bad, flush=True)
print("seed %d: %s, %d mismatches" % (seed, cases, bad))
sys.exit(1 if bad else 0)
