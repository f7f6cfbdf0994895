"""GPU tests of round 6's cold-path work (run with -m gpu on an MI355X), all through the C ABI and against the CPU oracle:
the context's allocation cache (a reused block never leaks the previous group's rows or guard), the pinned staging
windows (muse_group_stage / muse_group_commit), the row-pointer form of Muse.Run (muse_batch_run_row_ptrs), per-slot
streams under concurrent callers with kernel timing on, and the piecewise flush of per-Series appends."""
import threading

import numpy as np
import pytest

from _load import pkg

pytestmark = pytest.mark.gpu

SCORE_RTOL = 1e-6
SCORE_ATOL = 1e-12
TIE_GAP = 1e-12


@pytest.fixture(scope="module")
def muse():
    m = pkg()
    m.build.build()
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    return m


@pytest.fixture(scope="module")
def eng(muse):
    return muse.get_engine(0)


def _check(lag, mv, olag, omv, gap):
    lag, mv, olag, omv = map(np.asarray, (lag, mv, olag, omv))
    nan_o = np.isnan(omv)
    assert np.array_equal(np.isnan(mv), nan_o)
    ok = ~nan_o
    err = np.abs(mv[ok] - omv[ok])
    assert np.all(err <= SCORE_RTOL * np.abs(omv[ok]) + SCORE_ATOL), float(err.max())
    tie = (gap < TIE_GAP) & ok
    assert not ((lag != olag) & ~tie & ok).any()


def test_pool_block_reuse_never_leaks_rows_or_guard(muse, eng, oracle):
    """NewGroup -> Add -> NewBatch -> Run -> free, over and over on one context, with shapes that land in the SAME size class
    of the allocation cache: a long-series group full of large values, then a padded short-series group (N < n: the kernels
    read up to n - N samples IN FRONT of a row -- the guard for row 0, the previous row otherwise -- and mask them), whose
    block is the one the first group just handed back.  Every pass is checked row by row against the oracle; the cache is
    observed to hold blocks in between (the reuse happens), and muse_ctx_trim empties it."""
    rng = np.random.default_rng(6001)
    eng.trim()
    assert eng.pool_stats()[0] == 0
    shapes = [(64, 4096, 1e6), (511, 513, 1.0), (256, 1024, 1e-3), (300, 700, 1.0), (64, 4096, 1.0), (37, 5000, 1.0), (181, 1025, 1e3),
              (23, 8000, 1.0), (6, 20000, 1.0)]
    for M, N, scale in shapes:
        ref = rng.standard_normal(N)
        rows = rng.standard_normal((M, N)) * scale
        rows[0] += 3.0 * scale * np.roll(ref, 2)                     # row 0 reads the guard in front of it when N < n
        rows[M // 2] = 7.0 * scale                                   # sigma == 0
        dg = muse.DeviceGroup.from_rows(eng, rows)
        db = muse.DeviceBatch(eng, dg, ref)
        lag, mv = db.scores()
        olag, omv, gap = oracle.batch_scores(ref, rows)
        _check(lag, mv, olag, omv, gap)
        # the same rows through per-Series appends (the staging pair, flushed piece by piece)
        dg2 = muse.DeviceGroup(eng, N, 0)
        for r in range(M):
            dg2.append(rows[r])
        db2 = muse.DeviceBatch.like(db, dg2)
        lag2, mv2 = db2.scores()
        assert np.array_equal(lag, lag2) and np.array_equal(mv, mv2, equal_nan=True)
        db2.close()
        dg2.close()
        db.close()
        dg.close()
        assert eng.pool_stats()[1] > 0                               # freed blocks are kept ...
    dev_bytes, dev_blocks, host_bytes, host_blocks = eng.pool_stats()
    assert 0 < dev_bytes <= (1 << 30) and host_bytes <= (192 << 20)
    eng.trim()                                                       # ... until asked for
    assert eng.pool_stats() == (0, 0, 0, 0)


@pytest.mark.parametrize("N,M", [(480, 5000), (4096, 300), (513, 70000), (20000, 40)])
def test_staging_windows_match_append(muse, eng, oracle, N, M):
    """muse_group_stage / muse_group_commit (the host packs rows straight into pinned memory, commits piece by piece, in any
    order, from several threads) gives the group muse_group_append gives: rows read back bit for bit, scores identical, and
    equal to the oracle's.  513 x 70000 needs more than one 32 MB window (the two buffers alternate); rows appended the
    ordinary way before and after a window keep their order."""
    rng = np.random.default_rng(6100 + N)
    ref = rng.standard_normal(N)
    rows = rng.standard_normal((M, N))
    rows[::5] += 2.0 * np.roll(ref, -3)
    dg = muse.DeviceGroup(eng, N, 0)
    head = 3
    dg.append(rows[:head])                                            # ordinary appends in front of the windows
    i = head
    tail = 2
    while i < M - tail:
        win = dg.stage(M - tail - i)
        k = win.shape[0]
        assert 0 < k <= M - tail - i
        with pytest.raises(muse.MuseError):                           # one window at a time; nothing else on the group meanwhile
            dg.stage(1)
        with pytest.raises(muse.MuseError):
            dg.append(rows[:1])
        pieces = [(lo, min(k, lo + max(1, k // 7))) for lo in range(0, k, max(1, k // 7))]
        order = rng.permutation(len(pieces))                          # commits in any order ...
        errs = []

        def fill(idx):
            try:
                for j in idx:
                    lo, hi = pieces[j]
                    win[lo:hi] = rows[i + lo:i + hi]
                    dg.commit(lo, hi - lo)
            except Exception as e:                                    # noqa: BLE001 (reported below on the main thread)
                errs.append(e)
        th = [threading.Thread(target=fill, args=(order[w::3],)) for w in range(3)]   # ... from three threads
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        with pytest.raises(muse.MuseError):
            dg.commit(0, 1)                                           # the window is closed
        i += k
    dg.append(rows[M - tail:])
    assert dg.M == M
    step = max(1, M // 50)
    for first in range(0, M, step * 10):
        cnt = min(step, M - first)
        assert np.array_equal(dg.read(first, cnt), rows[first:first + cnt])
    assert np.array_equal(dg.read(M - 5, 5), rows[M - 5:])
    db = muse.DeviceBatch(eng, dg, ref)
    lag, mv = db.scores()
    dg_ref = muse.DeviceGroup.from_rows(eng, rows)
    db_ref = muse.DeviceBatch.like(db, dg_ref)
    lag_r, mv_r = db_ref.scores()
    assert np.array_equal(lag, lag_r) and np.array_equal(mv, mv_r)
    sub = np.unique(np.concatenate([np.arange(0, M, max(1, M // 400)), [0, head - 1, head, M - tail - 1, M - tail, M - 1]]))
    olag, omv, gap = oracle.batch_scores(ref, rows[sub])
    _check(lag[sub], mv[sub], olag, omv, gap)
    for h in (db_ref, dg_ref, db, dg):
        h.close()


def test_append_staged_helper_and_f32_refusal(muse, eng):
    rng = np.random.default_rng(6200)
    N, M = 1000, 2500
    series = [rng.standard_normal(N) for _ in range(M)]
    dg = muse.DeviceGroup(eng, N, 0)
    dg.append_staged(series)
    assert dg.M == M and np.array_equal(dg.read(0, M), np.stack(series))
    dg.close()
    g32 = muse.DeviceGroup(eng, N, 0, f32=True)
    with pytest.raises(muse.MuseError) as ei:
        g32.stage(4)
    assert ei.value.status == muse.binding.MUSE_ERR_UNSUPPORTED
    g32.close()


@pytest.mark.parametrize("N", [8, 480, 4096, 5000])
def test_run_row_ptrs_equals_run_rows(muse, eng, N):
    """muse_batch_run_row_ptrs (one pointer per series, gathered into the pinned slot) returns what muse_batch_run_rows returns
    for the same rows packed: one to 300 series, signed and abs, NaN first member, constant member, an empty group, a NULL row."""
    rng = np.random.default_rng(6300 + N)
    ref = rng.standard_normal(N)
    probe = muse.DeviceGroup(eng, N, 0)
    tmpl = muse.DeviceBatch(eng, probe, ref)
    for M in (1, 2, 50, 300):
        rows = rng.standard_normal((M, N))
        rows[::4] += np.roll(ref, 1)
        if M >= 50:
            rows[7] = 0.25
            rows[9] = rows[5]
        for nan_first in (False, True):
            r = rows.copy()
            if nan_first:
                r[0, N // 3] = np.nan
            series = [r[i].copy() for i in range(M)]                  # separate allocations, as the Series of a Muse.Run are
            for abs_scores in (False, True):
                w1, s1 = tmpl.run_rows(r, abs_scores=abs_scores)
                w2, s2 = tmpl.run_row_ptrs(series, abs_scores=abs_scores)
                assert s1 == s2 and w1.tolist() == w2.tolist(), (N, M, nan_first, abs_scores)
    w, s = tmpl.run_row_ptrs([])
    assert s == 0 and int(w["series"]) == -1
    with pytest.raises(muse.MuseError) as ei:
        tmpl.run_row_ptrs([np.zeros(N + 1)])
    assert ei.value.status == muse.binding.MUSE_ERR_LENGTH
    tmpl.close()
    probe.close()


def test_concurrent_run_rows_on_slot_streams_with_timing_on(muse, eng, oracle):
    """Sixteen host threads drive one template (muse_test.go:203-214) while kernel timing is ON (the event list is shared by
    the callers): every call returns the winner the single-threaded call returns, the launch count adds up, groups of
    different sizes and lengths share the slot pool (n = 512 and n = 8192: the latter stays on the context's stream)."""
    rng = np.random.default_rng(6400)
    cases = []
    for N in (480, 5000):
        ref = rng.standard_normal(N)
        probe = muse.DeviceGroup(eng, N, 0)
        tmpl = muse.DeviceBatch(eng, probe, ref)
        groups = []
        for g in range(24):
            M = int(rng.integers(1, 60))
            rows = rng.standard_normal((M, N))
            rows[rng.integers(0, M)] += 2.5 * np.roll(ref, int(rng.integers(-5, 6)))
            groups.append(rows)
        expect = [tmpl.run_rows(r) for r in groups]
        cases.append((tmpl, probe, groups, expect, ref))
    eng.kernel_time()
    eng.kernel_timing(True)
    errs, calls = [], [0] * 16

    def worker(w):
        try:
            for rep in range(6):
                for tmpl, _, groups, expect, _ in cases:
                    for gi in range(w % 3, len(groups), 3):
                        win, st = tmpl.run_rows(groups[gi]) if (rep + gi) % 2 else tmpl.run_row_ptrs([x.copy() for x in groups[gi]])
                        calls[w] += 1
                        if st != expect[gi][1] or win.tolist() != expect[gi][0].tolist():
                            errs.append((w, gi, win.tolist(), expect[gi][0].tolist()))
        except Exception as e:                                        # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(w,)) for w in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    eng.synchronize()
    eng.kernel_timing(False)
    ms, launches = eng.kernel_time()
    assert not errs, errs[:3]
    assert launches == sum(calls) and ms > 0.0
    # the winners against the oracle (one group per length)
    for tmpl, probe, groups, expect, ref in cases:
        rows = groups[0]
        olag, omv, gap = oracle.batch_scores(ref, rows)
        sc = np.clip(omv, -1.0, 1.0)
        best = int(np.argmax(np.abs(sc)))
        win = expect[0][0]
        assert abs(win["score"] - sc[best]) <= SCORE_RTOL * abs(sc[best]) + SCORE_ATOL
        tmpl.close()
        probe.close()
